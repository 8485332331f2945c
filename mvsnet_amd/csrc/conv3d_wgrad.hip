// Weight gradient of the 3x3x3 SAME convolutions / transposed convolutions of RegNetUS0 on the fp32
// matrix cores (in the reference: TensorFlow's Conv3DBackpropFilterV2 behind
// opt.compute_gradients, mvsnet/train.py:428-429, of the layers in mvsnet/cnn_wrapper/network.py:171-215,
// 300-329).
//
//      dW[tap][cb][cs] = sum_o  big[STRIDE*o + tap - pad][cb] * small[o][cs]
//
// with `small` on the layer's coarse side and `big` on its fine side: a convolution has big = layer input,
// small = output gradient (dW in the conv3d layout (3,3,3,Cin,Cout)); a stride-2 transposed convolution
// has big = output gradient, small = layer input (dW in the conv3d_transpose layout (3,3,3,Cout,Cin)).
// pad = 1 for stride 1, 0 for stride 2 (TensorFlow SAME on even sizes pads only at the end).
//
// As an MFMA contraction the voxels are K: v_mfma_f32_16x16x4_f32 takes 4 voxels per instruction, rows
// = (tap, cb), columns = cs.  A lane reads 4 consecutive cb of one (voxel, tap) as one ds_read_b128 and
// feeds them to 4 MFMAs (row m of MFMA j is channel 4*(m % QB) + j of tap m / QB), so an operand group
// covers 16/QB taps x CB channels; the 27 taps are dealt over the 4 waves (and over blockIdx.y when the
// accumulators would not fit: 27*CB*CS/64 floats per lane in total).  A workgroup owns a TH x 16 tile of
// `small` voxels and marches over big planes: one big plane (+halo) and the three small planes it pairs
// with (kd = 0,1,2) are resident in LDS.  Per-workgroup partial sums go to a scratch buffer and a second
// kernel adds them in float64 in a fixed order: deterministic, no atomics.
// Roofline: MFMA, 2*27*CB*CS flops per small voxel (half of the columns idle when CS = 8).
#include "conv_common.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;
constexpr int TW = 16;

struct WgradArgs {
    const float* big; const float* small; float* partial;
    int D, H, W;            // big dims
    int Ds, Hs, Ws, CS;     // small dims / channels
    int planes_per_wg;      // small planes per workgroup
};

template <int CB, int CT, int S, int TH>
struct WGeom {
    static constexpr int QB = CB / 4;
    static constexpr int TPG = 16 / QB;                    // taps per operand group
    static constexpr int GK = (9 + TPG - 1) / TPG;         // groups per kd
    static constexpr int NGRP = 3 * GK;
    static constexpr int GPW = (8 / CT) < 4 ? (8 / CT) : 4;   // groups per wave: 16*GPW*CT accumulator registers
    static constexpr int SLICES = (NGRP + 4 * GPW - 1) / (4 * GPW);
    static constexpr int PAD = (S == 1) ? 1 : 0;
    static constexpr int BH = TH * S + (S == 1 ? 2 : 1), BW = TW * S + (S == 1 ? 2 : 1);
    static constexpr int SB = CB + 4;                      // floats per big position
    static constexpr int SS = (CT == 1) ? 16 : 16 * CT + 16;   // floats per small position (bank spread for b32 reads)
    static constexpr int BIG_FLOATS = BH * BW * SB;
    static constexpr int SMALL_FLOATS = TH * TW * SS;
    static constexpr int LDS_BYTES = (BIG_FLOATS + 3 * SMALL_FLOATS) * 4;
    static constexpr int NSTEP = TH * TW / 4;
};

template <int CB, int CT, int S, int TH>
__global__ void __launch_bounds__(256)
wgrad_kernel(WgradArgs a) {
    using G = WGeom<CB, CT, S, TH>;
    constexpr int QB = G::QB, TPG = G::TPG, GK = G::GK, NGRP = G::NGRP, GPW = G::GPW, PAD = G::PAD;
    constexpr int BH = G::BH, BW = G::BW, SB = G::SB, SS = G::SS, NSTEP = G::NSTEP;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bigl = lds;
    float* smalll = lds + G::BIG_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int tiles_w = (a.Ws + TW - 1) / TW;
    const int tile_h = blockIdx.x / tiles_w, tile_w = blockIdx.x - tile_h * tiles_w;
    const int oh0 = tile_h * TH, ow0 = tile_w * TW;
    const int od0 = blockIdx.z * a.planes_per_wg, od1 = min(od0 + a.planes_per_wg, a.Ds);
    const int gbase = blockIdx.y * 4 * GPW;

    // zero the small planes once: channels CS .. 16*CT-1 and rows/cols beyond the volume stay zero
    for (int i = tid; i < 3 * G::SMALL_FLOATS; i += 256) smalll[i] = 0.f;

    // operand addressing
    const int m_lo = m % QB, m_hi = m / QB;
    int a_off[GPW], slot_kd[GPW];
#pragma unroll
    for (int sl = 0; sl < GPW; ++sl) {
        const int g = gbase + sl * 4 + wave;
        const int gi = g % GK;
        slot_kd[sl] = (g < NGRP) ? g / GK : -1;
        int t9 = gi * TPG + m_hi; if (t9 > 8) t9 = 8;      // rows of a short last group repeat tap 8; never stored
        const int kh = t9 / 3, kw = t9 - 3 * kh;
        a_off[sl] = ((kh * BW + kw + kq * S) * SB + 4 * m_lo);
    }
    const int b_off = kq * SS + m;

    f32x4 acc[GPW][4][CT];
#pragma unroll
    for (int sl = 0; sl < GPW; ++sl)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[sl][j][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging maps
    constexpr int NFB = BH * BW * QB, NITB = (NFB + 255) / 256;
    int gofb[NITB], lofb[NITB];
#pragma unroll
    for (int i = 0; i < NITB; ++i) {
        const int f = tid + 256 * i;
        const int pos = f / QB, c4 = f - pos * QB;
        const int r = pos / BW, c = pos - r * BW;
        const int gh = oh0 * S - PAD + r, gw = ow0 * S - PAD + c;
        const bool inb = f < NFB && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        gofb[i] = inb ? ((gh * a.W + gw) * CB + 4 * c4) * 4 : OOB;
        lofb[i] = f < NFB ? pos * SB + 4 * c4 : -1;
    }
    const int big_plane_bytes = a.H * a.W * CB * 4;
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.big, 0, a.D * big_plane_bytes, 0x00020000);
    const int csq = (a.CS + 3) / 4;                        // channel quads of `small` (last may be partial)
    const int nfs = TH * TW * csq;
    const bool vec_small = (a.CS % 4) == 0;
    const size_t small_plane = (size_t)a.Hs * a.Ws * a.CS;

    const int q_first = S * od0 - PAD, q_last = S * (od1 - 1) + 2 - PAD;
    for (int q = q_first; q <= q_last; ++q) {
        __syncthreads();                                    // previous plane's reads are done
        {   // big plane q (zeros outside the volume)
            const bool plane_ok = q >= 0 && q < a.D;
            const int soff = plane_ok ? q * big_plane_bytes : 0;
            u32x4_t v[NITB];
#pragma unroll
            for (int i = 0; i < NITB; ++i)
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, gofb[i] | (plane_ok ? 0 : OOB), soff, 0);
#pragma unroll
            for (int i = 0; i < NITB; ++i)
                if (lofb[i] >= 0) *(u32x4_t*)(bigl + lofb[i]) = v[i];
        }
        if ((q + PAD) % S == 0) {                           // the newest small plane this big plane pairs with (kd = 0)
            const int od = (q + PAD) / S;
            if (od >= od0 && od < od1) {
                float* sp = smalll + (od % 3) * G::SMALL_FLOATS;
                const float* gp = a.small + (size_t)od * small_plane;
                for (int f = tid; f < nfs; f += 256) {
                    const int pos = f / csq, c4 = f - pos * csq;
                    const int r = pos / TW, c = pos - r * TW;
                    const int oh = oh0 + r, ow = ow0 + c;
                    const bool inb = oh < a.Hs && ow < a.Ws;
                    const float* p = gp + ((size_t)oh * a.Ws + ow) * a.CS + 4 * c4;
                    if (vec_small) {
                        float4 v = inb ? *(const float4*)p : make_float4(0.f, 0.f, 0.f, 0.f);
                        *(float4*)(sp + pos * SS + 4 * c4) = v;
                    } else {
                        for (int k = 0; k < 4 && 4 * c4 + k < a.CS; ++k) sp[pos * SS + 4 * c4 + k] = inb ? p[k] : 0.f;
                    }
                }
            }
        }
        __syncthreads();

#pragma unroll
        for (int sl = 0; sl < GPW; ++sl) {
            const int kd = slot_kd[sl];
            if (kd < 0) continue;
            const int num = q + PAD - kd;
            if (num < 0 || (num % S) != 0) continue;
            const int od = num / S;
            if (od < od0 || od >= od1) continue;
            const float* ap = bigl + a_off[sl];
            const float* bp = smalll + (od % 3) * G::SMALL_FLOATS + b_off;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                constexpr int SPR = TW / 4;                 // steps per tile row
                const int oh = st / SPR, owb = 4 * (st % SPR);
                const f32x4 av = *(const f32x4*)(ap + (oh * S * BW + owb * S) * SB);
                float bv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bv[ct] = bp[(oh * TW + owb) * SS + 16 * ct];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[sl][j][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[ct], acc[sl][j][ct], 0, 0, 0);
            }
        }
    }

    // partial sums of this workgroup: (27, CB, CS) floats at slot (tile, chunk)
    const size_t wsz = (size_t)27 * CB * a.CS;
    float* out = a.partial + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * wsz;
#pragma unroll
    for (int sl = 0; sl < GPW; ++sl) {
        const int g = gbase + sl * 4 + wave;
        if (g >= NGRP) continue;
        const int kd = g / GK, gi = g % GK;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kq + r;
                    const int t9 = gi * TPG + row / QB, cb = 4 * (row % QB) + j, cs = 16 * ct + m;
                    if (t9 < 9 && cs < a.CS) out[((size_t)(kd * 9 + t9) * CB + cb) * a.CS + cs] = acc[sl][j][ct][r];
                }
    }
}

// Stride-1 variant with taps packed ACROSS kd ("small-stationary"): the workgroup marches over small planes with
// the three big planes a small plane pairs with resident in a ring, so a row's tap may sit in any of them and the
// 27 taps pack into ceil(27 / TPG) operand groups instead of 3 * ceil(9 / TPG).  That pays when the row tensor is
// narrow: for 3dconv0_1 the roles are swapped (rows = the 8-channel output gradient with its taps, columns = the 32
// cost-volume channels: R(tap, c8, c32) = sum_o g(o + tap - 1, c8) * cost(o, c32) = dW(2 - tap, c32, c8)), 8 taps
// per group -> 4 groups x 2 column tiles = 32 MFMAs per 4 voxels instead of 60, every tile full.
template <int CB, int CT, int TH>
__global__ void __launch_bounds__(256)
wgrad_flat_kernel(WgradArgs a) {
    constexpr int QB = CB / 4, TPG = 16 / QB, NGRP = (27 + TPG - 1) / TPG;
    constexpr int GPW = (NGRP + 3) / 4;
    constexpr int BH = TH + 2, BW = TW + 2, SB = CB + 4, SS = (CT == 1) ? 16 : 16 * CT + 16;
    constexpr int BIG_FLOATS = BH * BW * SB, SMALL_FLOATS = TH * TW * SS, NSTEP = TH * TW / 4;
    static_assert(GPW * 4 * CT <= 32, "accumulator budget");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* bigl = lds;                               // ring of 3 planes
    float* smalll = lds + 3 * BIG_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int tiles_w = (a.Ws + TW - 1) / TW;
    const int tile_h = blockIdx.x / tiles_w, tile_w = blockIdx.x - tile_h * tiles_w;
    const int oh0 = tile_h * TH, ow0 = tile_w * TW;
    const int od0 = blockIdx.z * a.planes_per_wg, od1 = min(od0 + a.planes_per_wg, a.Ds);

    for (int i = tid; i < SMALL_FLOATS; i += 256) smalll[i] = 0.f;        // channels CS .. 16*CT-1 stay zero

    const int m_lo = m % QB, m_hi = m / QB;
    int a_tap[GPW], a_kd[GPW];
#pragma unroll
    for (int sl = 0; sl < GPW; ++sl) {
        int t27 = (sl * 4 + wave) * TPG + m_hi; if (t27 > 26) t27 = 26;    // rows past tap 26 repeat it; never stored
        const int kd = t27 / 9, t9 = t27 - 9 * kd, kh = t9 / 3, kw = t9 - 3 * kh;
        a_kd[sl] = kd;
        a_tap[sl] = (kh * BW + kw + kq) * SB + 4 * m_lo;
    }
    const int b_off = kq * SS + m;
    f32x4 acc[GPW][4][CT];
#pragma unroll
    for (int sl = 0; sl < GPW; ++sl)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[sl][j][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NFB = BH * BW * QB, NITB = (NFB + 255) / 256;
    int gofb[NITB], lofb[NITB];
#pragma unroll
    for (int i = 0; i < NITB; ++i) {
        const int f = tid + 256 * i;
        const int pos = f / QB, c4 = f - pos * QB;
        const int r = pos / BW, c = pos - r * BW;
        const int gh = oh0 - 1 + r, gw = ow0 - 1 + c;
        const bool inb = f < NFB && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        gofb[i] = inb ? ((gh * a.W + gw) * CB + 4 * c4) * 4 : OOB;
        lofb[i] = f < NFB ? pos * SB + 4 * c4 : -1;
    }
    const int big_plane_bytes = a.H * a.W * CB * 4;
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.big, 0, a.D * big_plane_bytes, 0x00020000);
    const int csq = (a.CS + 3) / 4, nfs = TH * TW * csq;
    const size_t small_plane = (size_t)a.Hs * a.Ws * a.CS;

    auto stage_big = [&](int q) {                      // plane q -> ring slot (q + 3) % 3 (zeros outside the volume)
        const bool plane_ok = q >= 0 && q < a.D;
        float* dst = bigl + ((q + 3) % 3) * BIG_FLOATS;
        u32x4_t v[NITB];
#pragma unroll
        for (int i = 0; i < NITB; ++i)
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, gofb[i] | (plane_ok ? 0 : OOB), plane_ok ? q * big_plane_bytes : 0, 0);
#pragma unroll
        for (int i = 0; i < NITB; ++i)
            if (lofb[i] >= 0) *(u32x4_t*)(dst + lofb[i]) = v[i];
    };
    stage_big(od0 - 1);
    stage_big(od0);
    for (int od = od0; od < od1; ++od) {
        __syncthreads();                                // the previous plane's reads are done
        stage_big(od + 1);
        {
            const float* gp = a.small + (size_t)od * small_plane;
            for (int f = tid; f < nfs; f += 256) {
                const int pos = f / csq, c4 = f - pos * csq;
                const int r = pos / TW, c = pos - r * TW;
                const int oh = oh0 + r, ow = ow0 + c;
                const bool inb = oh < a.Hs && ow < a.Ws;
                const float* p = gp + ((size_t)oh * a.Ws + ow) * a.CS + 4 * c4;
                *(float4*)(smalll + pos * SS + 4 * c4) = inb ? *(const float4*)p : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < GPW; ++sl) {
            if ((sl * 4 + wave) * TPG > 26) continue;   // wave-uniform: this slot has no taps
            const float* ap = bigl + ((od + a_kd[sl] - 1 + 3) % 3) * BIG_FLOATS + a_tap[sl];
            const float* bp = smalll + b_off;
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                constexpr int SPR = TW / 4;
                const int oh = st / SPR, owb = 4 * (st % SPR);
                const f32x4 av = *(const f32x4*)(ap + (oh * BW + owb) * SB);
                float bv[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bv[ct] = bp[(oh * TW + owb) * SS + 16 * ct];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[sl][j][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[ct], acc[sl][j][ct], 0, 0, 0);
            }
        }
    }
    const size_t wsz = (size_t)27 * CB * a.CS;
    float* out = a.partial + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * wsz;
#pragma unroll
    for (int sl = 0; sl < GPW; ++sl)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kq + r;
                    const int t27 = (sl * 4 + wave) * TPG + row / QB, cb = 4 * (row % QB) + j, cs = 16 * ct + m;
                    if (t27 < 27 && cs < a.CS) out[((size_t)t27 * CB + cb) * a.CS + cs] = acc[sl][j][ct][r];
                }
}

// dW[i] = sum_p partial[p][i] in float64, fixed order: a workgroup owns 16 consecutive elements, its 16
// "p lanes" walk the partials 16 apart (64-byte segments per row), then fold through LDS in lane order.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ partial, int P, size_t wsz, float* __restrict__ dw) {
    __shared__ double red[16][17];
    const int e = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + e;
    double s = 0.0;
    if (i < wsz)
        for (int p = pl; p < P; p += 16) s += (double)partial[(size_t)p * wsz + i];
    red[pl][e] = s;
    __syncthreads();
    if (pl == 0 && i < wsz) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e];
        dw[i] = (float)t;
    }
}

struct Plan { int tiles, chunks, ppw, slices; size_t lds; };

template <int CB, int CT, int S, int TH>
Plan make_plan(int Ds, int Hs, int Ws) {
    using G = WGeom<CB, CT, S, TH>;
    Plan p;
    p.tiles = ((Hs + TH - 1) / TH) * ((Ws + TW - 1) / TW);
    p.slices = G::SLICES;
    long long want = 1024 / ((long long)p.tiles * p.slices);
    int chunks = (int)(want < 1 ? 1 : want);
    int maxc = Ds / 2 > 0 ? Ds / 2 : 1;                    // >= 2 small planes per workgroup (halo planes are re-staged)
    if (chunks > maxc) chunks = maxc;
    p.ppw = (Ds + chunks - 1) / chunks;
    p.chunks = (Ds + p.ppw - 1) / p.ppw;
    p.lds = G::LDS_BYTES;
    return p;
}

template <int CB, int CT, int S, int TH>
int run(const float* big, const float* small, int D, int H, int W, int CS, void* ws, size_t ws_bytes, float* dw,
        size_t* need, hipStream_t st) {
    const int Ds = (D + S - 1) / S, Hs = (H + S - 1) / S, Ws = (W + S - 1) / S;
    const Plan p = make_plan<CB, CT, S, TH>(Ds, Hs, Ws);
    const size_t wsz = (size_t)27 * CB * CS;
    const size_t bytes = (size_t)p.tiles * p.chunks * wsz * sizeof(float);
    if (need) { *need = bytes; return 0; }
    if (ws_bytes < bytes) return MVS_E_WORKSPACE;
    if ((long long)D * H * W * CB * 4 >= (1LL << 31)) return MVS_E_SHAPE;     // 32-bit buffer offsets
    WgradArgs a{big, small, (float*)ws, D, H, W, Ds, Hs, Ws, CS, p.ppw};
    auto kern = wgrad_kernel<CB, CT, S, TH>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    dim3 grid(p.tiles, p.slices, p.chunks);
    kern<<<grid, 256, p.lds, st>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    wgrad_reduce_kernel<<<mvs_cdiv((long long)wsz, 16), 256, 0, st>>>((const float*)ws, p.tiles * p.chunks, wsz, dw);
    return (int)hipGetLastError();
}

int dispatch(const float* big, const float* small, int D, int H, int W, int CB, int CS, int stride, void* ws,
             size_t ws_bytes, float* dw, size_t* need, hipStream_t st) {
    if (stride == 2 && ((D | H | W) & 1)) return MVS_E_SHAPE;     // pad_before = 0 holds for even sizes only
    if (CS == 1 && CB == 8 && stride == 1) {                      // 3dconv6_2: one small channel, VALU kernel (conv3d_c1.hip)
        int rows, ppw;
        int rc = mvs_wgrad_c1_plan(D, H, W, CB, &rows, &ppw);
        if (rc) return rc;
        const size_t wsz = (size_t)27 * CB, bytes = (size_t)rows * wsz * sizeof(float);
        if (need) { *need = bytes; return 0; }
        if (ws_bytes < bytes) return MVS_E_WORKSPACE;
        if ((rc = mvs_wgrad_c1_launch(big, small, D, H, W, CB, (float*)ws, st))) return rc;
        wgrad_reduce_kernel<<<mvs_cdiv((long long)wsz, 16), 256, 0, st>>>((const float*)ws, rows, wsz, dw);
        return (int)hipGetLastError();
    }
    if (CB == 8 && CS == 32 && stride == 1) {                     // swapped 3dconv0_1: taps packed across kd
        constexpr int TH = 8, CTF = 2;
        const int tiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        long long want = 1024 / tiles;
        int chunks = (int)(want < 1 ? 1 : want);
        if (chunks > (D / 2 > 0 ? D / 2 : 1)) chunks = D / 2 > 0 ? D / 2 : 1;
        const int ppw = (D + chunks - 1) / chunks; chunks = (D + ppw - 1) / ppw;
        const size_t wsz = (size_t)27 * CB * CS, bytes = (size_t)tiles * chunks * wsz * sizeof(float);
        if (need) { *need = bytes; return 0; }
        if (ws_bytes < bytes) return MVS_E_WORKSPACE;
        if ((long long)D * H * W * CB * 4 >= (1LL << 31)) return MVS_E_SHAPE;
        WgradArgs a{big, small, (float*)ws, D, H, W, D, H, W, CS, ppw};
        constexpr int SSF = 16 * CTF + 16;
        const size_t lds = (size_t)(3 * (TH + 2) * (TW + 2) * (8 + 4) + TH * TW * SSF) * sizeof(float);
        wgrad_flat_kernel<8, CTF, TH><<<dim3(tiles, 1, chunks), 256, lds, st>>>(a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        wgrad_reduce_kernel<<<mvs_cdiv((long long)wsz, 16), 256, 0, st>>>((const float*)ws, tiles * chunks, wsz, dw);
        return (int)hipGetLastError();
    }
    const int ct = (CS + 15) / 16;
#define WG_CASE(cb, ctv, s, th) \
    if (CB == cb && ct == ctv && stride == s) return run<cb, ctv, s, th>(big, small, D, H, W, CS, ws, ws_bytes, dw, need, st);
    WG_CASE(32, 1, 1, 8) WG_CASE(16, 1, 1, 8) WG_CASE(32, 2, 1, 8) WG_CASE(64, 4, 1, 4) WG_CASE(8, 1, 1, 8)
    WG_CASE(32, 1, 2, 4) WG_CASE(16, 2, 2, 4) WG_CASE(32, 4, 2, 4) WG_CASE(8, 1, 2, 4)
    WG_CASE(16, 1, 2, 4) WG_CASE(64, 2, 2, 4) WG_CASE(8, 1, 1, 8)
#undef WG_CASE
    return MVS_E_SHAPE;
}

}  // namespace

extern "C" size_t mvs_conv3d_wgrad_workspace_bytes(int D, int H, int W, int Cbig, int Csmall, int stride) {
    size_t need = 0;
    int rc = dispatch(nullptr, nullptr, D, H, W, Cbig, Csmall, stride, nullptr, 0, nullptr, &need, nullptr);
    return rc == 0 ? need : 0;
}

extern "C" int mvs_conv3d_wgrad_f32(const float* big, const float* small, int D, int H, int W, int Cbig,
                                    int Csmall, int stride, void* workspace, size_t workspace_bytes,
                                    float* dw, void* stream) {
    MVS_CHECK_ARG(big && small && workspace && dw && D > 0 && H > 0 && W > 0 && Cbig > 0 && Csmall > 0);
    MVS_CHECK_ARG(stride == 1 || stride == 2);
    return dispatch(big, small, D, H, W, Cbig, Csmall, stride, workspace, workspace_bytes, dw, nullptr, mvs_stream(stream));
}
