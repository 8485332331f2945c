// fp32-MFMA stride-2 3x3x3 convolution (the encoder's down-sampling layers 3dconv1_0/2_0/3_0,
// mvsnet/cnn_wrapper/mvsnetworks.py:130-133; tf.layers.conv3d SAME stride 2, network.py:210).
//
// Same input-stationary plane march as conv3d_mfma.hip.  Output voxel (od,oh,ow) reads input
// (2od+kd-pd, 2oh+kh-ph, 2ow+kw-pw) (pad_before = 0 for even sizes).  An input plane q = 2od+kd-pd
// feeds output plane od = (q+pd)/2 with kd = 0 and od-1 with kd = 2 when q+pd is even, and od with
// kd = 1 when it is odd; two accumulator blocks (od even / odd) are live at a time.  The staged slab
// keeps even and odd input columns in separate runs so that the 16 voxels of an MFMA column tile
// (consecutive ow, i.e. input columns 2ow+kw) are unit-stride in LDS for every kw.
#include "conv_common.h"

namespace {

constexpr int TW = CONV_TW;
constexpr int IW = 2 * TW + 1;          // staged input columns: 2*ow0-pw .. +32
constexpr int NEVEN = TW + 1;           // even-indexed staged columns (c = 0,2,..,32)

// COUT = 16 output channels per workgroup (blockIdx.y), TOH x 16 output pixels, 4 waves.
template <int CIN, int TOH, bool HAS_X2>
__global__ void __launch_bounds__(256, 1)
conv3d_s2_kernel(ConvArgs a) {
    constexpr int COUT = 16;
    constexpr int S = SlabGeom<CIN>::S;
    constexpr int IH = 2 * TOH + 1;
    constexpr int NPOS = IH * IW;
    constexpr int CQ = CIN / 4;
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    constexpr int V = TOH / 4;                     // output rows (voxel tiles) per wave
    constexpr int NROWS = 3 * COUT;
    constexpr int WROW = NROWS * 4;
    constexpr int W_FLOATS = 9 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                              // [9 taps][CQ][3 kd][16 co][4]
    float* slab = smem + W_FLOATS;                 // [2][NPOS][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;

    const int Do = (a.D + 1) / 2, Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;
    const int tiles_w = (Wo + TW - 1) / TW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int oh0 = tile_h * TOH, ow0 = tile_w * TW;
    const int co_base = blockIdx.y * COUT;
    const int od0 = blockIdx.z * a.planes_per_wg;
    const int od1 = min(od0 + a.planes_per_wg, Do);
    const int T = 2 * (od1 - od0) + 1;             // input planes 2*od0-pd .. 2*(od1-1)+2-pd
    const int q0 = 2 * od0 - a.pd;
    const int ih0 = 2 * oh0 - a.ph, iw0 = 2 * ow0 - a.pw;

    if (a.wprep) load_prepared_weights(wl, a.wprep, W_FLOATS);
    else for (int i = tid; i < W_FLOATS; i += 256) {
        int j = i & 3;
        int r = (i >> 2) % NROWS;
        int g = (i >> 2) / NROWS;
        int ciq = g % CQ, tap = g / CQ;
        int kd = r / COUT, co = r - kd * COUT;
        wl[i] = a.w[(((size_t)(kd * 9 + tap)) * CIN + ciq * 4 + j) * a.cout_total + co_base + co];
    }

    const int c4 = tid % CQ;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sc2 = sc, sh2 = sh;
    const bool has_aff = a.xs != nullptr || a.bn.stats != nullptr;
    if (a.xs) { sc = *(const float4*)(a.xs + 4 * c4); sh = *(const float4*)(a.xb + 4 * c4); }
    else if (a.bn.stats) bn_affine4(a.bn, 4 * c4, sc, sh);
    const bool has_aff2 = HAS_X2 && (a.x2s != nullptr || a.bn2.stats != nullptr);
    if (HAS_X2 && a.x2s) { sc2 = *(const float4*)(a.x2s + 4 * c4); sh2 = *(const float4*)(a.x2b + 4 * c4); }
    else if (HAS_X2 && a.bn2.stats) bn_affine4(a.bn2, 4 * c4, sc2, sh2);

    float4 pre[NIT];
    float4 pre2[HAS_X2 ? NIT : 1];

    // Per-thread staging map, identical for every plane: element offset inside one input plane
    // (-1 = outside the volume -> SAME padding zero) and float offset inside the LDS slab (-1 = none).
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int f = tid + 256 * i;
        int pos = f / CQ;
        int r = pos / IW, c = pos - r * IW;
        int gh = ih0 + r, gw = iw0 + c;
        bool inb = (f < NF4) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? (gh * a.W + gw) * CIN + 4 * c4 : -1;
        loff[i] = (f < NF4) ? (r * IW + ((c & 1) ? NEVEN + (c >> 1) : (c >> 1))) * S + 4 * c4 : -1;
    }
    const size_t plane_elems = (size_t)a.H * a.W * CIN;

    auto issue_loads = [&](int q) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
        const float* px = a.x + (size_t)(plane_ok ? q : 0) * plane_elems;
        const float* px2 = HAS_X2 ? a.x2 + (size_t)(plane_ok ? q : 0) * plane_elems : nullptr;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = plane_ok && goff[i] >= 0;
            pre[i] = ok ? *(const float4*)(px + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (HAS_X2) pre2[i] = ok ? *(const float4*)(px2 + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto write_slab = [&](int q, float* buf) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (loff[i] < 0) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plane_ok && goff[i] >= 0) {             // SAME padding pads the NORMALISED input with 0
                v = bn_relu4(pre[i], sc, sh, has_aff);
                if (HAS_X2) {
                    float4 v2 = bn_relu4(pre2[i], sc2, sh2, has_aff2);
                    v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
                }
            }
            *(float4*)(buf + loff[i]) = v;
        }
    };

    f32x4 acc[2][V];                               // block = output plane parity (relative to od0)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[b][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = (2 * (V * wave + v) * IW + n) * S + 4 * kq;
    const int a_lane = (kq * NROWS + n) * 4;

    // acc[BLK] += W[KD] * slab for all 9 in-plane taps
    auto sweep = [&](auto Bc, auto Kc, const float* buf) __attribute__((always_inline)) {
        constexpr int BLK = decltype(Bc)::value, KD = decltype(Kc)::value;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int kwoff = (kw == 0) ? 0 : (kw == 1 ? NEVEN : 1);
#pragma unroll
                for (int s = 0; s < CIN / 16; ++s) {
                    f32x4 bv[V];
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        bv[v] = *(const f32x4*)(buf + b_off[v] + (kh * IW + kwoff) * S + 16 * s);
                    f32x4 av = *(const f32x4*)(wl + a_lane + KD * COUT * 4 + ((kh * 3 + kw) * CQ + 4 * s) * WROW);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            acc[BLK][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[v][j], acc[BLK][v], 0, 0, 0);
                }
            }
        }
    };
    // both blocks in one pass over the slab (even input planes): block NEW gets kd=0, OLD gets kd=2
    auto sweep2 = [&](auto Nc, const float* buf) __attribute__((always_inline)) {
        constexpr int NEWB = decltype(Nc)::value, OLDB = 1 - NEWB;
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int kwoff = (kw == 0) ? 0 : (kw == 1 ? NEVEN : 1);
#pragma unroll
                for (int s = 0; s < CIN / 16; ++s) {
                    f32x4 bv[V];
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        bv[v] = *(const f32x4*)(buf + b_off[v] + (kh * IW + kwoff) * S + 16 * s);
                    const float* wp = wl + a_lane + ((kh * 3 + kw) * CQ + 4 * s) * WROW;
                    f32x4 a0 = *(const f32x4*)(wp);                     // kd = 0
                    f32x4 a2 = *(const f32x4*)(wp + 2 * COUT * 4);      // kd = 2
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            acc[NEWB][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], bv[v][j], acc[NEWB][v], 0, 0, 0);
                            acc[OLDB][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j], bv[v][j], acc[OLDB][v], 0, 0, 0);
                        }
                }
            }
        }
    };

    auto retire = [&](auto Bc, int od) __attribute__((always_inline)) {
        constexpr int BLK = decltype(Bc)::value;
        const bool plane_ok = (od >= od0) && (od < od1);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            int oh = oh0 + V * wave + v, ow = ow0 + n;
            if (plane_ok && oh < Ho && ow < Wo) {
                f32x4 r = acc[BLK][v];
                float* dst = a.y + ((((size_t)od * Ho + oh) * Wo) + ow) * a.cout_total + co_base + 4 * kq;
                *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
            }
            acc[BLK][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    issue_loads(q0);
    write_slab(q0, slab);
    __syncthreads();

    // t even: od = od0 + t/2 starts (kd=0), od-1 finishes (kd=2); t odd: od = od0 + (t-1)/2 gets kd=1
    auto plane = [&](auto Pc, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;      // t mod 4
        const int q = q0 + t;
        float* cur = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
        const bool more = (t + 1 < T);
        if (more) issue_loads(q + 1);
        const bool in_vol = (q >= 0) && (q < a.D);
        if (P == 0 || P == 2) {
            constexpr int NEWB = (P == 0) ? 0 : 1;
            const int od_new = od0 + t / 2;
            if (in_vol) {
                if (od_new < od1) sweep2(std::integral_constant<int, NEWB>{}, cur);
                else sweep(std::integral_constant<int, 1 - NEWB>{}, std::integral_constant<int, 2>{}, cur);
            }
            retire(std::integral_constant<int, 1 - NEWB>{}, od_new - 1);
        } else {
            constexpr int BLK = (P == 1) ? 0 : 1;
            if (in_vol) sweep(std::integral_constant<int, BLK>{}, std::integral_constant<int, 1>{}, cur);
        }
        if (more) write_slab(q + 1, nxt);
        __syncthreads();
    };
    for (int t = 0; t < T; t += 4) {
        plane(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) plane(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) plane(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < T) plane(std::integral_constant<int, 3>{}, t + 3);
    }

    if (a.stats) stats_commit<COUT>(st_s, st_q, false, slab, a.stats, a.cout_total, co_base);
}

template <int CIN, int TOH>
int launch_s2(const ConvArgs& a0, int Cout, hipStream_t st) {
    ConvArgs a = a0;
    const int Do = (a.D + 1) / 2, Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;
    const int tiles = ((Ho + TOH - 1) / TOH) * ((Wo + TW - 1) / TW);
    const int groups = Cout / 16;
    a.planes_per_wg = conv_pick_planes(Do, (long long)tiles * groups, 1);
    dim3 grid(tiles, groups, (Do + a.planes_per_wg - 1) / a.planes_per_wg);
    size_t smem = (size_t)(9 * (CIN / 4) * 48 * 4 + 2 * (2 * TOH + 1) * IW * SlabGeom<CIN>::S) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s2_kernel<CIN, TOH, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s2_kernel<CIN, TOH, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (a.x2) conv3d_s2_kernel<CIN, TOH, true><<<grid, 256, smem, st>>>(a);
    else conv3d_s2_kernel<CIN, TOH, false><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_conv3d_s2_mfma(const ConvArgs& a, int Cin, int Cout, hipStream_t st) {
    if (Cout % 16 != 0) return MVS_E_SHAPE;
    if (Cin == 32) return launch_s2<32, 4>(a, Cout, st);
    if (Cin == 16) return launch_s2<16, 4>(a, Cout, st);
    return MVS_E_SHAPE;
}
