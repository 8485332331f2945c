// fp32-MFMA 3x3x3 stride-1 convolution for EIGHT input channels (Cout a multiple of 32): the input gradient of
// 3dconv0_1 on the training path (8 -> 32: conv of the output gradient with the flipped, transposed kernel,
// backward.py).  The generic stride-1 kernel tiles K in groups of 16 input channels, so an 8-channel input had to be
// zero-padded to 16 (half of every MFMA multiplying zeros, plus the padding copy).  Here a K group is TWO taps that
// are adjacent along w: staged positions are 8 floats (32 B) with no padding, so the 16 floats behind a position are
// its own 8 channels followed by its right neighbour's, and one ds_read_b128 per lane at (position, 4*kq) feeds the
// four k-steps of taps (kw, kw+1) at once.  Per (kh): pairs (kw 0,1) and (kw 2, zero-weight pad) -> 6 groups per
// plane instead of 9.  Same input-stationary plane march as conv3d_mfma.hip (plane q feeds output planes q+1, q,
// q-1; rows = (kd, cout), columns = 16 voxels along w), no BatchNorm / skip / statistics: it is a plain convolution.
// Roofline: MFMA, 2*27*8*Cout flops per voxel at 75 % tile utilisation.
#include "conv_common.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;
constexpr int TH = 8, TWK = 16, PW = TWK + 3, PH = TH + 2;     // +2 halo columns, +1 column the (kw 2, pad) pair reaches
constexpr int S = 8;                                           // floats per staged position: unpadded on purpose
constexpr int NPOS = PH * PW, SLAB = NPOS * S;
constexpr int NGRP = 6;                                        // (kh, pair)

template <int COUT>
__global__ void __launch_bounds__(256)
conv3d_k8_kernel(ConvArgs a) {
    constexpr int MT = COUT / 16;
    constexpr int W_FLOATS = 3 * NGRP * 16 * COUT;             // [kd][group][k = 16][cout]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                                          // A fragments: ((kd*NGRP + g)*4 + kq)*COUT*4 + m*4 + j
    float* slab = smem + W_FLOATS;                             // [2][NPOS][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int tiles_w = (a.W + TWK - 1) / TWK;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TWK;
    const int d0 = blockIdx.z * a.planes_per_wg, d1 = min(d0 + a.planes_per_wg, a.D);
    const int T = d1 - d0 + 2;

    // weights: w (3,3,3,8,cout_total) -> k index of a group = 8*(tap of the pair) + ci; the pad tap (kw = 3) is zero
    for (int i = tid; i < W_FLOATS; i += 256) {
        const int j = i & 3, m = (i >> 2) % COUT;
        int r = (i >> 2) / COUT;
        const int kqq = r & 3; r >>= 2;
        const int g = r % NGRP, kd = r / NGRP;
        const int k = 4 * kqq + j, kh = g >> 1, kw = 2 * (g & 1) + (k >> 3), ci = k & 7;
        wl[i] = kw < 3 ? a.w[(((size_t)(kd * 9 + kh * 3 + kw)) * 8 + ci) * a.cout_total + m] : 0.f;
    }
    for (int i = tid; i < 2 * SLAB; i += 256) slab[i] = 0.f;   // the pad column stays zero

    // staging map: (PH rows) x (TWK + 2 columns) x 2 channel quads
    constexpr int NF4 = PH * (TWK + 2) * 2, NIT = (NF4 + 255) / 256;
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int f = tid + 256 * i;
        const int pos = f >> 1, c4 = f & 1;
        const int r = pos / (TWK + 2), c = pos - r * (TWK + 2);
        const int gh = h0 - 1 + r, gw = w0 - 1 + c;
        const bool inb = f < NF4 && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? ((gh * a.W + gw) * 8 + 4 * c4) * 4 : OOB;
        loff[i] = f < NF4 ? (r * PW + c) * S + 4 * c4 : -1;
    }
    const int plane_bytes = a.H * a.W * 8 * 4;
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.D * plane_bytes, 0x00020000);
    u32x4_t pre[NIT];
    auto issue = [&](int q) __attribute__((always_inline)) {
        const bool ok = q >= 0 && q < a.D;
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, goff[i] | (ok ? 0 : OOB), ok ? q * plane_bytes : 0, 0);
    };
    auto stage = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (loff[i] >= 0) *(u32x4_t*)(buf + loff[i]) = pre[i];
    };

    f32x4 acc[3][MT][2];                                       // [kd][row tile][tile row of this wave]
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 2; ++v) acc[kd][mt][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int a_lane = kq * COUT * 4 + n * 4;                  // + mt*64, + (kd*NGRP + g)*16*COUT
    int b_off[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) b_off[v] = ((2 * wave + v) * PW + n) * S + 4 * kq;

    issue(d0 - 1);
    __syncthreads();                                           // slab zeroed, weights in place
    stage(slab);
    issue(d0);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int q = d0 - 1 + t;
        const float* cur = slab + (t & 1) * SLAB;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            const int kh = g >> 1, kw0 = 2 * (g & 1);
            f32x4 bv[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) bv[v] = *(const f32x4*)(cur + b_off[v] + (kh * PW + kw0) * S);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const f32x4 av = *(const f32x4*)(wl + a_lane + mt * 64 + (kd * NGRP + g) * 16 * COUT);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int v = 0; v < 2; ++v)
                            acc[kd][mt][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[v][j], acc[kd][mt][v], 0, 0, 0);
                }
        }
        // output plane q-1 has now received kd = 2 (its last contribution)
        const int o = q - 1;
        if (o >= d0 && o < d1) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int h = h0 + 2 * wave + v, w = w0 + n;
                if (h < a.H && w < a.W) {
                    float* dst = a.y + (((size_t)o * a.H + h) * a.W + w) * a.cout_total + 4 * kq;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 r = acc[2][mt][v];
                        *(float4*)(dst + 16 * mt) = make_float4(r[0], r[1], r[2], r[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                acc[2][mt][v] = acc[1][mt][v]; acc[1][mt][v] = acc[0][mt][v]; acc[0][mt][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        // plane q+1 (requested a plane ago) -> the other buffer; request q+2
        if (t + 1 < T) { stage(slab + ((t + 1) & 1) * SLAB); issue(q + 2); }
        __syncthreads();
    }
}

}  // namespace

int mvs_conv3d_k8_launch(const ConvArgs& a0, int Cout, hipStream_t st) {
    if (a0.xs || a0.x2 || a0.bn.stats || a0.stats || a0.cout_total != Cout) return MVS_E_SHAPE;   // plain convolution only
    if (Cout != 32) return MVS_E_SHAPE;
    if ((long long)a0.D * a0.H * a0.W * 32 * 4 >= (1LL << 31)) return MVS_E_SHAPE;
    ConvArgs a = a0;
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TWK - 1) / TWK);
    a.planes_per_wg = conv_pick_planes(a.D, tiles, 2, 512);
    dim3 grid(tiles, 1, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    const size_t smem = (size_t)(3 * NGRP * 16 * 32 + 2 * SLAB) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3d_k8_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv3d_k8_kernel<32><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}
