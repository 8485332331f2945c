// fp32-MFMA stride-2 3x3x3 transposed convolution (decoder layers 3dconv4_0/5_0/6_0,
// mvsnet/cnn_wrapper/mvsnetworks.py:146-154; tf.layers.conv3d_transpose SAME, network.py:327).
//
// out[2i + k] += in[i] * W[k][co][ci] per axis, cropped to [0, 2n).  Per axis an even output 2m
// receives (i = m, k = 0) and (i = m-1, k = 2), an odd output 2m+1 receives (i = m, k = 1); so the
// 8 output parity classes of a 2x2x2 cell are small convolutions over the coarse input.  The kernel
// marches coarse input planes (staged once in LDS with a -1 halo in h and w): plane q adds its
// kd = 0 taps to the carried even plane 2q (which already holds kd = 2 of plane q-1) and stores it,
// produces the odd plane 2q+1 from kd = 1, and starts the next even plane from kd = 2.
// GEMM roles as in conv3d_mfma.hip: rows = cout, columns = 16 coarse positions along w, K = ci;
// in-plane parity classes (kh&1, kw&1) select the accumulator, taps with k = 2 read the (-1) shifted
// position.
#include "conv_common.h"
#include <cstdlib>

namespace {

constexpr int TW = CONV_TW;
constexpr int PW = TW + 1;              // staged row width (one halo column on the left)

template <int CIN, int COUT, int TH> struct DeconvGeom {
    static constexpr int LDS_BYTES = (27 * CIN * COUT + 2 * (TH + 1) * (CONV_TW + 1) * SlabGeom<CIN>::S) * 4;
    static constexpr int WGS_PER_CU = (2 * LDS_BYTES <= 160 * 1024) ? 2 : 1;
};

template <int CIN, int COUT, int TH, bool HAS_X2>
__global__ void __launch_bounds__(256, (DeconvGeom<CIN, COUT, TH>::WGS_PER_CU))
deconv3d_kernel(ConvArgs a) {
    constexpr int S = SlabGeom<CIN>::S;
    constexpr int NPOS = (TH + 1) * PW;
    constexpr int CQ = CIN / 4;
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    constexpr int V = TH / 4;
    constexpr int WROW = COUT * 4;                 // floats per (tap, ci-quad) group
    constexpr int W_FLOATS = 27 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                              // [27 taps][CQ][COUT][4]
    float* slab = smem + W_FLOATS;                 // [2][NPOS][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;

    const int tiles_w = (a.W + TW - 1) / TW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TW;
    const int co_base = blockIdx.y * COUT;
    const int q0 = blockIdx.z * a.planes_per_wg;
    const int q1 = min(q0 + a.planes_per_wg, a.D);
    const int T = q1 - q0 + 1;                     // coarse planes q0-1 .. q1-1
    const int Ho = 2 * a.H, Wo = 2 * a.W;

    // weights (kd,kh,kw,Cout,Cin) -> LDS [tap][ci/4][co][ci%4]
    if (a.wprep) load_prepared_weights(wl, a.wprep, W_FLOATS);
    else for (int i = tid; i < W_FLOATS; i += 256) {
        int j = i & 3;
        int co = (i >> 2) % COUT;
        int g = (i >> 2) / COUT;
        int ciq = g % CQ, tap = g / CQ;
        wl[i] = a.w[((size_t)tap * a.cout_total + co_base + co) * CIN + ciq * 4 + j];
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
        int r = pos / PW, c = pos - r * PW;
        int gh = h0 - 1 + r, gw = w0 - 1 + c;
        bool inb = (f < NF4) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? (gh * a.W + gw) * CIN + 4 * c4 : -1;
        loff[i] = (f < NF4) ? (pos) * S + 4 * c4 : -1;
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

    // accumulators: [plane block: 0 = carried even plane, 1 = odd plane, 2 = next even plane]
    //               [in-plane parity class 2*(oh&1) + (ow&1)][voxel tile]
    f32x4 acc[3][4][V];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[b][c][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = ((V * wave + v + 1) * PW + n + 1) * S + 4 * kq;
    const int a_lane = (kq * COUT + (n % COUT)) * 4;

    // all 9 in-plane taps of depth tap KD into plane block PB (PB == KD: 0 -> even, 1 -> odd, 2 -> next)
    auto sweep_kd = [&](auto Kc, const float* buf) __attribute__((always_inline)) {
        constexpr int KD = decltype(Kc)::value;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                constexpr int dummy = 0; (void)dummy;
                const int cls = (kh & 1) * 2 + (kw & 1);
                const int shift = -((kh == 2) ? PW : 0) - ((kw == 2) ? 1 : 0);     // k = 2 reads i = m-1
                const int tap = (KD * 3 + kh) * 3 + kw;
#pragma unroll
                for (int s = 0; s < CIN / 16; ++s) {
                    f32x4 bv[V];
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        bv[v] = *(const f32x4*)(buf + b_off[v] + shift * S + 16 * s);
                    f32x4 av = *(const f32x4*)(wl + a_lane + (tap * CQ + 4 * s) * WROW);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            acc[KD][cls][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[v][j], acc[KD][cls][v], 0, 0, 0);
                }
            }
        }
    };

    const bool lane_has_rows = (4 * kq < COUT);     // COUT = 8: rows 8..15 are duplicates
    auto store_block = [&](auto Bc, int od, int q) __attribute__((always_inline)) {
        constexpr int PB = decltype(Bc)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                int h = h0 + V * wave + v, w = w0 + n;
                if (lane_has_rows && h < a.H && w < a.W) {
                    int oh = 2 * h + (c >> 1), ow = 2 * w + (c & 1);
                    f32x4 r = acc[PB][c][v];
                    float* dst = a.y + ((((size_t)od * Ho + oh) * Wo) + ow) * a.cout_total + co_base + 4 * kq;
                    *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
                }
            }
        }
        (void)q;
    };

    issue_loads(q0 - 1);
    write_slab(q0 - 1, slab);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int q = q0 - 1 + t;
        float* cur = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
        const bool more = (t + 1 < T);
        if (more) issue_loads(q + 1);
        if (q >= 0) {
            if (q >= q0) {
                sweep_kd(std::integral_constant<int, 0>{}, cur);
                sweep_kd(std::integral_constant<int, 1>{}, cur);
            }
            if (q < q1 - 1) sweep_kd(std::integral_constant<int, 2>{}, cur);
        }
        if (q >= q0) {
            store_block(std::integral_constant<int, 0>{}, 2 * q, q);
            store_block(std::integral_constant<int, 1>{}, 2 * q + 1, q);
        }
        // rotate: next even plane becomes the carried one
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[0][c][v] = acc[2][c][v];
                acc[1][c][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[2][c][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        if (more) write_slab(q + 1, nxt);
        __syncthreads();
    }

    if (a.stats) stats_commit<COUT>(st_s, st_q, false, slab, conv_stats_row(a), a.cout_total, co_base);
}

template <int CIN, int COUT, int TH>
int launch_deconv(const ConvArgs& a0, int Cout, hipStream_t st) {
    ConvArgs a = a0;
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    const int groups = Cout / COUT;
    a.planes_per_wg = conv_pick_planes(a.D, (long long)tiles * groups, 1);
    dim3 grid(tiles, groups, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    size_t smem = (size_t)(27 * CIN * COUT + 2 * (TH + 1) * PW * SlabGeom<CIN>::S) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)deconv3d_kernel<CIN, COUT, TH, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)deconv3d_kernel<CIN, COUT, TH, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (a.x2) deconv3d_kernel<CIN, COUT, TH, true><<<grid, 256, smem, st>>>(a);
    else deconv3d_kernel<CIN, COUT, TH, false><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_deconv3d_mfma_launch(const ConvArgs& a, int Cin, int Cout, hipStream_t st) {
    // (weights prepared for the block kernels carry THEIR layout, mvs_conv_weight_layout: a layer this test sends there must
    //  not fall through to the kernels below with that array -- a.wprep is only ever built for the kernel chosen here)
    if (mvs_conv3d_os_covers(2, Cin, Cout)) return mvs_conv3d_os_launch(a, 2, Cin, Cout, st);
    if (Cout % 8 == 0 && ((Cin == 16 && Cout == 8) || Cin == 64)) {
        int rc = mvs_deconv3d_c8_launch(a, Cin, Cout, st);       // packed 8-channel kernel (deconv3d_c8.hip)
        if (rc != MVS_E_SHAPE) return rc;
    }
    if (Cin == 16 && Cout == 8) return launch_deconv<16, 8, 8>(a, Cout, st);
    if (Cin == 32 && Cout % 16 == 0) return launch_deconv<32, 16, 8>(a, Cout, st);
    if (Cin == 64 && Cout % 8 == 0) return launch_deconv<64, 8, 8>(a, Cout, st);
    if (Cin == 16 && Cout % 16 == 0) return launch_deconv<16, 16, 8>(a, Cout, st);
    return MVS_E_SHAPE;
}
