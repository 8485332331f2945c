// Output convolution 3dconv6_2: Cin (= base_filter) -> 1 channel, 3x3x3 SAME stride 1, no BN / ReLU /
// bias (mvsnet/cnn_wrapper/mvsnetworks.py:156-158), consuming BN+ReLU(3dconv6_0) + BN+ReLU(3dconv0_1)
// on load (the additive skip of :156-157).
//
// One output channel cannot fill an MFMA tile (1 of 16 rows), and the layer is HBM-bound anyway:
// 216 MAC per voxel against 2 x Cin x 4 bytes read.  So: VALU kernel, same input-stationary plane
// march as the MFMA kernels -- a workgroup owns an 8 x 32 (h x w) column, stages each normalised
// input plane once in LDS (12-float positions: conflict-free 16-B reads for consecutive lanes) and
// each thread adds the plane's 9 in-plane taps into the three output planes q+1, q, q-1.
// Roofline: HBM, 2*Cin*4 + 4 bytes per voxel.
#include "conv_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int OTH = 8, OTW = 32;
constexpr int OPW = OTW + 2;

template <int CIN>
__global__ void __launch_bounds__(256, 3)     // 3 workgroups per CU: HBM-latency bound, needs waves in flight
conv3d_out_kernel(ConvArgs a) {
    constexpr int S = CIN + 4;
    constexpr int CQ = CIN / 4;
    constexpr int NPOS = (OTH + 2) * OPW;
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");
    __shared__ __attribute__((aligned(16))) float slab[2][NPOS * S];
    // the 27*CIN weights are read through the scalar cache (constant address space, wave-uniform
    // addresses -> s_load into SGPRs): as LDS broadcast reads they took 3/4 of the LDS pipe
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat* wsh = (cfloat*)(a.w);

    const int tid = threadIdx.x;
    const int row = tid >> 5, col = tid & 31;
    const int tiles_w = (a.W + OTW - 1) / OTW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * OTH, w0 = tile_w * OTW;
    const int d0 = blockIdx.z * a.planes_per_wg;
    const int d1 = min(d0 + a.planes_per_wg, a.D);
    const int T = d1 - d0 + 2;
    // Even depth chunks march forwards, odd ones BACKWARDS: the chunks z and z + 1 of a tile (all of a tile's chunks run at the
    // same time and, the tile count being a multiple of 8, on the same XCD) then reach their two shared halo planes at the
    // same moment -- at the end of both marches, or at the start of both -- and the second reader finds them in that XCD's L2
    // instead of fetching them from HBM again (the depth halo was 23 of the layer's 304 MB of reads at the metric workload).
    const bool fwd = (blockIdx.z & 1) == 0;

    const int c4 = tid % CQ;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sc2 = sc, sh2 = sh;
    const bool has_x2 = a.x2 != nullptr;
    const bool has_aff = a.xs != nullptr || a.bn.stats != nullptr;
    if (a.xs) { sc = *(const float4*)(a.xs + 4 * c4); sh = *(const float4*)(a.xb + 4 * c4); }
    else if (a.bn.stats) bn_affine4(a.bn, 4 * c4, sc, sh);
    const bool has_aff2 = has_x2 && (a.x2s != nullptr || a.bn2.stats != nullptr);
    if (has_x2 && a.x2s) { sc2 = *(const float4*)(a.x2s + 4 * c4); sh2 = *(const float4*)(a.x2b + 4 * c4); }
    else if (has_x2 && a.bn2.stats) bn_affine4(a.bn2, 4 * c4, sc2, sh2);

    float4 pre[NIT], pre2[NIT];
    // Per-thread staging map, identical for every plane: element offset inside one input plane
    // (-1 = outside the volume -> SAME padding zero) and float offset inside the LDS slab (-1 = none).
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int f = tid + 256 * i;
        int pos = f / CQ;
        int r = pos / OPW, c = pos - r * OPW;
        int gh = h0 - 1 + r, gw = w0 - 1 + c;
        bool inb = (f < NF4) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? (gh * a.W + gw) * CIN + 4 * c4 : -1;
        loff[i] = (f < NF4) ? (pos) * S + 4 * c4 : -1;
    }
    const size_t plane_elems = (size_t)a.H * a.W * CIN;

    auto issue_loads = [&](int q) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
        const float* px = a.x + (size_t)(plane_ok ? q : 0) * plane_elems;
        const float* px2 = has_x2 ? a.x2 + (size_t)(plane_ok ? q : 0) * plane_elems : nullptr;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = plane_ok && goff[i] >= 0;
            pre[i] = ok ? *(const float4*)(px + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            pre2[i] = (ok && has_x2) ? *(const float4*)(px2 + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto write_slab = [&](int q, float* buf) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (loff[i] < 0) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plane_ok && goff[i] >= 0) {             // SAME padding pads the NORMALISED input with 0
                // BN + ReLU (+ skip) on channel pairs: v_pk_fma_f32 / v_pk_max_f32 / v_pk_add_f32
                const f32x2 z2 = {0.f, 0.f};
                f32x2 lo = {pre[i].x, pre[i].y}, hi = {pre[i].z, pre[i].w};
                if (has_aff) {
                    lo = __builtin_elementwise_max(lo * (f32x2){sc.x, sc.y} + (f32x2){sh.x, sh.y}, z2);
                    hi = __builtin_elementwise_max(hi * (f32x2){sc.z, sc.w} + (f32x2){sh.z, sh.w}, z2);
                }
                if (has_x2) {
                    f32x2 lo2 = {pre2[i].x, pre2[i].y}, hi2 = {pre2[i].z, pre2[i].w};
                    if (has_aff2) {
                        lo2 = __builtin_elementwise_max(lo2 * (f32x2){sc2.x, sc2.y} + (f32x2){sh2.x, sh2.y}, z2);
                        hi2 = __builtin_elementwise_max(hi2 * (f32x2){sc2.z, sc2.w} + (f32x2){sh2.z, sh2.w}, z2);
                    }
                    lo += lo2; hi += hi2;
                }
                v = make_float4(lo[0], lo[1], hi[0], hi[1]);
            }
            *(float4*)(buf + loff[i]) = v;
        }
    };

    // accN: the output plane one step AHEAD of the input plane q in march direction, accM: plane q, accO: one step behind
    // (complete after this input plane).  Forwards that is kd = 0 / 1 / 2, backwards kd = 2 / 1 / 0.
    f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
    const int base = (row * OPW + col) * S;
    const int step = fwd ? 1 : -1, q0 = fwd ? d0 - 1 : d1;
    const int kdN = fwd ? 0 : 2, kdO = fwd ? 2 : 0;

    issue_loads(q0);
    write_slab(q0, slab[0]);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int q = q0 + step * t;
        const float* cur = slab[t & 1];
        const bool more = (t + 1 < T);
        if (more) issue_loads(q + step);
        if (q >= 0 && q < a.D) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int cq = 0; cq < CQ; ++cq) {
                        float4 x = *(const float4*)(cur + base + (kh * OPW + kw) * S + 4 * cq);
                        cfloat* w0p = wsh + ((kdN * 9 + kh * 3 + kw) * CIN + 4 * cq);     // (3,3,3,Cin,1)
                        cfloat* w1p = wsh + ((1 * 9 + kh * 3 + kw) * CIN + 4 * cq);
                        cfloat* w2p = wsh + ((kdO * 9 + kh * 3 + kw) * CIN + 4 * cq);
                        // channel pairs: v_pk_fma_f32 with the weight pair straight from SGPRs (even / odd channel sums)
                        const f32x2 xa = {x.x, x.y}, xb = {x.z, x.w};
                        acc0 += xa * (f32x2){w0p[0], w0p[1]}; acc0 += xb * (f32x2){w0p[2], w0p[3]};
                        acc1 += xa * (f32x2){w1p[0], w1p[1]}; acc1 += xb * (f32x2){w1p[2], w1p[3]};
                        acc2 += xa * (f32x2){w2p[0], w2p[1]}; acc2 += xb * (f32x2){w2p[2], w2p[3]};
                    }
        }
        const int o = q - step, h = h0 + row, w = w0 + col;
        if (o >= d0 && o < d1 && h < a.H && w < a.W)
            a.y[(((size_t)o * a.H + h) * a.W) + w] = acc2[0] + acc2[1];
        acc2 = acc1; acc1 = acc0; acc0 = (f32x2){0.f, 0.f};
        if (more) write_slab(q + step, slab[(t + 1) & 1]);
        __syncthreads();
    }
}

template <int CIN>
int launch_out(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    const int tiles = ((a.H + OTH - 1) / OTH) * ((a.W + OTW - 1) / OTW);
    a.planes_per_wg = conv_pick_planes(a.D, tiles, 2, 768);
    dim3 grid(tiles, 1, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    conv3d_out_kernel<CIN><<<grid, 256, 0, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_conv3d_out_launch(const ConvArgs& a, int Cin, hipStream_t st) {
    if (a.stats) return MVS_E_SHAPE;               // the output layer has no BatchNorm
    if (Cin == 8) return launch_out<8>(a, st);
    if (Cin == 4) return launch_out<4>(a, st);
    if (Cin == 16) return launch_out<16>(a, st);
    return MVS_E_SHAPE;
}
