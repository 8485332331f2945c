// LAB VARIANT (not in libmvsnet_hip.so; built by `python -m mvsnet_amd.build --lab` into variants/libmvsnet_lab.so, tested by
// tests/test_gpu_lab.py): the LDS-staged warp + variance sweep (C = 32) that north_star describes.  Exact, measured no faster
// than the product's register tap cache (188 us against 186-190 us at the metric workload, round 2).
// north_star's shape of the kernel: for a tile of reference pixels and a run of depth planes, the footprint of the
// tile in every source view is staged into LDS ONCE (wide coalesced loads, zeros outside the image -> the per-tap zero
// fill of ImageProjectiveTransform needs no weight masking), and the sweep takes its four taps per view as ds_read_b128
// from there instead of re-fetching them through the texture-address path.
//   block  = 256 lanes = a 4 x 8 pixel tile x 8 lanes (4 channels each) x 8 planes;
//   box    = per view the bounding box of floor(sample point) over the tile and the plane run, + the tap column / row: a
//            projective map takes the tile to a convex quadrilateral and a pixel's sample point moves monotonically along
//            its epipolar line with 1/depth, so the extremes sit at the 4 tile corners of the first and last plane.  Every
//            wave computes the box itself (8 corner samples per view in its lanes, shuffle min / max): the block's only
//            barrier is the one after staging;
//   exact by construction: the bookkeeping lane of every (pixel, plane) checks that its taps lie inside the staged box; a
//            wave whose check fails -- or a block whose box exceeds the budget (wide baselines, tiny volumes) -- takes
//            the direct per-tap path (warp_sample) for its planes; mvs_cost_volume_fallback_rounds counts them;
//   no table in LDS: lane s of a pixel's 8 lanes keeps the bookkeeping of plane s (LDS byte offset of tap 00, the two
//            bilinear fractions) in registers and hands it round with ds_bpermute, so a block holds only the boxes
//            (CAP positions x 128 B per view): 3 workgroups = 12 waves per CU.
// Measured (round 2, metric workload): 188 us, the same as the register tap cache below (186-190 us); a first form with
// a table in LDS, 16 planes per block and LDS atomics for the box ran 227 us.  Counters (rocprofv3 --pmc): 73 M VALU
// instructions (the register cache: 70 M -- the blend, not the tap bookkeeping, is the VALU cost), LDS array busy 254 k
// cycles per CU = 106 us (16 tap reads + 12 bpermutes per lane and plane, staging writes), TA almost idle.  At fp32 x 32
// channels the three pipes this form needs -- vector ALU ~90 us, LDS ~106 us, HBM write 80 us -- are each near the kernel
// time of the register form, so moving the taps from the TA path to LDS buys nothing; it stays opt-in and tested
// (tests/test_gpu_parity.py::test_lds_staged_cost_volume_matches_the_register_cache_kernel).  At configs c2 / c3 the near
// planes move 0.6 / 1.4 pixels per plane and part / most of the blocks exceed the box budget.
#include "../common.h"
#include <climits>
#include <cstdlib>

namespace {
#include "../cost_volume_common.h"

// north_star's shape of the kernel: for a tile of reference pixels and a run of depth planes, the footprint of the
// tile in every source view is staged into LDS ONCE (wide coalesced loads, zeros outside the image -> the per-tap zero
// fill of ImageProjectiveTransform needs no weight masking), and the sweep takes its four taps per view as ds_read_b128
// from there instead of re-fetching them through the texture-address path.
//   block  = 256 lanes = a 4 x 8 pixel tile x 8 lanes (4 channels each) x 8 planes;
//   box    = per view the bounding box of floor(sample point) over the tile and the plane run, + the tap column / row: a
//            projective map takes the tile to a convex quadrilateral and a pixel's sample point moves monotonically along
//            its epipolar line with 1/depth, so the extremes sit at the 4 tile corners of the first and last plane.  Every
//            wave computes the box itself (8 corner samples per view in its lanes, shuffle min / max): the block's only
//            barrier is the one after staging;
//   exact by construction: the bookkeeping lane of every (pixel, plane) checks that its taps lie inside the staged box; a
//            wave whose check fails -- or a block whose box exceeds the budget (wide baselines, tiny volumes) -- takes
//            the direct per-tap path (warp_sample) for its planes; mvs_cost_volume_fallback_rounds counts them;
//   no table in LDS: lane s of a pixel's 8 lanes keeps the bookkeeping of plane s (LDS byte offset of tap 00, the two
//            bilinear fractions) in registers and hands it round with ds_bpermute, so a block holds only the boxes
//            (CAP positions x 128 B per view): 3 workgroups = 12 waves per CU.
// Measured (round 2, metric workload): 188 us, the same as the register tap cache below (186-190 us); a first form with
// a table in LDS, 16 planes per block and LDS atomics for the box ran 227 us.  Counters (rocprofv3 --pmc): 73 M VALU
// instructions (the register cache: 70 M -- the blend, not the tap bookkeeping, is the VALU cost), LDS array busy 254 k
// cycles per CU = 106 us (16 tap reads + 12 bpermutes per lane and plane, staging writes), TA almost idle.  At fp32 x 32
// channels the three pipes this form needs -- vector ALU ~90 us, LDS ~106 us, HBM write 80 us -- are each near the kernel
// time of the register form, so moving the taps from the TA path to LDS buys nothing; it stays opt-in and tested
// (tests/test_gpu_parity.py::test_lds_staged_cost_volume_matches_the_register_cache_kernel).  At configs c2 / c3 the near
// planes move 0.6 / 1.4 pixels per plane and part / most of the blocks exceed the box budget.
constexpr int CVL_TH = 4, CVL_TW = 8;      // tile of the LDS-staged sweep

__device__ int g_cvl_fallback_rounds;      // waves that took the direct path since the last reset

template <int NSRC, int CAP>
__global__ void __launch_bounds__(256, 3)
cost_volume_lds2_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                        const float* __restrict__ transforms, int depth_total, int d_begin, int d_count,
                        int H, int W, int variant, int negate, int tiles_x, float* __restrict__ cost) {
    constexpr int C = 32, LP = 8;
    extern __shared__ __attribute__((aligned(16))) float4 cvl_smem[];
    float4* box = cvl_smem;                                        // [NSRC][CAP positions][8 float4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 7, pixw = lane >> 3;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * CVL_TW + pixw, y = ty * CVL_TH + wave;
    const bool live = x < W && y < H;
    const float xf = (float)min(x, W - 1), yf = (float)min(y, H - 1);
    const int dl0 = blockIdx.y * LP, dl1 = min(dl0 + LP, d_count);
    const int c = sub * 4;

    auto sample_t = [&](const float4 ta, const float4 tb, float px, float py, float& sx, float& sy) __attribute__((always_inline)) {
        const float proj = tb.z * px + tb.w * py + 1.0f;
        const float inv = __builtin_amdgcn_rcpf(proj);
        sx = (ta.x * px + ta.y * py + ta.z) * inv;
        sy = (ta.w * px + tb.x * py + tb.y) * inv;
    };
    // transforms of this lane's bookkeeping plane (dl0 + sub), all views: in flight under the box computation
    float4 bta[NSRC], btb[NSRC];
    {
        const int dmy = d_begin + min(dl0 + sub, dl1 - 1);
#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const float* t = transforms + ((size_t)v * depth_total + dmy) * 8;
            bta[v] = ld4(t); btb[v] = ld4(t + 4);
        }
    }
    // ---- box of every view, computed by every wave: lane = (view, corner) for lane < 8 * NSRC -------------------------
    int bx0[NSRC], by0[NSRC], bw[NSRC], bh[NSRC];
    bool fits = true;
    {
        const int v = min(lane >> 3, NSRC - 1), k = lane & 7;
        const float cx = (float)((k & 1) ? min(tx * CVL_TW + CVL_TW - 1, W - 1) : tx * CVL_TW);
        const float cy = (float)((k & 2) ? min(ty * CVL_TH + CVL_TH - 1, H - 1) : ty * CVL_TH);
        const float* t = transforms + ((size_t)v * depth_total + d_begin + ((k & 4) ? dl1 - 1 : dl0)) * 8;
        float sx, sy;
        sample_t(ld4(t), ld4(t + 4), cx, cy, sx, sy);
        int lox = (int)floorf(sx), loy = (int)floorf(sy), hix = lox, hiy = loy;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            lox = min(lox, __shfl_xor(lox, o, 64)); loy = min(loy, __shfl_xor(loy, o, 64));
            hix = max(hix, __shfl_xor(hix, o, 64)); hiy = max(hiy, __shfl_xor(hiy, o, 64));
        }
#pragma unroll
        for (int vv = 0; vv < NSRC; ++vv) {
            bx0[vv] = __builtin_amdgcn_readlane(lox, vv * 8); by0[vv] = __builtin_amdgcn_readlane(loy, vv * 8);
            const long long w = (long long)__builtin_amdgcn_readlane(hix, vv * 8) - bx0[vv] + 2;
            const long long h = (long long)__builtin_amdgcn_readlane(hiy, vv * 8) - by0[vv] + 2;
            fits = fits && w >= 2 && h >= 2 && w <= CAP && h <= CAP && w * h <= CAP &&
                   bx0[vv] > -(1 << 24) && by0[vv] > -(1 << 24) && bx0[vv] < (1 << 24) && by0[vv] < (1 << 24);
            bw[vv] = (int)w; bh[vv] = (int)h;
        }
    }
    // ---- stage the boxes ---------------------------------------------------------------------------------------------
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, NSRC * H * W * C * 4, 0x00020000);
    if (fits) {
        constexpr int NIT = (CAP * 8 + 255) / 256;
        u32x4_t st[NSRC][NIT];
#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const int npos = bw[v] * bh[v];
            const int m = (65536 + bw[v] - 1) / bw[v];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pos = (tid + 256 * it) >> 3;
                const int r = (pos * m) >> 16, cc = pos - r * bw[v];
                const int gy = by0[v] + r, gx = bx0[v] + cc;
                const bool ok = pos < npos && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                st[v][it] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, ok ? ((v * H + gy) * W + gx) * (C * 4) + sub * 16 : (int)0x80000000u, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < NSRC; ++v)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = tid + 256 * it;
                if (i < bw[v] * bh[v] * 8)
                    box[v * CAP * 8 + i] = make_float4(__uint_as_float(st[v][it][0]), __uint_as_float(st[v][it][1]),
                                                       __uint_as_float(st[v][it][2]), __uint_as_float(st[v][it][3]));
            }
    }
    // ---- bookkeeping of plane dl0 + sub (registers), containment of its taps ------------------------------------------------
    int boff[NSRC]; float bfx[NSRC], bfy[NSRC];
    bool bad = !fits;
#pragma unroll
    for (int v = 0; v < NSRC; ++v) {
        float sx, sy;
        sample_t(bta[v], btb[v], xf, yf, sx, sy);
        const float x0 = floorf(sx), y0 = floorf(sy);
        const int rx = (int)x0 - bx0[v], ry = (int)y0 - by0[v];
        bad = bad || rx < 0 || ry < 0 || rx + 1 >= bw[v] || ry + 1 >= bh[v];
        boff[v] = (v * CAP + ry * bw[v] + rx) * 128 + sub * 0;
        bfx[v] = sx - x0; bfy[v] = sy - y0;
    }
    __syncthreads();

    const long long pix = (long long)min(y, H - 1) * W + min(x, W - 1);
    const float4 r4 = ld4(ref + (size_t)pix * C + c);
    const f32x2 rr0 = (f32x2){r4.x, r4.y}, rr1 = (f32x2){r4.z, r4.w};
    const f32x2 rq0 = rr0 * rr0, rq1 = rr1 * rr1;
    const float n = (float)(NSRC + 1);
    const float inv_n = 1.0f / n, inv_nn = 1.0f / (n * n);
    const char* boxb = reinterpret_cast<const char*>(box) + sub * 16;
    float* dst0 = cost + (size_t)pix * C + c;
    const size_t plane_stride = (size_t)H * W * C;
    auto finish = [&](int dl, f32x2 S0, f32x2 S1, f32x2 Q0, f32x2 Q1) __attribute__((always_inline)) {
        f32x2 o0, o1;
        if (variant == 0) { o0 = Q0 * inv_n - (S0 * S0) * inv_nn; o1 = Q1 * inv_n - (S1 * S1) * inv_nn; }
        else { const f32x2 m0 = S0 * inv_n, m1 = S1 * inv_n; o0 = Q0 * inv_n - m0 * m0; o1 = Q1 * inv_n - m1 * m1; }
        if (negate) { o0 = -o0; o1 = -o1; }
        if (live && dl < dl1) *reinterpret_cast<float4*>(dst0 + (size_t)dl * plane_stride) = make_float4(o0[0], o0[1], o1[0], o1[1]);
    };
    if (!__any(bad)) {
        const int grp = lane & ~7;
#pragma unroll
        for (int p = 0; p < LP; ++p) {
            f32x2 S0 = rr0, S1 = rr1, Q0 = rq0, Q1 = rq1;
            float4 t00[NSRC], t01[NSRC], t10[NSRC], t11[NSRC];
            float fx[NSRC], fy[NSRC];
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const int off = __shfl(boff[v], grp | p, 64);
                fx[v] = __shfl(bfx[v], grp | p, 64); fy[v] = __shfl(bfy[v], grp | p, 64);
                const char* a0 = boxb + off;
                const char* a1 = a0 + bw[v] * 128;
                t00[v] = *reinterpret_cast<const float4*>(a0); t01[v] = *reinterpret_cast<const float4*>(a0 + 128);
                t10[v] = *reinterpret_cast<const float4*>(a1); t11[v] = *reinterpret_cast<const float4*>(a1 + 128);
            }
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float gx = 1.0f - fx[v], gy = 1.0f - fy[v];
                const float w00 = gy * gx, w01 = gy * fx[v], w10 = fy[v] * gx, w11 = fy[v] * fx[v];
                const f32x2 w0 = w00 * (f32x2){t00[v].x, t00[v].y} + w01 * (f32x2){t01[v].x, t01[v].y} +
                                 w10 * (f32x2){t10[v].x, t10[v].y} + w11 * (f32x2){t11[v].x, t11[v].y};
                const f32x2 w1 = w00 * (f32x2){t00[v].z, t00[v].w} + w01 * (f32x2){t01[v].z, t01[v].w} +
                                 w10 * (f32x2){t10[v].z, t10[v].w} + w11 * (f32x2){t11[v].z, t11[v].w};
                S0 += w0; S1 += w1; Q0 += w0 * w0; Q1 += w1 * w1;
            }
            finish(dl0 + p, S0, S1, Q0, Q1);
        }
    } else {
        if (lane == 0) atomicAdd(&g_cvl_fallback_rounds, 1);
        for (int p = 0; p < dl1 - dl0; ++p) {
            f32x2 S0 = rr0, S1 = rr1, Q0 = rq0, Q1 = rq1;
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float4 wv = warp_sample<0>(src + (size_t)v * H * W * C, transforms + ((size_t)v * depth_total + d_begin + dl0 + p) * 8,
                                                 xf, yf, H, W, C, c);
                const f32x2 w0 = (f32x2){wv.x, wv.y}, w1 = (f32x2){wv.z, wv.w};
                S0 += w0; S1 += w1; Q0 += w0 * w0; Q1 += w1 * w1;
            }
            finish(dl0 + p, S0, S1, Q0, Q1);
        }
    }
}

template <int NSRC>
int launch_lds2_sweep(const float* ref, const float* src, const float* transforms, int depth_total,
                      int d_begin, int d_count, int H, int W, int variant, int negate, float* cost, hipStream_t st) {
    constexpr int CAP = 96;
    const int tiles_x = mvs_cdiv(W, CVL_TW), tiles_y = mvs_cdiv(H, CVL_TH);
    dim3 grid(tiles_x * tiles_y, mvs_cdiv(d_count, 8));
    const size_t smem = (size_t)(NSRC * CAP * 8) * sizeof(float4);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)cost_volume_lds2_kernel<NSRC, CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    cost_volume_lds2_kernel<NSRC, CAP><<<grid, 256, smem, st>>>(ref, src, transforms, depth_total, d_begin, d_count, H, W,
                                                                variant, negate, tiles_x, cost);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mvs_lab_cost_volume_lds_f32(const float* ref, const float* src, const float* transforms, int view_num,
                                           int depth_total, int d_begin, int d_count, int H, int W, int C, int variant,
                                           int negate, float* cost, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && cost);
    MVS_CHECK_ARG(view_num >= 2 && view_num <= 8 && depth_total >= 1 && d_begin >= 0 && d_count >= 1 &&
                  d_begin + d_count <= depth_total && H > 0 && W > 0);
    if (C != 32 || (long long)(view_num - 1) * H * W * C * 4 >= (1LL << 31)) return MVS_E_SHAPE;
    hipStream_t st = mvs_stream(stream);
    switch (view_num - 1) {
#define MVS_LSWEEP2(NS) case NS: return launch_lds2_sweep<NS>(ref, src, transforms, depth_total, d_begin, d_count, H, W, variant, negate, cost, st);
        MVS_LSWEEP2(1) MVS_LSWEEP2(2) MVS_LSWEEP2(3) MVS_LSWEEP2(4) MVS_LSWEEP2(5) MVS_LSWEEP2(6) MVS_LSWEEP2(7)
#undef MVS_LSWEEP2
    }
    return MVS_E_SHAPE;
}

// Rounds of 8 planes (per wave) of the LDS-staged sweep that took the direct path since the last call (a geometry whose
// footprints do not fit the LDS budget still gives exact results, only slower).  Synchronises.
extern "C" int mvs_lab_cost_volume_fallback_rounds(int* rounds) {
    MVS_CHECK_ARG(rounds);
    int zero = 0;
    hipError_t e = hipMemcpyFromSymbol(rounds, HIP_SYMBOL(g_cvl_fallback_rounds), sizeof(int));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_cvl_fallback_rounds), &zero, sizeof(int));
    return (int)e;
}
