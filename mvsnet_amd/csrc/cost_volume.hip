// Fused projective bilinear warp + cross-view variance -> (D,H,W,C) cost volume (R2 + R3).
// Reference behaviour: tf.contrib.image.transform BILINEAR with zero fill per tap
// (mvsnet/homography_warping.py:251-252) inside the cost loops of mvsnet/model.py:422-463
// (inference_mem), :315-334 (inference) and :680-693 (GRU body).
//
// Roofline: HBM write of the volume (D*H*W*C*4 bytes) + one read of the N feature maps.
// Layout: channel-last; one lane owns 4 consecutive channels (16 B) of one voxel, so the C/4
// lanes of a voxel read each bilinear tap as one contiguous C*4-byte segment and the block's
// stores are fully coalesced 16-B-per-lane rows of the volume.  Feature maps (N*H*W*C*4 bytes,
// 13 MB at the metric config) stay resident in L2 / Infinity Cache across the D planes.
#include "common.h"
#include <climits>
#include <cstdlib>

int mvs_cost_volume_mfma_launch(const float* ref, const float* src, const float* transforms, int n_src, int depth_total,
                                int d_begin, int d_count, int H, int W, int variant, int negate, float* cost,
                                hipStream_t st);      // cost_volume_mfma.hip

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_nt __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename R>
__device__ __forceinline__ float4 ldb(R rsrc, int byte_off) {      // buffer_load_dwordx4 ... offen
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

__device__ __forceinline__ float4 fma4(float w, float4 a, float4 acc) {
    acc.x += w * a.x; acc.y += w * a.y; acc.z += w * a.z; acc.w += w * a.w;
    return acc;
}

// Bilinear sample of `img` (H,W,C) at the projective image of pixel (x,y); channels [c, c+4).
template <int BORDER>
__device__ __forceinline__ float4 warp_sample(const float* __restrict__ img, const float* __restrict__ t,
                                              float xf, float yf, int H, int W, int C, int c) {
    float proj = t[6] * xf + t[7] * yf + 1.0f;
    float sx = (t[0] * xf + t[1] * yf + t[2]) / proj;
    float sy = (t[3] * xf + t[4] * yf + t[5]) / proj;
    float x0 = floorf(sx), y0 = floorf(sy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BORDER == 0) {
        // zero fill: each tap individually reads 0 outside [0,W)x[0,H)
        bool okx0 = (x0 >= 0.0f) && (x0 < (float)W);
        bool okx1 = (x1 >= 0.0f) && (x1 < (float)W);
        bool oky0 = (y0 >= 0.0f) && (y0 < (float)H);
        bool oky1 = (y1 >= 0.0f) && (y1 < (float)H);
        int ix0 = okx0 ? (int)x0 : 0, ix1 = okx1 ? (int)x1 : 0;
        int iy0 = oky0 ? (int)y0 : 0, iy1 = oky1 ? (int)y1 : 0;
        float4 v00 = (okx0 && oky0) ? ld4(img + ((size_t)iy0 * W + ix0) * C + c) : z;
        float4 v01 = (okx1 && oky0) ? ld4(img + ((size_t)iy0 * W + ix1) * C + c) : z;
        float4 v10 = (okx0 && oky1) ? ld4(img + ((size_t)iy1 * W + ix0) * C + c) : z;
        float4 v11 = (okx1 && oky1) ? ld4(img + ((size_t)iy1 * W + ix1) * C + c) : z;
        float wx1 = x1 - sx, wx0 = sx - x0, wy1 = y1 - sy, wy0 = sy - y0;
        float4 vf, vc, o;
        vf.x = wx1 * v00.x + wx0 * v01.x; vf.y = wx1 * v00.y + wx0 * v01.y;
        vf.z = wx1 * v00.z + wx0 * v01.z; vf.w = wx1 * v00.w + wx0 * v01.w;
        vc.x = wx1 * v10.x + wx0 * v11.x; vc.y = wx1 * v10.y + wx0 * v11.y;
        vc.z = wx1 * v10.z + wx0 * v11.z; vc.w = wx1 * v10.w + wx0 * v11.w;
        o.x = wy1 * vf.x + wy0 * vc.x; o.y = wy1 * vf.y + wy0 * vc.y;
        o.z = wy1 * vf.z + wy0 * vc.z; o.w = wy1 * vf.w + wy0 * vc.w;
        return o;
    } else {
        // clamp-to-border variant (reference dead code, homography_warping.py:140-173)
        float mx = (float)(W - 1), my = (float)(H - 1);
        float cx0 = fminf(fmaxf(x0, 0.f), mx), cx1 = fminf(fmaxf(x1, 0.f), mx);
        float cy0 = fminf(fmaxf(y0, 0.f), my), cy1 = fminf(fmaxf(y1, 0.f), my);
        int ix0 = (int)cx0, ix1 = (int)cx1, iy0 = (int)cy0, iy1 = (int)cy1;
        float4 v00 = ld4(img + ((size_t)iy0 * W + ix0) * C + c);
        float4 v01 = ld4(img + ((size_t)iy0 * W + ix1) * C + c);
        float4 v10 = ld4(img + ((size_t)iy1 * W + ix0) * C + c);
        float4 v11 = ld4(img + ((size_t)iy1 * W + ix1) * C + c);
        float wa = (cy1 - sy) * (cx1 - sx), wb = (cy1 - sy) * (sx - cx0);
        float wc = (sy - cy0) * (cx1 - sx), wd = (sy - cy0) * (sx - cx0);
        float4 o = z;
        o = fma4(wa, v00, o); o = fma4(wb, v01, o); o = fma4(wc, v10, o); o = fma4(wd, v11, o);
        return o;
    }
}

// grid: x = ceil(H*W*(C/4) / 256), y = planes.  One lane = (pixel, 4 channels) of one plane.
template <int BORDER>
__global__ void __launch_bounds__(256)
cost_volume_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                   const float* __restrict__ transforms, int n_src, int depth_total, int d_begin,
                   int H, int W, int C, int variant, int negate, float* __restrict__ cost) {
    const int cq = C >> 2;
    const long long total = (long long)H * W * cq;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int dl = blockIdx.y;            // plane index inside the output
    const int d = d_begin + dl;           // plane index inside the transform table
    int c = (int)(idx % cq) * 4;
    long long pix = idx / cq;
    int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    float xf = (float)x, yf = (float)y;

    float4 r = ld4(ref + (size_t)pix * C + c);
    float4 S = r;
    float4 Q = make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w);
    const size_t img_elems = (size_t)H * W * C;
    for (int v = 0; v < n_src; ++v) {
        const float* t = transforms + ((size_t)v * depth_total + d) * 8;   // block-uniform
        float4 w = warp_sample<BORDER>(src + v * img_elems, t, xf, yf, H, W, C, c);
        S.x += w.x; S.y += w.y; S.z += w.z; S.w += w.w;
        Q.x += w.x * w.x; Q.y += w.y * w.y; Q.z += w.z * w.z; Q.w += w.w * w.w;
    }
    const float n = (float)(n_src + 1);
    float4 o;
    if (variant == 0) {          // inference_mem: Q/N - S*S/(N*N)      (model.py:458-461)
        const float nn = n * n;
        o.x = Q.x / n - (S.x * S.x) / nn; o.y = Q.y / n - (S.y * S.y) / nn;
        o.z = Q.z / n - (S.z * S.z) / nn; o.w = Q.w / n - (S.w * S.w) / nn;
    } else {                     // inference / GRU: Q/N - (S/N)^2      (model.py:330-332)
        float ax = S.x / n, ay = S.y / n, az = S.z / n, aw = S.w / n;
        o.x = Q.x / n - ax * ax; o.y = Q.y / n - ay * ay;
        o.z = Q.z / n - az * az; o.w = Q.w / n - aw * aw;
    }
    if (negate) { o.x = -o.x; o.y = -o.y; o.z = -o.z; o.w = -o.w; }
    *reinterpret_cast<float4*>(cost + ((size_t)dl * H * W + pix) * C + c) = o;
}

// Depth-sweep variant (the 3D-CNN path): one lane = (pixel, 4 channels) sweeping DC consecutive
// planes.  Along depth a pixel's sample point slides along its epipolar line by a fraction of a
// pixel per plane (that is how plane-sweep depth intervals are chosen), so its 2x2 tap
// neighbourhood is kept in registers and re-fetched only when floor(sx) or floor(sy) changes:
// tap traffic to L1/L2 drops from 16 x 16 B per voxel-lane to a few, the kernel becomes bound by
// the coalesced HBM write of the volume.  Zero fill per tap exactly as warp_sample<0>.
template <int NSRC, int Q>       // Q float4 per lane: 4*Q channels per lane, C/(4*Q) lanes per pixel
__global__ void __launch_bounds__(256)
cost_volume_sweep_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                         const float* __restrict__ transforms, int depth_total, int d_begin,
                         int d_count, int planes_per_block, int H, int W, int C, int variant,
                         int negate, float* __restrict__ cost) {
    const int lg = C / (4 * Q);                           // lanes per pixel: power of two (host-checked)
    const long long total = (long long)H * W * lg;
    long long idx = (long long)xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int dl0 = blockIdx.y * planes_per_block;
    const int dl1 = min(dl0 + planes_per_block, d_count);
    const int sub = threadIdx.x & (lg - 1);
    const int c = sub * 4 * Q;                            // first channel of this lane
    const long long pix = idx / lg;
    const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    const float xf = (float)x, yf = (float)y;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x2 rr[2 * Q], rq[2 * Q];                           // reference feature and its square, channel pairs
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const float4 r = ld4(ref + (size_t)pix * C + c + 4 * k);
        rr[2 * k] = (f32x2){r.x, r.y}; rr[2 * k + 1] = (f32x2){r.z, r.w};
        rq[2 * k] = rr[2 * k] * rr[2 * k]; rq[2 * k + 1] = rr[2 * k + 1] * rr[2 * k + 1];
    }
    int c00[NSRC], c11[NSRC];                             // cached tap identity (byte offsets of taps 00 / 11)
    float4 t00[NSRC][Q], t01[NSRC][Q], t10[NSRC][Q], t11[NSRC][Q];
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, NSRC * H * W * C * 4, 0x00020000);
    const int pix_bytes = C * 4, row_bytes = W * C * 4, img_bytes = H * W * C * 4, lane_bytes = c * 4;
#pragma unroll
    for (int v = 0; v < NSRC; ++v) {
        c00[v] = -1; c11[v] = -1;
#pragma unroll
        for (int k = 0; k < Q; ++k) t00[v][k] = t01[v][k] = t10[v][k] = t11[v][k] = z4;
    }
    // One IEEE division each, hoisted out of the sweep: the per-plane scalings become multiplies
    // (differs from Q/N, S*S/(N*N) by at most 1 ulp; the single-pass E[x^2]-E[x]^2 form is kept).
    const float n = (float)(NSRC + 1);
    const float inv_n = 1.0f / n, inv_nn = 1.0f / (n * n);

    // Plane-vectorised bookkeeping.  The kernel is VALU-bound and everything except the blend itself
    // is identical for the lg lanes of a pixel, so lane `sub` does ALL the per-view bookkeeping of
    // plane (batch + sub): projective map, floor, clamped tap offsets, zero-fill-masked bilinear
    // weights (already multiplied out to one weight per tap).  The 8 numbers per (plane, pixel, view)
    // go through a wave-private LDS table: while sweeping plane p every lane of the pixel reads them
    // back with two broadcast ds_read_b128 per view and only pays for the loads and the blend.
    extern __shared__ __attribute__((aligned(16))) float4 book_all[];
    const int ppw = 64 / lg;                                 // pixels per wave
    float4* book = book_all + (size_t)(threadIdx.x >> 6) * (64 * NSRC * 2);      // [lg planes][ppw pixels][NSRC][2]
    const int pixl = (threadIdx.x & 63) / lg;

    float* dstp = cost + ((size_t)dl0 * H * W + pix) * C + c;          // this lane's 16 bytes of plane dl0; walks plane by plane
    const size_t plane_elems = (size_t)H * W * C;
    for (int dlb = dl0; dlb < dl1; dlb += lg) {
        {
            const int dmy = d_begin + min(dlb + sub, dl1 - 1);
            float4* mine = book + ((size_t)sub * ppw + pixl) * (NSRC * 2);
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float* t = transforms + ((size_t)v * depth_total + dmy) * 8;
                const float4 ta = ld4(t), tb = ld4(t + 4);
                float proj = tb.z * xf + tb.w * yf + 1.0f;
                float inv = __builtin_amdgcn_rcpf(proj);        // v_rcp_f32: 1 ulp, exact for proj = 1
                float sx = (ta.x * xf + ta.y * yf + ta.z) * inv;
                float sy = (ta.w * xf + tb.x * yf + tb.y) * inv;
                float x0 = floorf(sx), y0 = floorf(sy);
                int ix0 = (int)x0, iy0 = (int)y0;                // v_cvt saturates, NaN -> 0
                int jx0 = min(max(ix0, 0), W - 1), jx1 = min(max(ix0 + 1, 0), W - 1);
                int jy0 = min(max(iy0, 0), H - 1), jy1 = min(max(iy0 + 1, 0), H - 1);
                // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): clamped indices and the strides are < 2^24 (host-checked)
                const int r0 = v * img_bytes + (int)__umul24(jy0, row_bytes), r1 = v * img_bytes + (int)__umul24(jy1, row_bytes);
                const int cx0 = (int)__umul24(jx0, pix_bytes), cx1 = (int)__umul24(jx1, pix_bytes);
                const int o00 = r0 + cx0, o01 = r0 + cx1;
                const int o10 = r1 + cx0, o11 = r1 + cx1;
                // per-tap zero fill folded into the separable weights: a tap is dropped iff its row or
                // its column is outside the image, exactly as reading 0 for it (w * finite = 0)
                const float wx1 = (ix0 >= 0 && ix0 < W) ? (x0 + 1.0f) - sx : 0.0f;
                const float wx0 = (ix0 + 1 >= 0 && ix0 + 1 < W) ? sx - x0 : 0.0f;
                const float wy1 = (iy0 >= 0 && iy0 < H) ? (y0 + 1.0f) - sy : 0.0f;
                const float wy0 = (iy0 + 1 >= 0 && iy0 + 1 < H) ? sy - y0 : 0.0f;
                mine[2 * v] = make_float4(__int_as_float(o00), __int_as_float(o01), __int_as_float(o10), __int_as_float(o11));
                mine[2 * v + 1] = make_float4(wy1 * wx1, wy1 * wx0, wy0 * wx1, wy0 * wx0);
            }
        }
        const int np = min(lg, dl1 - dlb);
        // one plane: refill the register tap cache where the taps moved, blend, reduce, store
        auto plane = [&](int p, const float4 (&ofs)[NSRC], const float4 (&wts)[NSRC]) __attribute__((always_inline)) {
            // phase A: all views' loads are issued before the first one is consumed.
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float4 of = ofs[v];
                const int o00 = __float_as_int(of.x), o11 = __float_as_int(of.w);
                if (o00 != c00[v] || o11 != c11[v]) {
                    const int o01 = __float_as_int(of.y), o10 = __float_as_int(of.z);
#pragma unroll
                    for (int k = 0; k < Q; ++k) {
                        t00[v][k] = ldb(srsrc, o00 + lane_bytes + 16 * k); t01[v][k] = ldb(srsrc, o01 + lane_bytes + 16 * k);
                        t10[v][k] = ldb(srsrc, o10 + lane_bytes + 16 * k); t11[v][k] = ldb(srsrc, o11 + lane_bytes + 16 * k);
                    }
                    c00[v] = o00; c11[v] = o11;
                }
            }
            // phase B: bilinear blend + running sums, two channels per packed instruction
            f32x2 S[2 * Q], Qs[2 * Q];
#pragma unroll
            for (int k = 0; k < 2 * Q; ++k) { S[k] = rr[k]; Qs[k] = rq[k]; }
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float w00 = wts[v].x, w01 = wts[v].y, w10 = wts[v].z, w11 = wts[v].w;
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    f32x2 a0 = (f32x2){t00[v][k].x, t00[v][k].y}, a1 = (f32x2){t00[v][k].z, t00[v][k].w};
                    f32x2 b0 = (f32x2){t01[v][k].x, t01[v][k].y}, b1 = (f32x2){t01[v][k].z, t01[v][k].w};
                    f32x2 c0 = (f32x2){t10[v][k].x, t10[v][k].y}, c1 = (f32x2){t10[v][k].z, t10[v][k].w};
                    f32x2 e0 = (f32x2){t11[v][k].x, t11[v][k].y}, e1 = (f32x2){t11[v][k].z, t11[v][k].w};
                    f32x2 w0 = w00 * a0 + w01 * b0 + w10 * c0 + w11 * e0;
                    f32x2 w1 = w00 * a1 + w01 * b1 + w10 * c1 + w11 * e1;
                    S[2 * k] += w0; S[2 * k + 1] += w1;
                    Qs[2 * k] += w0 * w0; Qs[2 * k + 1] += w1 * w1;
                }
            }
            float* dst = dstp;
            dstp += plane_elems;
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                f32x2 o0, o1;
                if (variant == 0) {
                    o0 = Qs[2 * k] * inv_n - (S[2 * k] * S[2 * k]) * inv_nn;
                    o1 = Qs[2 * k + 1] * inv_n - (S[2 * k + 1] * S[2 * k + 1]) * inv_nn;
                } else {
                    f32x2 m0 = S[2 * k] * inv_n, m1 = S[2 * k + 1] * inv_n;
                    o0 = Qs[2 * k] * inv_n - m0 * m0; o1 = Qs[2 * k + 1] * inv_n - m1 * m1;
                }
                if (negate) { o0 = -o0; o1 = -o1; }
                // non-temporal: the 503 MB volume streams out and must not push the 13 MB of feature maps, which every plane
                // re-reads, out of the L2 (199.7 -> 185.6 us inside a depth map, 246 -> 228 us alone)
                __builtin_nontemporal_store((f32x4_nt){o0[0], o0[1], o1[0], o1[1]}, reinterpret_cast<f32x4_nt*>(dst + 4 * k));
            }
        };
        auto fetch = [&](int p, float4 (&ofs)[NSRC], float4 (&wts)[NSRC]) __attribute__((always_inline)) {
            const float4* bk = book + ((size_t)min(p, lg - 1) * ppw + pixl) * (NSRC * 2);
#pragma unroll
            for (int v = 0; v < NSRC; ++v) { ofs[v] = bk[2 * v]; wts[v] = bk[2 * v + 1]; }
        };
        // (reading plane p+1's entries during plane p costs a second register set and an occupancy
        // step: measured slower)
        for (int p = 0; p < np; ++p) {
            float4 ofs[NSRC], wts[NSRC];
            fetch(p, ofs, wts);
            __builtin_amdgcn_sched_barrier(0);
            plane(p, ofs, wts);
        }
    }
}

template <int NSRC>
void launch_sweep(const float* ref, const float* src, const float* transforms, int depth_total,
                  int d_begin, int d_count, int H, int W, int C, int variant, int negate,
                  float* cost, hipStream_t st) {
    static const int ppb_env = getenv("MVS_CV_PPB") ? atoi(getenv("MVS_CV_PPB")) : 0;
    const int ppb0 = ppb_env > 0 ? ppb_env : 16;
    const int ppb = d_count < ppb0 ? d_count : ppb0;
    // Q = 2 (8 channels per lane) halves the per-lane bookkeeping per channel but needs 236 VGPRs
    // (2 waves/SIMD instead of 3): measured 0.258 ms vs 0.243 ms at the metric config, so it stays off.
    const bool wide = false;
    const int lg = wide ? C / 8 : C / 4;
    long long total = (long long)H * W * lg;
    dim3 grid(mvs_cdiv(total, 256), mvs_cdiv(d_count, ppb));
    const size_t smem = (size_t)4 * 64 * NSRC * 2 * sizeof(float4);     // bookkeeping table, 2 KB per wave and view
    if (wide)
        cost_volume_sweep_kernel<NSRC, 2><<<grid, 256, smem, st>>>(ref, src, transforms, depth_total, d_begin,
                                                                d_count, ppb, H, W, C, variant, negate, cost);
    else
        cost_volume_sweep_kernel<NSRC, 1><<<grid, 256, smem, st>>>(ref, src, transforms, depth_total, d_begin,
                                                                d_count, ppb, H, W, C, variant, negate, cost);
}

// ---- LDS-staged sweep (C = 32), OPT-IN: MVS_CV_LDS=1 ------------------------------------------------------------------
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

template <int BORDER>
__global__ void __launch_bounds__(256)
warp_kernel(const float* __restrict__ img, const float* __restrict__ t, int H, int W, int C,
            float* __restrict__ out) {
    const int cq = C >> 2;
    const long long total = (long long)H * W * cq;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int c = (int)(idx % cq) * 4;
    long long pix = idx / cq;
    int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    float4 w = warp_sample<BORDER>(img, t, (float)x, (float)y, H, W, C, c);
    *reinterpret_cast<float4*>(out + (size_t)pix * C + c) = w;
}

}  // namespace

extern "C" int mvs_cost_volume_f32(const float* ref, const float* src, const float* transforms,
                                   int view_num, int depth_total, int d_begin, int d_count,
                                   int H, int W, int C, int variant, int negate, int border,
                                   float* cost, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && cost);
    MVS_CHECK_ARG(view_num >= 2 && depth_total >= 1 && d_begin >= 0 && d_count >= 1 &&
                  d_begin + d_count <= depth_total && H > 0 && W > 0 && C > 0);
    if (C % 4 != 0) return MVS_E_SHAPE;
    long long total = (long long)H * W * (C / 4);
    dim3 grid(mvs_cdiv(total, 256), d_count);
    const int cq_ = C / 4;
    const bool cq_pow2 = cq_ <= 16 && (cq_ & (cq_ - 1)) == 0;     // lanes of a pixel stay inside one wave
    // 32-bit byte offsets into the source maps: all views together must stay below 2 GiB
    const bool off32 = (long long)(view_num - 1) * H * W * C * 4 < (1LL << 31);
    // The LDS-staged sweep is OPT-IN (MVS_CV_LDS=1): exact, tested, and measured no faster than the register tap cache at
    // the metric workload (188 vs 186-190 us, round 2) -- see the note above cost_volume_lds2_kernel.
    static const int use_lds = getenv("MVS_CV_LDS") ? atoi(getenv("MVS_CV_LDS")) : 0;
    // MFMA-blend sweeps (cost_volume_mfma.hip), OPT-IN: MVS_CV_MFMA=1 (LDS-staged) / 2 (direct).  Exact and tested; measured
    // 262-293 us at the metric workload against 186-200 us of the register tap cache below.
    static const int use_mfma = getenv("MVS_CV_MFMA") ? atoi(getenv("MVS_CV_MFMA")) : 0;
    if (border == 0 && C == 32 && view_num >= 2 && off32 && H < 65535 && W < 65535 && use_mfma >= 1 && use_lds < 1)
        return mvs_cost_volume_mfma_launch(ref, src, transforms, view_num - 1, depth_total, d_begin, d_count, H, W, variant,
                                           negate, cost, mvs_stream(stream));
    if (border == 0 && C == 32 && view_num >= 2 && view_num <= 8 && off32 && use_lds >= 1) {  // LDS-staged sweep (opt-in)
        hipStream_t st = mvs_stream(stream);
        switch (view_num - 1) {
#define MVS_LSWEEP2(NS) case NS: return launch_lds2_sweep<NS>(ref, src, transforms, depth_total, d_begin, d_count, H, W, variant, negate, cost, st);
            MVS_LSWEEP2(1) MVS_LSWEEP2(2) MVS_LSWEEP2(3) MVS_LSWEEP2(4) MVS_LSWEEP2(5) MVS_LSWEEP2(6) MVS_LSWEEP2(7)
#undef MVS_LSWEEP2
        }
    }
    const bool u24 = H < (1 << 24) && W < (1 << 24) && (long long)W * C * 4 < (1 << 24);     // the sweep's 24-bit offset multiplies
    if (border == 0 && view_num <= 8 && cq_pow2 && off32 && u24) {      // depth sweep with register tap reuse
        hipStream_t st = mvs_stream(stream);
#define MVS_SWEEP(NS) case NS: launch_sweep<NS>(ref, src, transforms, depth_total, d_begin, d_count, H, W, C, variant, negate, cost, st); break;
        switch (view_num - 1) {
            MVS_SWEEP(1) MVS_SWEEP(2) MVS_SWEEP(3) MVS_SWEEP(4) MVS_SWEEP(5) MVS_SWEEP(6) MVS_SWEEP(7)
        }
#undef MVS_SWEEP
        MVS_LAUNCH_RET();
    }
    if (border == 0)
        cost_volume_kernel<0><<<grid, 256, 0, mvs_stream(stream)>>>(
            ref, src, transforms, view_num - 1, depth_total, d_begin, H, W, C, variant, negate, cost);
    else
        cost_volume_kernel<1><<<grid, 256, 0, mvs_stream(stream)>>>(
            ref, src, transforms, view_num - 1, depth_total, d_begin, H, W, C, variant, negate, cost);
    MVS_LAUNCH_RET();
}

// Rounds of 8 planes (per wave) of the LDS-staged sweep that took the direct path since the last call (test / tuning
// aid: a geometry whose footprints do not fit the LDS budget still gives exact results, only slower).  Synchronises.
extern "C" int mvs_cost_volume_fallback_rounds(int* rounds) {
    MVS_CHECK_ARG(rounds);
    int zero = 0;
    hipError_t e = hipMemcpyFromSymbol(rounds, HIP_SYMBOL(g_cvl_fallback_rounds), sizeof(int));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_cvl_fallback_rounds), &zero, sizeof(int));
    return (int)e;
}

extern "C" int mvs_warp_f32(const float* image, const float* transform8, int H, int W, int C,
                            int border, float* out, void* stream) {
    MVS_CHECK_ARG(image && transform8 && out && H > 0 && W > 0 && C > 0);
    if (C % 4 != 0) return MVS_E_SHAPE;
    long long total = (long long)H * W * (C / 4);
    if (border == 0)
        warp_kernel<0><<<mvs_cdiv(total, 256), 256, 0, mvs_stream(stream)>>>(image, transform8, H, W, C, out);
    else
        warp_kernel<1><<<mvs_cdiv(total, 256), 256, 0, mvs_stream(stream)>>>(image, transform8, H, W, C, out);
    MVS_LAUNCH_RET();
}
