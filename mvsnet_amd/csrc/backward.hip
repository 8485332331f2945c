// Backward kernels of the plane-sweep path for training (the reference gets them from
// TensorFlow's autodiff of mvsnet/model.py:257-372 via opt.compute_gradients, mvsnet/train.py:428-429):
//   soft-argmin (model.py:343-366), BatchNorm+ReLU in batch-statistics mode (network.py:492-509),
//   warp + variance (homography_warping.py:211-253, model.py:315-334), and the RMSProp update
//   (train.py:259, tf.train.RMSPropOptimizer defaults).
// Input gradients of the 3D convolutions reuse the forward MFMA kernels (a stride-2 convolution's
// input gradient IS conv3d_transpose with the same kernel array and vice versa; stride 1 takes the
// flipped, transposed kernel); weight gradients are in conv3d_wgrad.hip.
//
// All of these are HBM-bound elementwise / reduction / scatter passes: float4 per lane, channel-last.
#include "common.h"
#include <climits>

namespace {

__device__ __forceinline__ float depth_at(int d, int D, float start, float interval, int inverse) {
    float end = start + ((float)D - 1.0f) * interval;
    float denom = (float)(D > 1 ? D - 1 : 1);
    if (inverse) {
        float a = 1.0f / start, b = 1.0f / end;
        return 1.0f / (a + (float)d * ((b - a) / denom));
    }
    return start + (float)d * ((end - start) / denom);
}

// depth = sum_d P_d z_d, P = softmax(-reg)  =>  d depth / d reg_d = -P_d (z_d - depth).
// The 4-bucket probability map prob = P[l0] + P[r0] + P[l1] + P[r1] (model.py:45-144; the bucket indices come from
// floor / ceil of the depth and carry no gradient, exactly as in TensorFlow) adds, when a gradient for it is given
// (training through the refinement network, which takes the map as an input channel),
//   d prob / d reg_d = -P_d (m_d - prob),  m_d = how many of the four buckets are plane d.
// One lane per pixel, three coalesced sweeps over depth (max, sums, write).
__global__ void __launch_bounds__(256)
softargmin_bwd_kernel(const float* __restrict__ reg, const float* __restrict__ g_depth,
                      const float* __restrict__ g_prob, int D, int HW,
                      float start, float interval, int inverse, float* __restrict__ g_reg) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const float* col = reg + pix;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, -col[(size_t)d * HW]);
    float se = 0.f, sz = 0.f;
    for (int d = 0; d < D; ++d) {
        float e = __expf(-col[(size_t)d * HW] - m);
        se += e; sz += e * depth_at(d, D, start, interval, inverse);
    }
    const float depth = sz / se, g = (g_depth ? g_depth[pix] : 0.f) / se;
    int l0 = 0, r0 = 0, l1 = 0, r1 = 0;
    float gp = 0.f, prob = 0.f;
    if (g_prob) {                                              // bucket indices as in softargmin.hip
        if (inverse) {
            float end = start + ((float)D - 1.0f) * interval;
            float inv_s = 1.0f / start, inv_e = 1.0f / end;
            float inv_int = (inv_s - inv_e) / ((float)D - 1.0f);
            float idx = (1.0f / depth - inv_e) / inv_int;
            l0 = D - (int)ceilf(idx) - 1; r0 = D - (int)floorf(idx) - 1;
        } else {
            float idx = (depth - start) / interval;
            l0 = (int)floorf(idx); r0 = (int)ceilf(idx);
        }
        l0 = min(max(l0, 0), D - 1); r0 = min(max(r0, 0), D - 1);
        l1 = min(max(l0 - 1, 0), D - 1); r1 = min(max(r0 + 1, 0), D - 1);
        auto P = [&](int d) { return __expf(-col[(size_t)d * HW] - m) / se; };
        prob = (P(l0) + P(r0)) + (P(l1) + P(r1));
        gp = g_prob[pix] / se;
    }
    for (int d = 0; d < D; ++d) {
        float e = __expf(-col[(size_t)d * HW] - m);
        float v = -g * e * (depth_at(d, D, start, interval, inverse) - depth);
        if (g_prob) {
            const float md = (float)((d == l0) + (d == r0) + (d == l1) + (d == r1));
            v -= gp * e * (md - prob);
        }
        g_reg[(size_t)d * HW + pix] = v;
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// out = act(y*s+t) [+ act(y2*s2+t2)]; act = ReLU when the scale pointer is given, identity otherwise:
// the normalised layer input the forward kernels build on load, materialised for the weight gradient.
__global__ void __launch_bounds__(256)
bn_relu_kernel(const float* __restrict__ y, const float* __restrict__ s, const float* __restrict__ t,
               const float* __restrict__ y2, const float* __restrict__ s2, const float* __restrict__ t2,
               size_t n4, int cq, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cq) * 4;
        float4 v = ld4(y + 4 * i);
        if (s) {
            float4 a = ld4(s + c), b = ld4(t + c);
            v.x = relu(v.x * a.x + b.x); v.y = relu(v.y * a.y + b.y);
            v.z = relu(v.z * a.z + b.z); v.w = relu(v.w * a.w + b.w);
        }
        if (y2) {
            float4 u = ld4(y2 + 4 * i);
            if (s2) {
                float4 a = ld4(s2 + c), b = ld4(t2 + c);
                u.x = relu(u.x * a.x + b.x); u.y = relu(u.y * a.y + b.y);
                u.z = relu(u.z * a.z + b.z); u.w = relu(u.w * a.w + b.w);
            }
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        st4(out + 4 * i, v);
    }
}

// float64 atomics on one address serialise at ~80 ns each: the ~1000 workgroups of a pass-1 launch add into
// BN_BWD_SLOTS partial rows that pass 2 folds.
constexpr int BN_BWD_SLOTS = 32;

// BatchNorm(batch statistics) + ReLU backward, pass 1: per channel  sum gz  and  sum gz * xhat  with
// z = y*scale + shift, gz = (g1 [+ g2]) * [z > 0], xhat = (y - mean) * inv_std.
// A thread keeps one channel quad (the grid stride is a multiple of C/4); float partials per thread,
// LDS tree per quad, one float64 atomic per channel per workgroup.
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const float* __restrict__ y, const double* __restrict__ stats, double count, float eps,
                     const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ g1, const float* __restrict__ g2, size_t n4, int cq,
                     double* __restrict__ sums) {
    __shared__ float red[256][8];
    const int tid = threadIdx.x;
    const int c = (tid % cq) * 4, C = cq * 4;
    float mean[4], inv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double mu = stats[c + k] / count, var = stats[C + c + k] / count - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[k] = (float)mu; inv[k] = (float)(1.0 / sqrt(var + (double)eps));
    }
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ld4(y + 4 * i), g = ld4(g1 + 4 * i);
        if (g2) { float4 h = ld4(g2 + 4 * i); g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
        const float ss[4] = {sc.x, sc.y, sc.z, sc.w}, tt[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gz = (vv[k] * ss[k] + tt[k] > 0.f) ? gg[k] : 0.f;
            a[k] += gz; b[k] += gz * ((vv[k] - mean[k]) * inv[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid][k] = a[k]; red[tid][4 + k] = b[k]; }
    __syncthreads();
    if (tid < cq) {                                    // threads tid, tid+cq, ... share this quad
        double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
        for (int j = tid; j < 256; j += cq)
#pragma unroll
            for (int k = 0; k < 4; ++k) { sa[k] += red[j][k]; sb[k] += red[j][4 + k]; }
        double* slot = sums + (size_t)(blockIdx.x % BN_BWD_SLOTS) * 2 * C;     // spread the float64 atomics
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(slot + c + k, sa[k]);
            atomicAdd(slot + C + c + k, sb[k]);
        }
    }
}

// pass 2: g_y = gamma * inv_std * (gz - mean(gz) - xhat * mean(gz * xhat)); block 0 also emits
// g_gamma = sum gz*xhat and g_beta = sum gz.
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ y, const double* __restrict__ stats, double count, float eps,
                    const float* __restrict__ scale, const float* __restrict__ shift,
                    const float* __restrict__ gamma, const float* __restrict__ g1,
                    const float* __restrict__ g2, const double* __restrict__ sums, size_t n4, int cq,
                    float* __restrict__ g_y, float* __restrict__ g_gamma, float* __restrict__ g_beta) {
    const int tid = threadIdx.x;
    const int c = (tid % cq) * 4, C = cq * 4;
    __shared__ double tot[2 * 64];                     // folded slot rows: [sum gz | sum gz*xhat] per channel (C <= 64)
    if (tid < 2 * C) {
        double t = 0.0;
        for (int sl = 0; sl < BN_BWD_SLOTS; ++sl) t += sums[(size_t)sl * 2 * C + tid];
        tot[tid] = t;
    }
    __syncthreads();
    float mean[4], inv[4], k0[4], k1[4], k2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double mu = stats[c + k] / count, var = stats[C + c + k] / count - mu * mu;
        if (var < 0.0) var = 0.0;
        double is = 1.0 / sqrt(var + (double)eps);
        mean[k] = (float)mu; inv[k] = (float)is;
        k0[k] = (float)((double)gamma[c + k] * is);
        const double s1 = tot[c + k], s2 = tot[C + c + k];
        k1[k] = (float)(s1 / count);
        k2[k] = (float)(s2 / count);
        if (blockIdx.x == 0 && tid < cq) {
            if (g_beta) g_beta[c + k] = (float)s1;
            if (g_gamma) g_gamma[c + k] = (float)s2;
        }
    }
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ld4(y + 4 * i), g = ld4(g1 + 4 * i);
        if (g2) { float4 h = ld4(g2 + 4 * i); g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
        const float ss[4] = {sc.x, sc.y, sc.z, sc.w}, tt[4] = {sh.x, sh.y, sh.z, sh.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gz = (vv[k] * ss[k] + tt[k] > 0.f) ? gg[k] : 0.f;
            o[k] = k0[k] * (gz - k1[k] - (vv[k] - mean[k]) * inv[k] * k2[k]);
        }
        st4(g_y + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// Warp + variance backward.  cost = Q/N - (S/N)^2 with S = F_ref + sum_v W_v, Q = F_ref^2 + sum_v W_v^2
// (both forward variants have this derivative):  d cost / d X = (2/N) (X - S/N) for X in {F_ref, W_v}.
// One lane = (pixel, 4 channels) marching over `planes_per_block` planes: recomputes the taps exactly as
// the forward does (zero fill per tap), accumulates the reference-view gradient in registers and
// scatters each source view's gradient through its four bilinear weights with float atomics
// (red.add, no return: the taps of neighbouring pixels coalesce in L2).
struct Tap { int o00, o01, o10, o11; float w00, w01, w10, w11; };
__device__ __forceinline__ Tap make_tap(const float* __restrict__ t, float xf, float yf, int H, int W, int C, int c) {
    float proj = t[6] * xf + t[7] * yf + 1.0f;
    float sx = (t[0] * xf + t[1] * yf + t[2]) / proj;
    float sy = (t[3] * xf + t[4] * yf + t[5]) / proj;
    float x0 = floorf(sx), y0 = floorf(sy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    bool okx0 = (x0 >= 0.0f) && (x0 < (float)W), okx1 = (x1 >= 0.0f) && (x1 < (float)W);
    bool oky0 = (y0 >= 0.0f) && (y0 < (float)H), oky1 = (y1 >= 0.0f) && (y1 < (float)H);
    int ix0 = okx0 ? (int)x0 : 0, ix1 = okx1 ? (int)x1 : 0;
    int iy0 = oky0 ? (int)y0 : 0, iy1 = oky1 ? (int)y1 : 0;
    float wx1 = x1 - sx, wx0 = sx - x0, wy1 = y1 - sy, wy0 = sy - y0;
    Tap p;
    p.o00 = (okx0 && oky0) ? (iy0 * W + ix0) * C + c : -1;
    p.o01 = (okx1 && oky0) ? (iy0 * W + ix1) * C + c : -1;
    p.o10 = (okx0 && oky1) ? (iy1 * W + ix0) * C + c : -1;
    p.o11 = (okx1 && oky1) ? (iy1 * W + ix1) * C + c : -1;
    p.w00 = wy1 * wx1; p.w01 = wy1 * wx0; p.w10 = wy0 * wx1; p.w11 = wy0 * wx0;
    return p;
}
__device__ __forceinline__ float4 tap_gather(const float* __restrict__ img, const Tap& p) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a = p.o00 >= 0 ? ld4(img + p.o00) : z, b = p.o01 >= 0 ? ld4(img + p.o01) : z;
    float4 c = p.o10 >= 0 ? ld4(img + p.o10) : z, d = p.o11 >= 0 ? ld4(img + p.o11) : z;
    float4 o;
    o.x = p.w00 * a.x + p.w01 * b.x + p.w10 * c.x + p.w11 * d.x;
    o.y = p.w00 * a.y + p.w01 * b.y + p.w10 * c.y + p.w11 * d.y;
    o.z = p.w00 * a.z + p.w01 * b.z + p.w10 * c.z + p.w11 * d.z;
    o.w = p.w00 * a.w + p.w01 * b.w + p.w10 * c.w + p.w11 * d.w;
    return o;
}
__device__ __forceinline__ void red_add4(float* p, float w, float4 g) {
    unsafeAtomicAdd(p + 0, w * g.x); unsafeAtomicAdd(p + 1, w * g.y);
    unsafeAtomicAdd(p + 2, w * g.z); unsafeAtomicAdd(p + 3, w * g.w);
}

constexpr int CVB_MAX_SRC = 8;

// One view's pending scatter: the 2x2 cell the lane's sample point is in and the gradient gathered for its
// four taps.  Along depth the sample point slides along the epipolar line by a fraction of a pixel per
// plane, so consecutive planes mostly hit the SAME cell: their contributions are summed here and go to
// memory (float atomics) only when the cell changes -- several times fewer atomics than one per plane
// (11.9 -> 2.9 ms at N=3, D=192, 120x160).  Tried and dropped: scattering into an LDS window of the source
// patch per 8x8 / 16x8 pixel tile (ds_add_f32, flushed when the patch drifts out): ~20x fewer global
// atomics but 5.3 ms -- a barrier per plane and the LDS read-modify-writes cost more than they saved.
struct Pending {
    int ix0, iy0;                    // floor of the sample point (may be -1 .. W-1 / H-1); INT_MIN = empty
    float4 a00, a01, a10, a11;
};
__device__ __forceinline__ void pend_flush(Pending& p, float* __restrict__ gs, int H, int W, int C, int c) {
    if (p.ix0 == INT_MIN) return;
    const bool okx0 = p.ix0 >= 0 && p.ix0 < W, okx1 = p.ix0 + 1 >= 0 && p.ix0 + 1 < W;
    const bool oky0 = p.iy0 >= 0 && p.iy0 < H, oky1 = p.iy0 + 1 >= 0 && p.iy0 + 1 < H;
    float* b = gs + ((long long)p.iy0 * W + p.ix0) * C + c;
    if (okx0 && oky0) red_add4(b, 1.f, p.a00);
    if (okx1 && oky0) red_add4(b + C, 1.f, p.a01);
    if (okx0 && oky1) red_add4(b + (long long)W * C, 1.f, p.a10);
    if (okx1 && oky1) red_add4(b + (long long)W * C + C, 1.f, p.a11);
}
__device__ __forceinline__ void acc4(float4& a, float w, float4 g) {
    a.x += w * g.x; a.y += w * g.y; a.z += w * g.z; a.w += w * g.w;
}

template <int NSRC>
__global__ void __launch_bounds__(256)
cost_volume_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                       const float* __restrict__ transforms, int D, int planes_per_block,
                       int H, int W, int C, const float* __restrict__ g1, const float* __restrict__ g2,
                       float* __restrict__ g_ref, float* __restrict__ g_src) {
    const int cq = C >> 2;
    const long long total = (long long)H * W * cq;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cq) * 4;
    const long long pix = idx / cq;
    const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    const float xf = (float)x, yf = (float)y;
    const size_t img = (size_t)H * W * C;
    const float n = (float)(NSRC + 1), two_n = 2.0f / n;
    const float4 r = ld4(ref + (size_t)pix * C + c);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gr = z4;
    Pending pd[NSRC];
#pragma unroll
    for (int v = 0; v < NSRC; ++v) { pd[v].ix0 = INT_MIN; pd[v].iy0 = 0; pd[v].a00 = pd[v].a01 = pd[v].a10 = pd[v].a11 = z4; }
    const int d0 = blockIdx.y * planes_per_block, d1 = min(d0 + planes_per_block, D);
    for (int d = d0; d < d1; ++d) {
        const size_t vo = ((size_t)d * H * W + pix) * C + c;
        float4 g = ld4(g1 + vo);
        if (g2) { float4 h = ld4(g2 + vo); g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        g.x *= two_n; g.y *= two_n; g.z *= two_n; g.w *= two_n;
        Tap tp[NSRC]; float4 wv[NSRC]; int cx[NSRC], cy[NSRC];
        float4 S = r;
#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const float* t = transforms + ((size_t)v * D + d) * 8;
            tp[v] = make_tap(t, xf, yf, H, W, C, c);
            {   // the cell index itself (make_tap only keeps the in-range tap offsets)
                float proj = t[6] * xf + t[7] * yf + 1.0f;
                float fx = floorf((t[0] * xf + t[1] * yf + t[2]) / proj), fy = floorf((t[3] * xf + t[4] * yf + t[5]) / proj);
                // far outside the image nothing is in range: one shared "nowhere" cell keeps the ints finite
                const bool near_img = fx >= -1.f && fx < (float)W && fy >= -1.f && fy < (float)H;
                cx[v] = near_img ? (int)fx : -2; cy[v] = near_img ? (int)fy : -2;
            }
            wv[v] = tap_gather(src + v * img, tp[v]);
            S.x += wv[v].x; S.y += wv[v].y; S.z += wv[v].z; S.w += wv[v].w;
        }
        S.x /= n; S.y /= n; S.z /= n; S.w /= n;
        gr.x += g.x * (r.x - S.x); gr.y += g.y * (r.y - S.y);
        gr.z += g.z * (r.z - S.z); gr.w += g.w * (r.w - S.w);
#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            if (cx[v] != pd[v].ix0 || cy[v] != pd[v].iy0) {
                pend_flush(pd[v], g_src + v * img, H, W, C, c);
                pd[v].ix0 = cx[v]; pd[v].iy0 = cy[v];
                pd[v].a00 = pd[v].a01 = pd[v].a10 = pd[v].a11 = z4;
            }
            const float4 gw = make_float4(g.x * (wv[v].x - S.x), g.y * (wv[v].y - S.y),
                                          g.z * (wv[v].z - S.z), g.w * (wv[v].w - S.w));
            acc4(pd[v].a00, tp[v].w00, gw); acc4(pd[v].a01, tp[v].w01, gw);
            acc4(pd[v].a10, tp[v].w10, gw); acc4(pd[v].a11, tp[v].w11, gw);
        }
    }
#pragma unroll
    for (int v = 0; v < NSRC; ++v) pend_flush(pd[v], g_src + v * img, H, W, C, c);
    float* pr = g_ref + (size_t)pix * C + c;
    unsafeAtomicAdd(pr + 0, gr.x); unsafeAtomicAdd(pr + 1, gr.y);
    unsafeAtomicAdd(pr + 2, gr.z); unsafeAtomicAdd(pr + 3, gr.w);
}

// ---- Atomic-free, deterministic variant of the warp + variance backward -----------------------------------------
// Pass 1 (reference frame): a lane = (pixel, 4 channels) marches over a chunk of planes, recomputes the taps and S,
// stores every source view's warped-sample gradient  GW(v, d, pixel, c) = (2/N) g (W_v - S/N)  and keeps the
// reference-view gradient in registers (one partial row per chunk).
// Pass 2 (source frame): a lane = one source pixel of one view with all C channels in registers marches over a chunk
// of planes and GATHERS: the reference pixels whose sample point falls within one pixel of it are found through the
// inverse plane homography (a neighbourhood sized by the local Jacobian), each candidate is forward-mapped with
// exactly the expression pass 1 used and contributes its bilinear weight times its GW row (128 contiguous bytes).
// A last kernel adds the chunk rows in a fixed order.  No atomics: bit-reproducible, and 2.9 -> ~0.7 ms at N=3,
// D=192, 120x160 (the float atomics of the scatter version run at ~80 G/s however they are batched).
constexpr int CVG_CH1 = 48, CVG_CH2 = 24;            // planes per chunk of pass 1 / pass 2

template <int NSRC>
__global__ void __launch_bounds__(256)
cvb_pass1_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ transforms,
                 int D, int H, int W, int C, const float* __restrict__ g1, const float* __restrict__ g2,
                 float* __restrict__ gw, float* __restrict__ ref_part) {
    const int cq = C >> 2;
    const long long total = (long long)H * W * cq;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cq) * 4;
    const long long pix = idx / cq;
    const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    const float xf = (float)x, yf = (float)y;
    const size_t img = (size_t)H * W * C;
    const float n = (float)(NSRC + 1), two_n = 2.0f / n;
    const float4 r = ld4(ref + (size_t)pix * C + c);
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    const int d0 = blockIdx.y * CVG_CH1, d1 = min(d0 + CVG_CH1, D);
    for (int d = d0; d < d1; ++d) {
        const size_t vo = ((size_t)d * H * W + pix) * C + c;
        float4 g = ld4(g1 + vo);
        if (g2) { float4 h = ld4(g2 + vo); g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
        g.x *= two_n; g.y *= two_n; g.z *= two_n; g.w *= two_n;
        float4 wv[NSRC];
        float4 S = r;
#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const Tap tp = make_tap(transforms + ((size_t)v * D + d) * 8, xf, yf, H, W, C, c);
            wv[v] = tap_gather(src + v * img, tp);
            S.x += wv[v].x; S.y += wv[v].y; S.z += wv[v].z; S.w += wv[v].w;
        }
        S.x /= n; S.y /= n; S.z /= n; S.w /= n;
        gr.x += g.x * (r.x - S.x); gr.y += g.y * (r.y - S.y);
        gr.z += g.z * (r.z - S.z); gr.w += g.w * (r.w - S.w);
#pragma unroll
        for (int v = 0; v < NSRC; ++v)
            st4(gw + ((size_t)v * D + d) * img + (size_t)pix * C + c,
                make_float4(g.x * (wv[v].x - S.x), g.y * (wv[v].y - S.y), g.z * (wv[v].z - S.z), g.w * (wv[v].w - S.w)));
    }
    st4(ref_part + (size_t)blockIdx.y * img + (size_t)pix * C + c, gr);
}

template <int CQ>
__global__ void __launch_bounds__(256)
cvb_pass2_kernel(const float* __restrict__ transforms, int D, int H, int W, const float* __restrict__ gw,
                 float* __restrict__ src_part) {
    constexpr int C = 4 * CQ;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= H * W) return;
    const int v = blockIdx.z, n_src = gridDim.z;
    const int ys = s / W, xs = s - ys * W;
    const float xsf = (float)xs, ysf = (float)ys;
    const size_t img = (size_t)H * W * C;
    float4 acc[CQ];
#pragma unroll
    for (int k = 0; k < CQ; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int d0 = blockIdx.y * CVG_CH2, d1 = min(d0 + CVG_CH2, D);
    for (int d = d0; d < d1; ++d) {
        const float* t = transforms + ((size_t)v * D + d) * 8;
        const float a0 = t[0], a1 = t[1], a2 = t[2], b0 = t[3], b1 = t[4], b2 = t[5], c0 = t[6], c1 = t[7];
        // adjugate of [[a0 a1 a2] [b0 b1 b2] [c0 c1 1]]: the inverse map up to scale
        const float i00 = b1 - b2 * c1, i01 = a2 * c1 - a1, i02 = a1 * b2 - a2 * b1;
        const float i10 = b2 * c0 - b0, i11 = a0 - a2 * c0, i12 = a2 * b0 - a0 * b2;
        const float i20 = b0 * c1 - b1 * c0, i21 = a1 * c0 - a0 * c1, i22 = a0 * b1 - a1 * b0;
        auto inv_map = [&](float u, float w_, float& px, float& py) {
            const float q = 1.0f / (i20 * u + i21 * w_ + i22);
            px = (i00 * u + i01 * w_ + i02) * q; py = (i10 * u + i11 * w_ + i12) * q;
        };
        float px, py, pxu, pyu, pxv, pyv;
        inv_map(xsf, ysf, px, py); inv_map(xsf + 1.0f, ysf, pxu, pyu); inv_map(xsf, ysf + 1.0f, pxv, pyv);
        // reference pixels within one source pixel of s: |dp| <= |J^-1| (1,1) plus a margin for the curvature
        float bx = fabsf(pxu - px) + fabsf(pxv - px) + 0.5f, by = fabsf(pyu - py) + fabsf(pyv - py) + 0.5f;
        if (!(bx < 8.0f)) bx = 8.0f;                     // also catches NaN; see the note on extreme minification
        if (!(by < 8.0f)) by = 8.0f;
        if (!(fabsf(px) < 1e8f) || !(fabsf(py) < 1e8f)) continue;       // the plane does not see this source pixel
        const int x_lo = max(0, (int)ceilf(px - bx)), x_hi = min(W - 1, (int)floorf(px + bx));
        const int y_lo = max(0, (int)ceilf(py - by)), y_hi = min(H - 1, (int)floorf(py + by));
        const float* gwd = gw + ((size_t)v * D + d) * img;
        for (int yy = y_lo; yy <= y_hi; ++yy)
            for (int xx = x_lo; xx <= x_hi; ++xx) {
                const float xf = (float)xx, yf = (float)yy;
                const float proj = c0 * xf + c1 * yf + 1.0f;               // exactly make_tap's arithmetic
                const float sx = (a0 * xf + a1 * yf + a2) / proj, sy = (b0 * xf + b1 * yf + b2) / proj;
                const float x0 = floorf(sx), y0 = floorf(sy);
                float wx, wy;
                if (xsf == x0) wx = (x0 + 1.0f) - sx; else if (xsf == x0 + 1.0f) wx = sx - x0; else continue;
                if (ysf == y0) wy = (y0 + 1.0f) - sy; else if (ysf == y0 + 1.0f) wy = sy - y0; else continue;
                const float wgt = wy * wx;
                const float* row = gwd + ((size_t)yy * W + xx) * C;
#pragma unroll
                for (int k = 0; k < CQ; ++k) {
                    const float4 gv = ld4(row + 4 * k);
                    acc[k].x += wgt * gv.x; acc[k].y += wgt * gv.y; acc[k].z += wgt * gv.z; acc[k].w += wgt * gv.w;
                }
            }
    }
    float* out = src_part + ((size_t)blockIdx.y * n_src + v) * img + (size_t)s * C;
#pragma unroll
    for (int k = 0; k < CQ; ++k) st4(out + 4 * k, acc[k]);
}

// out(i) = sum over `rows` partial rows, fixed order
__global__ void __launch_bounds__(256)
cvb_fold_kernel(const float* __restrict__ part, int rows, size_t n4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = ld4(part + 4 * i);
    for (int r = 1; r < rows; ++r) { const float4 p = ld4(part + (size_t)r * n4 * 4 + 4 * i); s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
    st4(out + 4 * i, s);
}

// ---- GroupNorm (+ReLU) of the 2D towers for training (Network.conv_gn / deconv_gn, network.py:217-276,350-409:
// groups of 8 channels, biased variance, eps 1e-5).  torch's group_norm spends ~0.4 ms per layer in its
// moments kernel on channel-last tensors (12 of the towers' 14.5 ms forward); these are plain HBM passes.
// x (V, HW, C) channel-last; stats (V, 2, C) float64 per-channel [sum, sumsq] (group moments are folded from
// the 8 channel sums of a group wherever they are needed); sums (V, 2, C) float64 [sum gz, sum gz*xhat].
constexpr int GN_CH = 8;
constexpr int GN_BWD_SLOTS = 8;        // copies of the backward sums the workgroups spread their float64 atomics over

// mode 0: stats += [x, x^2];  mode 1: sums += [gz, gz*xhat], gz = g * [gamma*xhat+beta > 0] (when relu)
template <int MODE>
__global__ void __launch_bounds__(256)
gn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ g, const double* __restrict__ stats,
                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu,
                 size_t hw, int cq, double* __restrict__ out) {
    __shared__ float red[4 * 32][8];                       // [wave][channel quad <= 32][sum, weighted sum]
    const int tid = threadIdx.x, v = blockIdx.y;
    const int c = (tid % cq) * 4, C = cq * 4;
    const size_t n4 = hw * cq;
    const float* xv = x + (size_t)v * n4 * 4;
    const float* gv = MODE ? g + (size_t)v * n4 * 4 : nullptr;
    float mean = 0.f, inv = 1.f, ga[4] = {1, 1, 1, 1}, be[4] = {0, 0, 0, 0};
    if (MODE) {
        const int c0 = c & ~(GN_CH - 1);
        double s = 0.0, q = 0.0;
        for (int k = 0; k < GN_CH; ++k) { s += stats[((size_t)v * 2) * C + c0 + k]; q += stats[((size_t)v * 2 + 1) * C + c0 + k]; }
        const double nn = (double)hw * GN_CH, mu = s / nn;
        double var = q / nn - mu * mu; if (var < 0.0) var = 0.0;
        mean = (float)mu; inv = (float)(1.0 / sqrt(var + (double)eps));
        for (int k = 0; k < 4; ++k) { ga[k] = gamma[c + k]; be[k] = beta[c + k]; }
    }
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    auto fold = [&](const float4 xx, const float4 g4) {
        const float vv[4] = {xx.x, xx.y, xx.z, xx.w};
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[k] += vv[k]; b[k] += vv[k] * vv[k]; }
        } else {
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (vv[k] - mean) * inv;
                const float gz = (!relu || ga[k] * xh + be[k] > 0.f) ? gg[k] : 0.f;
                a[k] += gz; b[k] += gz * xh;
            }
        }
    };
    // four elements per trip, their loads issued together: with one per trip a full-resolution layer was ~19 dependent trips of
    // two loads per thread on 384 workgroups -- latency-bound at 38 us per launch on average (round 6's trace of the training step)
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 xx[4], g4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xx[u] = ld4(xv + 4 * (i + u * stride));
            g4[u] = MODE ? ld4(gv + 4 * (i + u * stride)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) fold(xx[u], g4[u]);
    }
    for (; i < n4; i += stride) fold(ld4(xv + 4 * i), MODE ? ld4(gv + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f));
    // lanes l, l + cq, l + 2 cq, ... of a wave hold the same channel quad (cq divides 64): butterfly over the offsets >= cq, then
    // the four waves' rows through LDS (with 8 channels the old tree had TWO threads walk 128 rows each: +5 us per launch)
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int o = 32; o >= cq; o >>= 1) { a[k] += __shfl_xor(a[k], o, 64); b[k] += __shfl_xor(b[k], o, 64); }
    if ((tid & 63) < cq) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[(tid >> 6) * 32 + (tid & 63)][k] = a[k]; red[(tid >> 6) * 32 + (tid & 63)][4 + k] = b[k]; }
    }
    __syncthreads();
    if (tid < cq) {
        double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) { sa[k] += red[w * 32 + tid][k]; sb[k] += red[w * 32 + tid][4 + k]; }
        // MODE 1: the partial sums go to one of GN_BWD_SLOTS copies of `out`, which the apply pass adds up as it reads them.
        // Atomics on ONE address are performed one after the other by the L2, ~40 ns each (tools/r6_gn_reduce_probe.py: a
        // launch of 384 workgroups cost 15 us more than one of 128 whatever the tensor's size): round 6 first let every
        // workgroup add to per-layer totals as well, then take a ticket so that the last one would fold -- either way 384
        // serialised atomics, 30-52 us per launch in the training step against 5-23 us for the element-wise pass over the
        // same tensors.  No cross-workgroup step is left here: 16 atomics per address and slot.
        double* dst = out + (MODE ? (size_t)(blockIdx.x % GN_BWD_SLOTS) * gridDim.y * 2 * C : 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            atomicAdd(dst + ((size_t)v * 2) * C + c + k, sa[k]);
            atomicAdd(dst + ((size_t)v * 2 + 1) * C + c + k, sb[k]);
        }
    }
}

// mode 0: y = act(gamma*xhat + beta);  mode 1: dx = inv * (gamma*gz - mean_g(gamma*gz) - xhat * mean_g(gamma*gz*xhat))
template <int MODE>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ g, const double* __restrict__ stats,
                const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
                float eps, int relu, size_t hw, int cq, float* __restrict__ out, double* __restrict__ tot = nullptr) {
    const int tid = threadIdx.x, v = blockIdx.y;
    const int c = (tid % cq) * 4, C = cq * 4;
    const size_t n4 = hw * cq;
    const int c0 = c & ~(GN_CH - 1);
    const size_t plane = (size_t)gridDim.y * 2 * C;       // one slot of the backward sums: (V, 2, C)
    if (MODE && tot && blockIdx.x == 0 && blockIdx.y == 0) {
        // (2, C) over all views and slots: d beta, d gamma -- one workgroup, one thread per (statistic, channel); C <= 128
        if (tid < 2 * C) {
            double acc = 0.0;
            for (unsigned vv = 0; vv < gridDim.y; ++vv)
#pragma unroll
                for (int sl = 0; sl < GN_BWD_SLOTS; ++sl) acc += sums[sl * plane + (size_t)vv * 2 * C + tid];
            tot[tid] += acc;
        }
    }
    // the slots of this view's backward sums, added up ONCE per workgroup (one thread per (statistic, channel), C <= 128), not by
    // every thread for its own group (128 float64 loads per thread: the pass took 37 instead of 15 us on a 240 x 320 x 16 layer)
    __shared__ double folded[2 * 128];
    if (MODE) {
        if (tid < 2 * C) {
            double acc = 0.0;
#pragma unroll
            for (int sl = 0; sl < GN_BWD_SLOTS; ++sl) acc += sums[sl * plane + (size_t)v * 2 * C + tid];
            folded[tid] = acc;
        }
        __syncthreads();
    }
    double s = 0.0, q = 0.0, ta = 0.0, tb = 0.0;
    for (int k = 0; k < GN_CH; ++k) {
        s += stats[((size_t)v * 2) * C + c0 + k]; q += stats[((size_t)v * 2 + 1) * C + c0 + k];
        if (MODE) {
            ta += (double)gamma[c0 + k] * folded[c0 + k];
            tb += (double)gamma[c0 + k] * folded[C + c0 + k];
        }
    }
    const double nn = (double)hw * GN_CH, mu = s / nn;
    double var = q / nn - mu * mu; if (var < 0.0) var = 0.0;
    const float mean = (float)mu, inv = (float)(1.0 / sqrt(var + (double)eps));
    const float m1 = (float)(ta / nn), m2 = (float)(tb / nn);
    float ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ga[k] = gamma[c + k]; be[k] = beta[c + k]; }
    const float* xv = x + (size_t)v * n4 * 4;
    const float* gv = MODE ? g + (size_t)v * n4 * 4 : nullptr;
    float* ov = out + (size_t)v * n4 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 xx = ld4(xv + 4 * i);
        const float vv[4] = {xx.x, xx.y, xx.z, xx.w};
        float o[4];
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = ga[k] * ((vv[k] - mean) * inv) + be[k];
                o[k] = relu ? fmaxf(z, 0.f) : z;
            }
        } else {
            const float4 g4 = ld4(gv + 4 * i);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (vv[k] - mean) * inv;
                const float gz = (!relu || ga[k] * xh + be[k] > 0.f) ? gg[k] : 0.f;
                o[k] = inv * (ga[k] * gz - m1 - xh * m2);
            }
        }
        st4(ov + 4 * i, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// tf.train.RMSPropOptimizer (decay 0.9, momentum 0, epsilon 1e-10, not centered; its `rms` slot starts
// at ONE):  ms += (g*g - ms) * (1 - decay);  mom = momentum*mom + lr * g / sqrt(ms + eps);  w -= mom.
// One launch over the flat parameter buffer (all variables of the model are views into it).
__global__ void __launch_bounds__(256)
rmsprop_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ ms,
               float* __restrict__ mom, size_t n, float lr, float decay, float momentum, float eps,
               float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        float m = ms[i] + (gi * gi - ms[i]) * (1.0f - decay);
        float mo = momentum * mom[i] + lr * gi / sqrtf(m + eps);
        ms[i] = m; mom[i] = mo; w[i] -= mo;
    }
}

// tf.train.MomentumOptimizer (train.py:262-263): accum = momentum*accum + g; w -= lr*accum.
__global__ void __launch_bounds__(256)
momentum_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ accum, size_t n, float lr,
                float momentum, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float a = momentum * accum[i] + g[i] * grad_scale;
        accum[i] = a; w[i] -= lr * a;
    }
}

// tf.train.AdamOptimizer (train.py:266): m, v moments; w -= lr_t * m / (sqrt(v) + eps) with
// lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) formed by the caller.
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            size_t n, float lr_t, float beta1, float beta2, float eps, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i] * grad_scale;
        float mi = m[i] + (gi - m[i]) * (1.0f - beta1);
        float vi = v[i] + (gi * gi - v[i]) * (1.0f - beta2);
        m[i] = mi; v[i] = vi; w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

inline int grid_for(size_t n4) { size_t b = (n4 + 255) / 256; return (int)(b < 4096 ? (b ? b : 1) : 4096); }

}  // namespace

extern "C" int mvs_softargmin_bwd_f32(const float* reg, const float* g_depth, const float* g_prob, int D, int H, int W,
                                      float depth_start, float depth_interval, int inverse_depth,
                                      float* g_reg, void* stream) {
    MVS_CHECK_ARG(reg && (g_depth || g_prob) && g_reg && D > 0 && H > 0 && W > 0);
    const int HW = H * W;
    softargmin_bwd_kernel<<<mvs_cdiv(HW, 256), 256, 0, mvs_stream(stream)>>>(
        reg, g_depth, g_prob, D, HW, depth_start, depth_interval, inverse_depth, g_reg);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_bn_bwd_sum_slots(void) { return BN_BWD_SLOTS; }

extern "C" int mvs_bn_relu_f32(const float* y, const float* scale, const float* shift, const float* y2,
                               const float* scale2, const float* shift2, size_t voxels, int C, float* out,
                               void* stream) {
    MVS_CHECK_ARG(y && out && voxels > 0 && C > 0);
    if (C % 4 || 256 % (C / 4)) return MVS_E_SHAPE;
    MVS_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (scale2 == nullptr) == (shift2 == nullptr));
    const size_t n4 = voxels * (size_t)(C / 4);
    bn_relu_kernel<<<grid_for(n4), 256, 0, mvs_stream(stream)>>>(y, scale, shift, y2, scale2, shift2, n4, C / 4, out);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_bn_bwd_reduce_f32(const float* y, const double* stats, double count, float eps,
                                     const float* scale, const float* shift, const float* g1,
                                     const float* g2, size_t voxels, int C, double* sums, void* stream) {
    MVS_CHECK_ARG(y && stats && scale && shift && g1 && sums && voxels > 0 && C > 0 && count > 0);
    if (C % 4 || 256 % (C / 4)) return MVS_E_SHAPE;
    const size_t n4 = voxels * (size_t)(C / 4);
    int grid = grid_for(n4); if (grid > 1024) grid = 1024;
    bn_bwd_reduce_kernel<<<grid, 256, 0, mvs_stream(stream)>>>(y, stats, count, eps, scale, shift, g1, g2, n4, C / 4, sums);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_bn_bwd_apply_f32(const float* y, const double* stats, double count, float eps,
                                    const float* scale, const float* shift, const float* gamma,
                                    const float* g1, const float* g2, const double* sums, size_t voxels,
                                    int C, float* g_y, float* g_gamma, float* g_beta, void* stream) {
    MVS_CHECK_ARG(y && stats && scale && shift && gamma && g1 && sums && g_y && voxels > 0 && C > 0 && count > 0);
    if (C % 4 || 256 % (C / 4) || C > 64) return MVS_E_SHAPE;
    const size_t n4 = voxels * (size_t)(C / 4);
    bn_bwd_apply_kernel<<<grid_for(n4), 256, 0, mvs_stream(stream)>>>(y, stats, count, eps, scale, shift, gamma, g1, g2,
                                                                      sums, n4, C / 4, g_y, g_gamma, g_beta);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_cost_volume_bwd_f32(const float* ref, const float* src, const float* transforms,
                                       int view_num, int depth_num, int H, int W, int C, const float* g1,
                                       const float* g2, float* g_ref, float* g_src, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && g1 && g_ref && g_src && view_num >= 2 && depth_num > 0 && H > 0 && W > 0);
    if (C % 4 || view_num - 1 > CVB_MAX_SRC) return MVS_E_SHAPE;
    if ((long long)H * W * C >= (1LL << 31)) return MVS_E_SHAPE;
    hipStream_t st = mvs_stream(stream);
    const long long total = (long long)H * W * (C / 4);
    const int bx = mvs_cdiv(total, 256);
    int ppb = depth_num;                              // long runs along depth keep the pending cells alive
    while (ppb > 16 && (long long)bx * mvs_cdiv(depth_num, ppb) < 2048) ppb = (ppb + 1) / 2;
    dim3 grid(bx, mvs_cdiv(depth_num, ppb));
#define CVB_LAUNCH(NS) cost_volume_bwd_kernel<NS><<<grid, 256, 0, st>>>(ref, src, transforms, depth_num, ppb, H, W, C, \
                                                                        g1, g2, g_ref, g_src)
    switch (view_num - 1) {
        case 1: CVB_LAUNCH(1); break; case 2: CVB_LAUNCH(2); break; case 3: CVB_LAUNCH(3); break;
        case 4: CVB_LAUNCH(4); break; case 5: CVB_LAUNCH(5); break; case 6: CVB_LAUNCH(6); break;
        case 7: CVB_LAUNCH(7); break; default: CVB_LAUNCH(8); break;
    }
#undef CVB_LAUNCH
    MVS_LAUNCH_RET();
}

extern "C" int mvs_rmsprop_step_f32(float* w, const float* g, float* ms, float* mom, size_t n, float lr,
                                    float decay, float momentum, float eps, float grad_scale, void* stream) {
    MVS_CHECK_ARG(w && g && ms && mom && n > 0);
    rmsprop_kernel<<<grid_for(n), 256, 0, mvs_stream(stream)>>>(w, g, ms, mom, n, lr, decay, momentum, eps, grad_scale);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_momentum_step_f32(float* w, const float* g, float* accum, size_t n, float lr, float momentum,
                                     float grad_scale, void* stream) {
    MVS_CHECK_ARG(w && g && accum && n > 0);
    momentum_kernel<<<grid_for(n), 256, 0, mvs_stream(stream)>>>(w, g, accum, n, lr, momentum, grad_scale);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_adam_step_f32(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float beta1,
                                 float beta2, float eps, float grad_scale, void* stream) {
    MVS_CHECK_ARG(w && g && m && v && n > 0);
    adam_kernel<<<grid_for(n), 256, 0, mvs_stream(stream)>>>(w, g, m, v, n, lr_t, beta1, beta2, eps, grad_scale);
    MVS_LAUNCH_RET();
}

// The inference kernels' GroupNorm sums -- (V, C/8, slots, 2) float64 partial [sum, sumsq] per 8-channel group (csrc/unet2d*.hip) --
// in the per-channel layout of the kernels above: every channel of a group carries an eighth of the group's totals, so the
// group moments folded from "the 8 channel sums" are the forward's own.  The training towers need no second pass over the
// activations for statistics the forward convolution already produced (round 6: 31 launches and 0.6 ms of a 6 ms step).
__global__ void gn_slots_to_channel_sums_kernel(const double* __restrict__ slots, int V, int C, int nslot, double* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (v, k, c)
    if (i >= V * 2 * C) return;
    const int c = i % C, k = (i / C) & 1, v = i / (2 * C);
    const double* p = slots + (((size_t)v * (C / GN_CH) + c / GN_CH) * nslot) * 2 + k;
    double t = 0.0;
    for (int s_ = 0; s_ < nslot; ++s_) t += p[2 * s_];
    stats[i] = t * 0.125;
}

extern "C" int mvs_gn_slots_to_channel_sums_f64(const double* slots, int V, int C, int nslot, double* stats, void* stream) {
    MVS_CHECK_ARG(slots && stats && V > 0 && C > 0 && nslot > 0);
    if (C % GN_CH) return MVS_E_SHAPE;
    gn_slots_to_channel_sums_kernel<<<mvs_cdiv((long long)V * 2 * C, 256), 256, 0, mvs_stream(stream)>>>(slots, V, C, nslot, stats);
    MVS_LAUNCH_RET();
}

// All layers of a tower in one launch: layer i's slots start `slot_off[i]` float64 behind `slots`, its (V, 2, C_i) statistics
// `stat_off[i]` behind `stats` (the jobs ride in the kernel arguments; blockIdx.y = layer).
constexpr int GN_MANY_MAX = 64;
struct GnManyJobs { long long slot_off[GN_MANY_MAX], stat_off[GN_MANY_MAX]; int C[GN_MANY_MAX]; };

__global__ void gn_slots_to_channel_sums_many_kernel(const double* __restrict__ slots, int V, int nslot, double* __restrict__ stats,
                                                     GnManyJobs jobs) {
    const int C = jobs.C[blockIdx.y];
    const double* sl = slots + jobs.slot_off[blockIdx.y];
    double* out = stats + jobs.stat_off[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < V * 2 * C; i += gridDim.x * blockDim.x) {
        const int c = i % C, k = (i / C) & 1, v = i / (2 * C);
        const double* p = sl + (((size_t)v * (C / GN_CH) + c / GN_CH) * nslot) * 2 + k;
        double t = 0.0;
        for (int s_ = 0; s_ < nslot; ++s_) t += p[2 * s_];
        out[i] = t * 0.125;
    }
}

extern "C" int mvs_gn_slots_to_channel_sums_many_f64(int n, const double* slots, const long long* slot_off, const int* C, int V,
                                                     int nslot, double* stats, const long long* stat_off, void* stream) {
    MVS_CHECK_ARG(n > 0 && slots && slot_off && C && stats && stat_off && V > 0 && nslot > 0);
    for (int i = 0; i < n; ++i) {
        MVS_CHECK_ARG(C[i] > 0 && slot_off[i] >= 0 && stat_off[i] >= 0);
        if (C[i] % GN_CH) return MVS_E_SHAPE;
    }
    for (int first = 0; first < n; first += GN_MANY_MAX) {
        const int m = n - first < GN_MANY_MAX ? n - first : GN_MANY_MAX;
        GnManyJobs jobs;
        int cmax = 1;
        for (int k = 0; k < m; ++k) {
            jobs.slot_off[k] = slot_off[first + k]; jobs.stat_off[k] = stat_off[first + k]; jobs.C[k] = C[first + k];
            if (C[first + k] > cmax) cmax = C[first + k];
        }
        hipLaunchKernelGGL(gn_slots_to_channel_sums_many_kernel, dim3(mvs_cdiv((long long)V * 2 * cmax, 256), m), dim3(256), 0,
                           mvs_stream(stream), slots, V, nslot, stats, jobs);
    }
    MVS_LAUNCH_RET();
}

// GroupNorm entry points: mode selects the pass (see the kernels above).
static int gn_check(const void* x, int V, size_t hw, int C) {
    if (!x || V <= 0 || hw == 0 || C <= 0) return MVS_E_BADARG;
    if (C % GN_CH || 256 % (C / 4) || C > 128) return MVS_E_SHAPE;     // a wave's lanes cover whole rows of C / 4 quads; LDS rows for <= 32 quads
    return 0;
}
// Workgroups per view of the reductions (each ends with float64 atomics on shared cache lines: see GN_BWD_SLOTS).
#ifndef GN_REDUCE_BLOCKS
#define GN_REDUCE_BLOCKS 128
#endif

static dim3 gn_grid(size_t hw, int C, int V, int cap) {
    size_t b = (hw * (size_t)(C / 4) + 255) / 256;
    return dim3((unsigned)(b < (size_t)cap ? (b ? b : 1) : cap), V);
}

extern "C" int mvs_gn_stats_f32(const float* x, int V, size_t hw, int C, double* stats, void* stream) {
    int rc = gn_check(x, V, hw, C); if (rc) return rc;
    MVS_CHECK_ARG(stats);
    gn_reduce_kernel<0><<<gn_grid(hw, C, V, GN_REDUCE_BLOCKS), 256, 0, mvs_stream(stream)>>>(x, nullptr, nullptr, nullptr, nullptr, 0.f, 0,
                                                                                hw, C / 4, stats);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_gn_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                                int relu, int V, size_t hw, int C, float* y, void* stream) {
    int rc = gn_check(x, V, hw, C); if (rc) return rc;
    MVS_CHECK_ARG(stats && gamma && beta && y);
    gn_apply_kernel<0><<<gn_grid(hw, C, V, 2048), 256, 0, mvs_stream(stream)>>>(x, nullptr, stats, nullptr, gamma, beta, eps, relu,
                                                                                hw, C / 4, y);
    MVS_LAUNCH_RET();
}

// `sums`: mvs_gn_bwd_sums_doubles(V, C) float64, zeroed by the caller: mvs_gn_bwd_sum_slots() copies of (V, 2, C) that the
// workgroups spread their atomics over; the apply pass (and whoever wants d gamma / d beta) adds the copies up.
extern "C" int mvs_gn_bwd_sum_slots(void) { return GN_BWD_SLOTS; }
extern "C" size_t mvs_gn_bwd_sums_doubles(int V, int C) {
    return (V > 0 && C > 0) ? (size_t)GN_BWD_SLOTS * V * 2 * C : 0;
}

extern "C" int mvs_gn_bwd_reduce_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                                     int relu, const float* g, int V, size_t hw, int C, double* sums, void* stream) {
    int rc = gn_check(x, V, hw, C); if (rc) return rc;
    MVS_CHECK_ARG(stats && gamma && beta && g && sums);
    gn_reduce_kernel<1><<<gn_grid(hw, C, V, GN_REDUCE_BLOCKS), 256, 0, mvs_stream(stream)>>>(x, g, stats, gamma, beta, eps, relu, hw, C / 4, sums);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_gn_bwd_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                                    int relu, const float* g, const double* sums, int V, size_t hw, int C, float* dx,
                                    void* stream) {
    int rc = gn_check(x, V, hw, C); if (rc) return rc;
    MVS_CHECK_ARG(stats && gamma && beta && g && sums && dx);
    gn_apply_kernel<1><<<gn_grid(hw, C, V, 2048), 256, 0, mvs_stream(stream)>>>(x, g, stats, sums, gamma, beta, eps, relu,
                                                                                hw, C / 4, dx);
    MVS_LAUNCH_RET();
}

// The same, and the sums over ALL views and slots ADDED to totals (2, C) float64 [d beta, d gamma] by the first workgroup: the
// parameter gradients without a reduction launch per layer.  C <= 128.
extern "C" int mvs_gn_bwd_apply_tot_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                                        int relu, const float* g, const double* sums, double* totals, int V, size_t hw, int C,
                                        float* dx, void* stream) {
    int rc = gn_check(x, V, hw, C); if (rc) return rc;
    MVS_CHECK_ARG(stats && gamma && beta && g && sums && totals && dx);
    gn_apply_kernel<1><<<gn_grid(hw, C, V, 2048), 256, 0, mvs_stream(stream)>>>(x, g, stats, sums, gamma, beta, eps, relu,
                                                                                hw, C / 4, dx, totals);
    MVS_LAUNCH_RET();
}

// Deterministic two-pass variant (see cvb_pass1_kernel).  Workspace: GW (N-1, D, H, W, C) + chunk rows.
static void cvg_layout(int n_src, int D, int H, int W, int C, size_t* gw, size_t* refp, size_t* srcp) {
    const size_t img = (size_t)H * W * C;
    *gw = (size_t)n_src * D * img;
    *refp = (size_t)mvs_cdiv(D, CVG_CH1) * img;
    *srcp = (size_t)mvs_cdiv(D, CVG_CH2) * n_src * img;
}

extern "C" size_t mvs_cost_volume_bwd_workspace_bytes(int view_num, int depth_num, int H, int W, int C) {
    if (view_num < 2 || depth_num <= 0 || H <= 0 || W <= 0 || (C != 32 && C != 16)) return 0;
    size_t a, b, c;
    cvg_layout(view_num - 1, depth_num, H, W, C, &a, &b, &c);
    return (a + b + c) * sizeof(float);
}

extern "C" int mvs_cost_volume_bwd_gather_f32(const float* ref, const float* src, const float* transforms,
                                              int view_num, int depth_num, int H, int W, int C, const float* g1,
                                              const float* g2, void* workspace, size_t workspace_bytes,
                                              float* g_ref, float* g_src, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && g1 && g_ref && g_src && workspace && view_num >= 2 && depth_num > 0 && H > 0 && W > 0);
    if ((C != 32 && C != 16) || view_num - 1 > CVB_MAX_SRC) return MVS_E_SHAPE;
    if ((long long)H * W * C >= (1LL << 31)) return MVS_E_SHAPE;
    const int n_src = view_num - 1;
    size_t ngw, nref, nsrc;
    cvg_layout(n_src, depth_num, H, W, C, &ngw, &nref, &nsrc);
    if (workspace_bytes < (ngw + nref + nsrc) * sizeof(float)) return MVS_E_WORKSPACE;
    float* gw = (float*)workspace; float* refp = gw + ngw; float* srcp = refp + nref;
    hipStream_t st = mvs_stream(stream);
    const size_t img = (size_t)H * W * C;
    {
        dim3 grid(mvs_cdiv((long long)H * W * (C / 4), 256), mvs_cdiv(depth_num, CVG_CH1));
#define CVG1(NS) cvb_pass1_kernel<NS><<<grid, 256, 0, st>>>(ref, src, transforms, depth_num, H, W, C, g1, g2, gw, refp)
        switch (n_src) {
            case 1: CVG1(1); break; case 2: CVG1(2); break; case 3: CVG1(3); break; case 4: CVG1(4); break;
            case 5: CVG1(5); break; case 6: CVG1(6); break; case 7: CVG1(7); break; default: CVG1(8); break;
        }
#undef CVG1
    }
    {
        dim3 grid(mvs_cdiv((long long)H * W, 256), mvs_cdiv(depth_num, CVG_CH2), n_src);
        if (C == 32) cvb_pass2_kernel<8><<<grid, 256, 0, st>>>(transforms, depth_num, H, W, gw, srcp);
        else cvb_pass2_kernel<4><<<grid, 256, 0, st>>>(transforms, depth_num, H, W, gw, srcp);
    }
    cvb_fold_kernel<<<mvs_cdiv((long long)(img / 4), 256), 256, 0, st>>>(refp, mvs_cdiv(depth_num, CVG_CH1), img / 4, g_ref);
    cvb_fold_kernel<<<mvs_cdiv((long long)(n_src * img / 4), 256), 256, 0, st>>>(srcp, mvs_cdiv(depth_num, CVG_CH2), n_src * img / 4, g_src);
    MVS_LAUNCH_RET();
}
