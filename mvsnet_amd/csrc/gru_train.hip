// Training of the ConvGRU regulariser (mvsnet/model.py:505-599, mvsnet/convgru.py:82-122): the part of
// back-propagation through time that is sequential in the plane index, for ONE cell over all D planes.
//
// A cell's two convolutions act on concat([x, h]) and concat([x, r*h]).  Their x parts (and biases) do not depend on
// the recurrence: the host computes them for all planes as one batched convolution, px (D,H,W,3F) = [gates | candidate].
// What is left per plane is three launches forward
//     g = px_g + conv3x3(h, Wgh)                      + LayerNorm moments of g_r, g_u           gt_gates_kernel
//     c = px_c + conv3x3(r*h, Woh), r = sigmoid(LN g_r) + moments of c, r*h kept                gt_out_kernel
//     h' = u*h + (1-u)*tanh(LN c),  u = sigmoid(LN g_u)                                         gt_blend_kernel
// (the blend is folded into the staging of the next plane's gate convolution: two launches per plane) and three backward (plane D-1 down to 0), with every activation recomputed from the kept raw convolutions g, c:
//     dz_c, dz_u, dh = dh'*u                          + per-channel sums of the two LayerNorms  gt_bwd_blend_kernel
//     dc = LN'(dz_c) [kept: gradient of px_c]; d(rh) = conv3x3(dc, Woh^T); dh += d(rh)*r; dz_r  gt_bwd_out_kernel
//     dg = LN'(dz_r | dz_u) [kept: gradient of px_g]; dh += conv3x3(dg, Wgh^T)                  gt_bwd_gates_kernel
// (the blend backward of plane d-1 is folded into the epilogue of plane d's gate kernel: two launches per plane).
// LN'(dz) = inv_std * (gamma*dz - mean(gamma*dz) - xhat * mean(gamma*dz*xhat)), the means over all H*W*F elements of
// the plane; the per-channel sums A[f] = sum dz, B[f] = sum dz*xhat are kept per plane (they are the gradients of beta
// and gamma).  Everything that is NOT sequential -- x-part input gradients, all weight and bias gradients -- is a
// batched convolution over the kept px-gradients, done by the host (gru_train.py).
// Convolutions here are 3x3 over <= 32 channels on a 160 x 120 grid: launch-bound VALU kernels (a workgroup stages an
// 18-wide halo tile in LDS, weights through the scalar cache, output channels split over NS wave-uniform groups).
#include "common.h"

namespace {

constexpr int SLOTS_F = 8;     // forward LayerNorm moment slots per plane (float64 atomics on one address serialise)
constexpr int SLOTS_B = 16;    // backward per-channel sum slots per plane and LayerNorm
constexpr int TW = 16, PW = TW + 2;
typedef const __attribute__((address_space(4))) float cfloat;

template <int F> struct Geo {
    static constexpr int NS = F >= 16 ? 4 : (F >= 8 ? 2 : 1);     // output-channel groups per workgroup
    static constexpr int TH = 16 / NS, PH = TH + 2, NPIX = 256 / NS;
};
template <int CIN> constexpr int lds_stride() { return CIN >= 8 ? CIN + 4 : CIN; }

// hardware exp / reciprocal (1 ulp): forward and backward evaluate the gates with the same expressions
__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f); }

template <int N>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&o)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int i = 0; i < N / 4; ++i) { float4 t = *(const float4*)(p + 4 * i); o[4*i] = t.x; o[4*i+1] = t.y; o[4*i+2] = t.z; o[4*i+3] = t.w; }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { float2 t = *(const float2*)(p + 2 * i); o[2*i] = t.x; o[2*i+1] = t.y; }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = p[i];
    }
}

// LayerNorm moments of norm k (0 reset, 1 update, 2 candidate) of one plane, folded over the slots
__device__ __forceinline__ void ln_moments(const double* __restrict__ st, int k, double n, float& mean, float& is) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < SLOTS_F; ++i) { s += st[i * 6 + 2 * k]; q += st[i * 6 + 2 * k + 1]; }
    const double m = s / n;
    double v = q / n - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    is = (float)(1.0 / sqrt(v + 1e-12));                            // tf.contrib.layers.layer_norm: variance_epsilon 1e-12
}

// 3x3 taps of one pixel out of the staged tile: acc[j] += sum tile[.., ci] * w[tap][ci][co0 + j]
template <int CIN, int COT, int CPT>
__device__ __forceinline__ void conv_taps(const float* tile, int ly, int lx, cfloat* w, int co0, float (&acc)[CPT]) {
    constexpr int LC = lds_stride<CIN>();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            float v[CIN];
            load_vec<CIN>(tile + ((ly + kh) * PW + lx + kw) * LC, v);
            cfloat* wt = w + (kh * 3 + kw) * CIN * COT + co0;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int j = 0; j < CPT; ++j) acc[j] += v[ci] * wt[ci * COT + j];
        }
}

struct TileIdx { int y0, x0, ly, lx, py, px, co_group; bool valid; };
template <int F>
__device__ __forceinline__ TileIdx tile_index(int H, int W) {
    using G = Geo<F>;
    TileIdx t;
    const int tiles_x = (W + TW - 1) / TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    t.y0 = ty * G::TH; t.x0 = tx * TW;
    const int pi = threadIdx.x % G::NPIX;
    t.co_group = __builtin_amdgcn_readfirstlane(threadIdx.x / G::NPIX);
    t.ly = pi >> 4; t.lx = pi & 15;
    t.py = t.y0 + t.ly; t.px = t.x0 + t.lx;
    t.valid = t.py < H && t.px < W;
    return t;
}

// ---- forward ---------------------------------------------------------------------------------------------------
// Staging is dealt out in (position, channel quad) units over all 256 threads; the LayerNorm parameters sit in LDS.
template <int F> struct Unit { static constexpr int Q = F >= 4 ? 4 : F, NQ = F / Q; };

template <int F>
__device__ __forceinline__ void stage_ln(const float* __restrict__ ln, float* lnl) {
    if (threadIdx.x < 6 * F) lnl[threadIdx.x] = ln[threadIdx.x];
    __syncthreads();
}

template <int F>
__global__ void __launch_bounds__(256)
gt_gates_kernel(const float* __restrict__ px, const float* __restrict__ hsrc, const float* __restrict__ cprev,
                const float* __restrict__ gprev, const double* __restrict__ stats_prev, const float* __restrict__ ln,
                const float* __restrict__ wgh, int H, int W, float* __restrict__ hcur, float* __restrict__ g,
                double* __restrict__ stats) {
    using G = Geo<F>;
    using U = Unit<F>;
    constexpr int CO = 2 * F, CPT = CO / G::NS, LC = lds_stride<F>(), Q = U::Q, NQ = U::NQ;
    __shared__ __attribute__((aligned(16))) float tile[G::PH * PW * LC];
    __shared__ float lnl[6 * F];
    __shared__ float red[4][4];
    const TileIdx t = tile_index<F>(H, W);
    // the state entering this plane is the previous plane's blend h = u*h' + (1-u)*tanh(LN c) (convgru.py:98,102,
    // 114-120), evaluated while staging (halo positions are recomputed by the neighbouring tiles; the tile's own
    // pixels are written out: the candidate convolution and the backward pass read them).  Plane 0: hcur is given.
    float mu = 0.f, isu = 0.f, mc = 0.f, isc = 0.f;
    if (cprev) {
        ln_moments(stats_prev, 1, (double)H * W * F, mu, isu);
        ln_moments(stats_prev, 2, (double)H * W * F, mc, isc);
    }
    stage_ln<F>(ln, lnl);
    for (int ui = threadIdx.x; ui < G::PH * PW * NQ; ui += 256) {
        const int f = ui / NQ, ch0 = (ui - f * NQ) * Q;
        const int r = f / PW, c = f - r * PW, gy = t.y0 - 1 + r, gx = t.x0 - 1 + c;
        float v[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t q = (size_t)gy * W + gx;
            if (cprev) {
                float cv[Q], gu[Q];
                load_vec<Q>(hsrc + q * F + ch0, v);
                load_vec<Q>(cprev + q * F + ch0, cv);
                load_vec<Q>(gprev + q * 2 * F + F + ch0, gu);
#pragma unroll
                for (int i = 0; i < Q; ++i) {
                    const float u = sigm(lnl[2 * F + ch0 + i] * ((gu[i] - mu) * isu) + lnl[3 * F + ch0 + i]);
                    const float y = tanh_fast(lnl[4 * F + ch0 + i] * ((cv[i] - mc) * isc) + lnl[5 * F + ch0 + i]);
                    v[i] = u * v[i] + (1.0f - u) * y;
                }
                if (r >= 1 && r <= G::TH && c >= 1 && c <= TW) {
#pragma unroll
                    for (int i = 0; i < Q; ++i) hcur[q * F + ch0 + i] = v[i];
                }
            } else {
                load_vec<Q>(hcur + q * F + ch0, v);
            }
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) tile[f * LC + ch0 + i] = v[i];
    }
    __syncthreads();
    const int co0 = t.co_group * CPT;
    float acc[CPT];
    const size_t p = (size_t)t.py * W + t.px;
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = t.valid ? px[p * 3 * F + co0 + j] : 0.f;
    conv_taps<F, CO, CPT>(tile, t.ly, t.lx, (cfloat*)wgh, co0, acc);
    float s[4] = {0.f, 0.f, 0.f, 0.f};                             // sum, sum of squares: reset part, update part
    if (t.valid) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            g[p * CO + co0 + j] = acc[j];
            const int k = (co0 + j) < F ? 0 : 2;
            s[k] += acc[j]; s[k + 1] += acc[j] * acc[j];
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float v = wave_sum(s[k]); if (lane == 0) red[wv][k] = v; }
    __syncthreads();
    if (threadIdx.x < 4)
        atomicAdd(&stats[(blockIdx.x % SLOTS_F) * 6 + threadIdx.x],
                  (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
}

template <int F>
__global__ void __launch_bounds__(256)
gt_out_kernel(const float* __restrict__ px, const float* __restrict__ hprev, const float* __restrict__ g,
              double* __restrict__ stats, const float* __restrict__ woh, const float* __restrict__ ln,
              int H, int W, float* __restrict__ rh, float* __restrict__ c) {
    using G = Geo<F>;
    using U = Unit<F>;
    constexpr int CPT = F / G::NS, LC = lds_stride<F>(), Q = U::Q, NQ = U::NQ;
    __shared__ __attribute__((aligned(16))) float tile[G::PH * PW * LC];
    __shared__ float lnl[6 * F];
    __shared__ float red[4][2];
    const TileIdx t = tile_index<F>(H, W);
    float mean, is;
    ln_moments(stats, 0, (double)H * W * F, mean, is);
    stage_ln<F>(ln, lnl);
    for (int ui = threadIdx.x; ui < G::PH * PW * NQ; ui += 256) {
        const int f = ui / NQ, ch0 = (ui - f * NQ) * Q;
        const int r = f / PW, cc = f - r * PW, gy = t.y0 - 1 + r, gx = t.x0 - 1 + cc;
        float v[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t q = (size_t)gy * W + gx;
            float gr[Q];
            load_vec<Q>(hprev + q * F + ch0, v);
            load_vec<Q>(g + q * 2 * F + ch0, gr);
#pragma unroll
            for (int i = 0; i < Q; ++i) v[i] *= sigm(lnl[ch0 + i] * ((gr[i] - mean) * is) + lnl[F + ch0 + i]);   // convgru.py:97,101,107
            if (r >= 1 && r <= G::TH && cc >= 1 && cc <= TW) {
#pragma unroll
                for (int i = 0; i < Q; ++i) rh[q * F + ch0 + i] = v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) tile[f * LC + ch0 + i] = v[i];
    }
    __syncthreads();
    const int co0 = t.co_group * CPT;
    float acc[CPT];
    const size_t p = (size_t)t.py * W + t.px;
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = t.valid ? px[p * 3 * F + 2 * F + co0 + j] : 0.f;
    conv_taps<F, F, CPT>(tile, t.ly, t.lx, (cfloat*)woh, co0, acc);
    float s0 = 0.f, s1 = 0.f;
    if (t.valid) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) { c[p * F + co0 + j] = acc[j]; s0 += acc[j]; s1 += acc[j] * acc[j]; }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (lane == 0) { red[wv][0] = s0; red[wv][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2)
        atomicAdd(&stats[(blockIdx.x % SLOTS_F) * 6 + 4 + threadIdx.x],
                  (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
}

template <int F>
__global__ void __launch_bounds__(256)
gt_blend_kernel(const float* __restrict__ c, const float* __restrict__ g, const float* __restrict__ hprev,
                const double* __restrict__ stats, const float* __restrict__ ln, int HW, float* __restrict__ hout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW * F) return;
    const int f = i % F, pix = i / F;
    float mu, isu, mc, isc;
    ln_moments(stats, 1, (double)HW * F, mu, isu);
    ln_moments(stats, 2, (double)HW * F, mc, isc);
    const float u = sigm(ln[2 * F + f] * ((g[(size_t)pix * 2 * F + F + f] - mu) * isu) + ln[3 * F + f]);   // :98,102
    const float y = tanh_fast(ln[4 * F + f] * ((c[i] - mc) * isc) + ln[5 * F + f]);                            // :114,117
    hout[i] = u * hprev[i] + (1.0f - u) * y;                                                               // :120
}

// ---- backward --------------------------------------------------------------------------------------------------
// sum over the lanes that hold the same channel (lane % F); lanes 0..F-1 hold the totals
template <int F>
__device__ __forceinline__ float channel_sum(float v) {
#pragma unroll
    for (int o = 32; o >= F; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int F>
__global__ void __launch_bounds__(256)
gt_bwd_blend_kernel(const float* __restrict__ gh, const float* __restrict__ dh_rec, const float* __restrict__ c,
                    const float* __restrict__ g, const float* __restrict__ hprev, const double* __restrict__ stats,
                    const float* __restrict__ ln, int HW, float* __restrict__ dzc, float* __restrict__ dzu,
                    float* __restrict__ dh_next, double* __restrict__ part) {
    __shared__ float red[4][4][F];
    const int f = threadIdx.x % F;                                  // 256 and the grid stride are multiples of F
    float mu, isu, mc, isc;
    ln_moments(stats, 1, (double)HW * F, mu, isu);
    ln_moments(stats, 2, (double)HW * F, mc, isc);
    const float gu_ = ln[2 * F + f], bu_ = ln[3 * F + f], gc_ = ln[4 * F + f], bc_ = ln[5 * F + f];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW * F; i += gridDim.x * 256) {
        const int pix = i / F;
        const float gn = (g[(size_t)pix * 2 * F + F + f] - mu) * isu, cn = (c[i] - mc) * isc;
        const float u = sigm(gu_ * gn + bu_), y = tanh_fast(gc_ * cn + bc_);
        const float d = dh_rec[i] + gh[i];
        const float vc = d * (1.0f - u) * (1.0f - y * y);
        const float vu = d * (hprev[i] - y) * u * (1.0f - u);
        dh_next[i] = d * u;
        dzc[i] = vc; dzu[i] = vu;
        s[0] += vc; s[1] += vc * cn; s[2] += vu; s[3] += vu * gn;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float v = channel_sum<F>(s[k]); if (lane < F) red[wv][k][lane] = v; }
    __syncthreads();
    if (threadIdx.x < 4 * F) {
        const int k = threadIdx.x / F, ff = threadIdx.x % F;
        const double tot = (double)red[0][k][ff] + (double)red[1][k][ff] + (double)red[2][k][ff] + (double)red[3][k][ff];
        const int norm = k < 2 ? 2 : 1, ab = k & 1;
        atomicAdd(&part[((size_t)(norm * SLOTS_B + blockIdx.x % SLOTS_B) * 2 + ab) * F + ff], tot);
    }
}

// mean(gamma*dz) and mean(gamma*dz*xhat) of LayerNorm `norm` from the per-channel slot sums; all threads get both
template <int F>
__device__ __forceinline__ void ln_bwd_means(const double* __restrict__ part, int norm, const float* __restrict__ gamma,
                                             double n, double* lds, float& m1, float& m2) {
    if (threadIdx.x < 2 * F) {
        const int ab = threadIdx.x / F, f = threadIdx.x % F;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < SLOTS_B; ++i) s += part[((size_t)(norm * SLOTS_B + i) * 2 + ab) * F + f];
        lds[threadIdx.x] = s * (double)gamma[f];
    }
    __syncthreads();
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int f = 0; f < F; ++f) { a += lds[f]; b += lds[F + f]; }
    m1 = (float)(a / n); m2 = (float)(b / n);
    __syncthreads();
}

template <int F>
__global__ void __launch_bounds__(256)
gt_bwd_out_kernel(const float* __restrict__ dzc, const float* __restrict__ c, const float* __restrict__ g,
                  const float* __restrict__ hprev, const double* __restrict__ stats, double* __restrict__ part,
                  const float* __restrict__ woh_t, const float* __restrict__ ln, int H, int W,
                  float* __restrict__ gpx, float* __restrict__ dh_next, float* __restrict__ dzr) {
    using G = Geo<F>;
    using U = Unit<F>;
    constexpr int CPT = F / G::NS, LC = lds_stride<F>(), Q = U::Q, NQ = U::NQ;
    __shared__ __attribute__((aligned(16))) float tile[G::PH * PW * LC];
    __shared__ float lnl[6 * F];
    __shared__ double fold[2 * F];
    __shared__ float red[4][2][CPT];
    const TileIdx t = tile_index<F>(H, W);
    const double n = (double)H * W * F;
    float mr, isr, mc, isc, m1, m2;
    ln_moments(stats, 0, n, mr, isr);
    ln_moments(stats, 2, n, mc, isc);
    ln_bwd_means<F>(part, 2, ln + 4 * F, n, fold, m1, m2);
    stage_ln<F>(ln, lnl);
    for (int ui = threadIdx.x; ui < G::PH * PW * NQ; ui += 256) {
        const int f = ui / NQ, ch0 = (ui - f * NQ) * Q;
        const int r = f / PW, cc = f - r * PW, gy = t.y0 - 1 + r, gx = t.x0 - 1 + cc;
        float v[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t q = (size_t)gy * W + gx;
            float cv[Q];
            load_vec<Q>(dzc + q * F + ch0, v);
            load_vec<Q>(c + q * F + ch0, cv);
#pragma unroll
            for (int i = 0; i < Q; ++i) v[i] = isc * (lnl[4 * F + ch0 + i] * v[i] - m1 - (cv[i] - mc) * isc * m2);
            if (r >= 1 && r <= G::TH && cc >= 1 && cc <= TW) {
#pragma unroll
                for (int i = 0; i < Q; ++i) gpx[q * 3 * F + 2 * F + ch0 + i] = v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) tile[f * LC + ch0 + i] = v[i];
    }
    __syncthreads();
    const int co0 = t.co_group * CPT;
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
    conv_taps<F, F, CPT>(tile, t.ly, t.lx, (cfloat*)woh_t, co0, acc);       // d(r*h)
    float sa[CPT], sb[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) { sa[j] = 0.f; sb[j] = 0.f; }
    if (t.valid) {
        const size_t p = (size_t)t.py * W + t.px;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int f = co0 + j;
            const float gn = (g[p * 2 * F + f] - mr) * isr;
            const float r = sigm(lnl[f] * gn + lnl[F + f]);
            dh_next[p * F + f] += acc[j] * r;
            const float vz = acc[j] * hprev[p * F + f] * r * (1.0f - r);
            dzr[p * F + f] = vz;
            sa[j] = vz; sb[j] = vz * gn;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float a = wave_sum(sa[j]), b = wave_sum(sb[j]);
        if (lane == 0) { red[wv][0][j] = a; red[wv][1][j] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * F) {
        const int ab = threadIdx.x / F, f = threadIdx.x % F;
        const int grp = f / CPT, j = f % CPT;
        constexpr int WPG = 4 / G::NS;                              // waves per output-channel group
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < WPG; ++w) tot += (double)red[grp * WPG + w][ab][j];
        atomicAdd(&part[((size_t)(0 * SLOTS_B + blockIdx.x % SLOTS_B) * 2 + ab) * F + f], tot);
    }
}

template <int F>
__global__ void __launch_bounds__(256)
gt_bwd_gates_kernel(const float* __restrict__ dzr, const float* __restrict__ dzu, const float* __restrict__ g,
                    const double* __restrict__ stats, const double* __restrict__ part,
                    const float* __restrict__ wgh_t, const float* __restrict__ ln, int H, int W,
                    float* __restrict__ gpx, const float* __restrict__ dh_next,
                    // blend backward of the plane below (null gh_p on plane 0): its inputs, and where its results go
                    const float* __restrict__ gh_p, const float* __restrict__ c_p, const float* __restrict__ g_p,
                    const float* __restrict__ h_pp, const double* __restrict__ stats_p, double* __restrict__ part_p,
                    float* __restrict__ dzc_o, float* __restrict__ dzu_o, float* __restrict__ dh_o) {
    using G = Geo<F>;
    using U = Unit<F>;
    constexpr int CIN = 2 * F, CPT = F / G::NS, LC = lds_stride<CIN>(), Q = U::Q, NQ = U::NQ;
    __shared__ __attribute__((aligned(16))) float tile[G::PH * PW * LC];
    __shared__ float lnl[6 * F];
    __shared__ double fold[2 * F];
    const TileIdx t = tile_index<F>(H, W);
    const double n = (double)H * W * F;
    float mr, isr, mu, isu, m1r, m2r, m1u, m2u;
    ln_moments(stats, 0, n, mr, isr);
    ln_moments(stats, 1, n, mu, isu);
    ln_bwd_means<F>(part, 0, ln, n, fold, m1r, m2r);
    ln_bwd_means<F>(part, 1, ln + 2 * F, n, fold, m1u, m2u);
    stage_ln<F>(ln, lnl);
    for (int ui = threadIdx.x; ui < G::PH * PW * 2 * NQ; ui += 256) {
        const int f = ui / (2 * NQ), qd = ui - f * 2 * NQ, half = qd / NQ, ch0 = (qd - half * NQ) * Q;
        const int r = f / PW, cc = f - r * PW, gy = t.y0 - 1 + r, gx = t.x0 - 1 + cc;
        float v[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) v[i] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t q = (size_t)gy * W + gx;
            const float mean = half ? mu : mr, is = half ? isu : isr, m1 = half ? m1u : m1r, m2 = half ? m2u : m2r;
            float gv[Q];
            load_vec<Q>((half ? dzu : dzr) + q * F + ch0, v);
            load_vec<Q>(g + q * CIN + half * F + ch0, gv);
#pragma unroll
            for (int i = 0; i < Q; ++i) v[i] = is * (lnl[half * 2 * F + ch0 + i] * v[i] - m1 - (gv[i] - mean) * is * m2);
            if (r >= 1 && r <= G::TH && cc >= 1 && cc <= TW) {
#pragma unroll
                for (int i = 0; i < Q; ++i) gpx[q * 3 * F + half * F + ch0 + i] = v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) tile[f * LC + half * F + ch0 + i] = v[i];
    }
    __syncthreads();
    const int co0 = t.co_group * CPT;
    float acc[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
    conv_taps<CIN, F, CPT>(tile, t.ly, t.lx, (cfloat*)wgh_t, co0, acc);
    if (!gh_p) {                                                    // plane 0 of the call: hand the state gradient out
        if (dh_o && t.valid) {
            const size_t p = (size_t)t.py * W + t.px;
#pragma unroll
            for (int j = 0; j < CPT; ++j) dh_o[p * F + co0 + j] = dh_next[p * F + co0 + j] + acc[j];
        }
        return;
    }
    // acc + dh_next is the complete gradient of the state entering this plane = leaving the plane below: run that
    // plane's blend backward right here (same arithmetic as gt_bwd_blend_kernel), one launch less per plane
    __shared__ float red[4][4][CPT];
    float mc, isc;
    ln_moments(stats_p, 1, n, mu, isu);                             // from here on: the moments of the plane below
    ln_moments(stats_p, 2, n, mc, isc);
    float s4[4][CPT];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < CPT; ++j) s4[k][j] = 0.f;
    if (t.valid) {
        const size_t p = (size_t)t.py * W + t.px;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int f = co0 + j;
            const size_t i = p * F + f;
            const float gn = (g_p[p * 2 * F + F + f] - mu) * isu, cn = (c_p[i] - mc) * isc;
            const float u = sigm(lnl[2 * F + f] * gn + lnl[3 * F + f]), y = tanh_fast(lnl[4 * F + f] * cn + lnl[5 * F + f]);
            const float d = dh_next[i] + acc[j] + gh_p[i];
            const float vc = d * (1.0f - u) * (1.0f - y * y);
            const float vu = d * (h_pp[i] - y) * u * (1.0f - u);
            dh_o[i] = d * u;
            dzc_o[i] = vc; dzu_o[i] = vu;
            s4[0][j] = vc; s4[1][j] = vc * cn; s4[2][j] = vu; s4[3][j] = vu * gn;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < CPT; ++j) { const float v = wave_sum(s4[k][j]); if (lane == 0) red[wv][k][j] = v; }
    __syncthreads();
    if (threadIdx.x < 4 * F) {
        const int k = threadIdx.x / F, f = threadIdx.x % F, grp = f / CPT, j = f % CPT;
        constexpr int WPG = 4 / G::NS;
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < WPG; ++w) tot += (double)red[grp * WPG + w][k][j];
        const int norm = k < 2 ? 2 : 1, ab = k & 1;
        atomicAdd(&part_p[((size_t)(norm * SLOTS_B + blockIdx.x % SLOTS_B) * 2 + ab) * F + f], tot);
    }
}

template <int F>
int cell_fwd(const float* px, const float* wgh, const float* woh, const float* ln, int D, int H, int W,
             float* g, float* c, float* rh, float* h, double* stats, hipStream_t st) {
    using G = Geo<F>;
    const size_t hw = (size_t)H * W;
    const int tiles = ((H + G::TH - 1) / G::TH) * ((W + TW - 1) / TW);
    const int eb = mvs_cdiv((long long)hw * F, 256);
    for (int d = 0; d < D; ++d) {
        const float* pxd = px + d * hw * 3 * F;
        float* hp = h + d * hw * F;
        float* gd = g + d * hw * 2 * F;
        float* cd = c + d * hw * F;
        double* sd = stats + (size_t)d * SLOTS_F * 6;
        // h[d] = blend of plane d-1, formed inside the gate convolution's staging (h[0] is the caller's)
        if (d == 0)
            gt_gates_kernel<F><<<tiles, 256, 0, st>>>(pxd, nullptr, nullptr, nullptr, nullptr, ln, wgh, H, W, h, gd, sd);
        else
            gt_gates_kernel<F><<<tiles, 256, 0, st>>>(pxd, h + (d - 1) * hw * F, c + (d - 1) * hw * F, g + (d - 1) * hw * 2 * F,
                                                      sd - SLOTS_F * 6, ln, wgh, H, W, h + d * hw * F, gd, sd);
        gt_out_kernel<F><<<tiles, 256, 0, st>>>(pxd, hp, gd, sd, woh, ln, H, W, rh + d * hw * F, cd);
    }
    gt_blend_kernel<F><<<eb, 256, 0, st>>>(c + (D - 1) * hw * F, g + (D - 1) * hw * 2 * F, h + (D - 1) * hw * F,
                                           stats + (size_t)(D - 1) * SLOTS_F * 6, ln, (int)hw, h + D * hw * F);
    return (int)hipGetLastError();
}

template <int F>
int cell_bwd(const float* gh, const float* g, const float* c, const float* h, const double* stats, const float* wgh_t,
             const float* woh_t, const float* ln, int D, int H, int W, float* gpx, double* part, float* scratch,
             const float* dh_in, float* dh_out, hipStream_t st) {
    using G = Geo<F>;
    const size_t hw = (size_t)H * W, pf = hw * F;
    const int tiles = ((H + G::TH - 1) / G::TH) * ((W + TW - 1) / TW);
    int eb = mvs_cdiv((long long)pf, 256 * 8);                      // ~8 elements per thread
    if (eb < 1) eb = 1;
    float *dzc = scratch, *dzr = scratch + pf, *dzu[2] = {scratch + 2 * pf, scratch + 3 * pf},
          *dh[2] = {scratch + 4 * pf, scratch + 5 * pf};
    const size_t pstride = (size_t)3 * SLOTS_B * 2 * F;
    // the state gradient entering plane d from above sits in dh[d & 1]; a plane's blend backward leaves dh'*u in the
    // other buffer, the two convolution kernels add their parts to it.  dz_u alternates as well: the gate kernel of
    // plane d still stages plane d's while its epilogue writes plane d-1's.
    {
        const int d = D - 1;
        gt_bwd_blend_kernel<F><<<eb, 256, 0, st>>>(gh + d * pf, dh_in ? dh_in : dh[d & 1], c + d * pf, g + d * hw * 2 * F, h + d * pf,
                                                   stats + (size_t)d * SLOTS_F * 6, ln, (int)hw, dzc, dzu[d & 1],
                                                   dh[(d + 1) & 1], part + d * pstride);
    }
    for (int d = D - 1; d >= 0; --d) {
        const float* gd = g + d * hw * 2 * F;
        const float* cd = c + d * pf;
        const float* hp = h + d * pf;
        const double* sd = stats + (size_t)d * SLOTS_F * 6;
        double* pd = part + d * pstride;
        float* gpd = gpx + d * hw * 3 * F;
        float* nxt = dh[(d + 1) & 1];
        gt_bwd_out_kernel<F><<<tiles, 256, 0, st>>>(dzc, cd, gd, hp, sd, pd, woh_t, ln, H, W, gpd, nxt, dzr);
        if (d > 0)
            gt_bwd_gates_kernel<F><<<tiles, 256, 0, st>>>(dzr, dzu[d & 1], gd, sd, pd, wgh_t, ln, H, W, gpd, nxt,
                                                          gh + (d - 1) * pf, c + (d - 1) * pf, g + (d - 1) * hw * 2 * F,
                                                          h + (d - 1) * pf, sd - SLOTS_F * 6, pd - pstride,
                                                          dzc, dzu[(d - 1) & 1], dh[d & 1]);
        else
            gt_bwd_gates_kernel<F><<<tiles, 256, 0, st>>>(dzr, dzu[0], gd, sd, pd, wgh_t, ln, H, W, gpd, nxt, nullptr, nullptr,
                                                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, dh_out);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mvs_gru_train_slots(int* forward_slots, int* backward_slots) {
    if (forward_slots) *forward_slots = SLOTS_F;
    if (backward_slots) *backward_slots = SLOTS_B;
    return 0;
}

extern "C" int mvs_gru_train_cell_fwd_f32(const float* px, const float* wgh, const float* woh, const float* ln,
                                          int D, int H, int W, int F, float* g, float* c, float* rh, float* h,
                                          double* stats, void* stream) {
    MVS_CHECK_ARG(px && wgh && woh && ln && g && c && rh && h && stats && D > 0 && H > 0 && W > 0);
    if ((long long)H * W * 3 * F * D >= (1LL << 40)) return MVS_E_SHAPE;
    hipStream_t st = mvs_stream(stream);
    switch (F) {
        case 16: return cell_fwd<16>(px, wgh, woh, ln, D, H, W, g, c, rh, h, stats, st);
        case 8: return cell_fwd<8>(px, wgh, woh, ln, D, H, W, g, c, rh, h, stats, st);
        case 4: return cell_fwd<4>(px, wgh, woh, ln, D, H, W, g, c, rh, h, stats, st);
        case 2: return cell_fwd<2>(px, wgh, woh, ln, D, H, W, g, c, rh, h, stats, st);
        case 1: return cell_fwd<1>(px, wgh, woh, ln, D, H, W, g, c, rh, h, stats, st);
        default: return MVS_E_SHAPE;
    }
}

extern "C" int mvs_gru_train_cell_bwd_f32(const float* gh, const float* g, const float* c, const float* h,
                                          const double* stats, const float* wgh_t, const float* woh_t, const float* ln,
                                          int D, int H, int W, int F, float* gpx, double* part, float* scratch,
                                          const float* dh_in, float* dh_out, void* stream) {
    MVS_CHECK_ARG(gh && g && c && h && stats && wgh_t && woh_t && ln && gpx && part && scratch && D > 0 && H > 0 && W > 0);
    hipStream_t st = mvs_stream(stream);
    switch (F) {
        case 16: return cell_bwd<16>(gh, g, c, h, stats, wgh_t, woh_t, ln, D, H, W, gpx, part, scratch, dh_in, dh_out, st);
        case 8: return cell_bwd<8>(gh, g, c, h, stats, wgh_t, woh_t, ln, D, H, W, gpx, part, scratch, dh_in, dh_out, st);
        case 4: return cell_bwd<4>(gh, g, c, h, stats, wgh_t, woh_t, ln, D, H, W, gpx, part, scratch, dh_in, dh_out, st);
        case 2: return cell_bwd<2>(gh, g, c, h, stats, wgh_t, woh_t, ln, D, H, W, gpx, part, scratch, dh_in, dh_out, st);
        case 1: return cell_bwd<1>(gh, g, c, h, stats, wgh_t, woh_t, ln, D, H, W, gpx, part, scratch, dh_in, dh_out, st);
        default: return MVS_E_SHAPE;
    }
}
