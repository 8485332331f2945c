// ConvGRU regulariser (R-MVSNet) and winner-take-all depth sweep (R8, R9, K10, K11).
// Reference behaviour: ConvGRUCell.__call__ (mvsnet/convgru.py:82-122) with group_norm reducing
// to tf.contrib.layers.layer_norm for every filter count on the path (convgru.py:24-31), and the
// while_loop body / tail of inference_winner_take_all (mvsnet/model.py:676-751).
//
// v1 kernels: shape-generic VALU convolution over the channel concatenation [xa | xb] (no concat
// is materialised), LayerNorm moments accumulated by the producing convolution, gate / blend
// element-wise stages, WTA update.  The sweep keeps all state resident in HBM/L2; only the
// current cost slice (H*W*C) exists, never the (D,H,W,C) volume.
#include "common.h"
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <mutex>

// cell-1 MFMA convolutions (gru_mfma.hip); MVS_E_SHAPE outside their tiling
int mvs_gru1_split_weights(const float* w_gates, const float* w_out, int CA, int F, float* wx, float* wgh, float* woh,
                           hipStream_t st);
int mvs_gru1_xpart_mfma(const float* x, const float* wxg, const float* wxo, const float* bias_g, const float* bias_o,
                        int H, int W, int planes, float* px, hipStream_t st);
int mvs_gru1_gates_h_mfma(const float* h, const float* wgh, const float* px, int H, int W, float* g, double* stats,
                          int views, size_t vstride, hipStream_t st);
int mvs_gru1_gates_h_blend_mfma(const float* h_before, const float* c_prev, const float* g_prev, const double* stats_c,
                                const double* stats_u, const float* o_gamma, const float* o_beta, const float* u_gamma,
                                const float* u_beta, float* h_out, const float* wgh, const float* px, int H, int W,
                                float* g, double* stats, int views, size_t vstride, hipStream_t st);
int mvs_gru1_out_h_mfma(const float* h, const float* g, const double* g_stats, const float* r_gamma, const float* r_beta,
                        const float* woh, const float* px, int H, int W, float* c, double* stats, int views, size_t vstride,
                        hipStream_t st);
int mvs_gru1_full_weights(const float* w_gates, const float* w_out, int CA, int F, float* wg, float* wo, hipStream_t st);
int mvs_gru1_gates_full_mfma(const float* x, const float* h, const float* wg, const float* bias, int H, int W, float* g,
                             double* stats, int views, size_t vstride, hipStream_t st);
int mvs_gru1_gates_full_blend_mfma(const float* x, const float* h_before, const float* c_prev, const float* g_prev,
                                   const double* stats_c, const double* stats_u, const float* o_gamma, const float* o_beta,
                                   const float* u_gamma, const float* u_beta, float* h_out, const float* wg,
                                   const float* bias, int H, int W, float* g, double* stats, int views, size_t vstride,
                                   hipStream_t st);
int mvs_gru1_out_full_mfma(const float* x, const float* h, const float* g, const double* g_stats, const float* r_gamma,
                           const float* r_beta, const float* wo, const float* bias, int H, int W, float* c, double* stats,
                           int views, size_t vstride, hipStream_t st);
int mvs_cost_volume_threads_f32(const float* ref, const float* src, const float* transforms, int view_num, int depth_total, int d_begin,
                                int d_count, int H, int W, int C, int variant, int negate, int border, float* cost, int threads, void* stream);
// the fused two-launches-per-plane sweep (gru_fused.hip)
struct GruFusedWs {
    char* base;
    float* x; float* S[3][2]; float* G[3][2]; float* Cb[3]; double* stats;
    float *max_prob, *depth, *exp_sum;
    float *w1g, *w1c, *wsg, *wsc;
};
constexpr int GRU_FUSED_RING = 64;       // LayerNorm-sum rows of the fused sweep: plane p uses row p % 64
constexpr int GRU_FUSED_SLOTS = 8;       // copies of a row the workgroups spread their float64 atomics over (gru_fused.hip)
constexpr int GRU_FUSED_SLOT_STRIDE = 32;                  // doubles between copies: every copy on cache lines of its own (18 used)
constexpr int GRU_FUSED_ROW = GRU_FUSED_SLOT_STRIDE * GRU_FUSED_SLOTS;      // doubles per row: [slot][cell][6]
int mvs_gru_fused_prepare_weights(const float* const* params, const GruFusedWs& ws, hipStream_t st);
int mvs_gru_fused_step(const GruFusedWs& ws, const float* const* params, int t, int depth_num, const float* x_t, int H, int W,
                       int views, size_t vstride, const float* depth_values, hipStream_t st);
namespace {

template <int CO>
__global__ void __launch_bounds__(256)
conv2d_cat_kernel(const float* __restrict__ xa, int Ca, const float* __restrict__ xb, int Cb,
                  const float* __restrict__ w, const float* __restrict__ bias, int H, int W,
                  float* __restrict__ y, double* __restrict__ stats, int groups) {
    __shared__ float red[4][2][2];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    const bool valid = pix < HW;
    const int Ct = Ca + Cb;
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = bias ? bias[j] : 0.f;
    if (valid) {
        const int py = pix / W, px = pix - py * W;
        for (int kh = 0; kh < 3; ++kh) {
            int iy = py + kh - 1; if (iy < 0 || iy >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                int ix = px + kw - 1; if (ix < 0 || ix >= W) continue;
                const size_t p = (size_t)iy * W + ix;
                const float* wt = w + (size_t)(kh * 3 + kw) * Ct * CO;
                for (int ci = 0; ci < Ca; ++ci) {
                    float xv = xa[p * Ca + ci];
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j] += xv * wt[(size_t)ci * CO + j];
                }
                for (int ci = 0; ci < Cb; ++ci) {
                    float xv = xb[p * Cb + ci];
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j] += xv * wt[(size_t)(Ca + ci) * CO + j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CO; ++j) y[(size_t)pix * CO + j] = acc[j];
    }
    if (stats) {
        const int per = CO / groups;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int g = 0; g < groups; ++g) {
            float s = 0.f, q = 0.f;
            if (valid) {
#pragma unroll
                for (int j = 0; j < CO; ++j)
                    if (j / per == g) { s += acc[j]; q += acc[j] * acc[j]; }
            }
            s = wave_sum(s); q = wave_sum(q);
            if (lane == 0) { red[wv][g][0] = s; red[wv][g][1] = q; }
        }
        __syncthreads();
        if (threadIdx.x < groups * 2) {
            int g = threadIdx.x >> 1, k = threadIdx.x & 1;
            double t = (double)red[0][g][k] + (double)red[1][g][k] + (double)red[2][g][k] + (double)red[3][g][k];
            atomicAdd(&stats[g * 2 + k], t);
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct LN { float inv_r, mean_r; };

// LayerNorm affine from accumulated moments: y = x*inv[c] + (beta[c] - mean*inv[c]),
// inv[c] = gamma[c] / sqrt(var + 1e-12)   (tf.contrib.layers.layer_norm)
__device__ __forceinline__ void ln_affine(const double* st, double n, float gamma, float beta,
                                          float& a, float& b) {
    double mean = st[0] / n;
    double var = st[1] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    double inv = (double)gamma / sqrt(var + 1e-12);
    a = (float)inv;
    b = (float)((double)beta - mean * inv);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256)
gru_gates_kernel(const float* __restrict__ g, const double* __restrict__ stats,
                 const float* __restrict__ rg, const float* __restrict__ rb,
                 const float* __restrict__ ug, const float* __restrict__ ub,
                 const float* __restrict__ h, int HW, int F, float* __restrict__ rh,
                 float* __restrict__ u) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)HW * F) return;
    int f = (int)(i % F);
    long long pix = i / F;
    double n = (double)HW * F;
    float a, b;
    ln_affine(stats, n, rg[f], rb[f], a, b);
    float r = sigmoidf(g[pix * 2 * F + f] * a + b);               // convgru.py:97,101
    ln_affine(stats + 2, n, ug[f], ub[f], a, b);
    float uu = sigmoidf(g[pix * 2 * F + F + f] * a + b);          // convgru.py:98,102
    rh[i] = r * h[i];                                             // convgru.py:107
    u[i] = uu;
}

__global__ void __launch_bounds__(256)
gru_blend_kernel(const float* __restrict__ c, const double* __restrict__ stats,
                 const float* __restrict__ og, const float* __restrict__ ob,
                 const float* __restrict__ u, int HW, int F, float* __restrict__ h) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)HW * F) return;
    int f = (int)(i % F);
    float a, b;
    ln_affine(stats, (double)HW * F, og[f], ob[f], a, b);
    float yv = tanhf(c[i] * a + b);                               // convgru.py:114,117
    float uu = u[i];
    h[i] = uu * h[i] + (1.0f - uu) * yv;                          // convgru.py:120
}

__global__ void __launch_bounds__(256)
wta_update_kernel(const float* __restrict__ reg, float depth_value, int HW,
                  float* __restrict__ max_prob, float* __restrict__ depth_image,
                  float* __restrict__ exp_sum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    float p = expf(reg[i]);                                       // model.py:703
    float mp = max_prob[i];
    if (mp < p) { max_prob[i] = p; depth_image[i] = depth_value; }  // :721-728 (strict <)
    exp_sum[i] += p;                                              // :731
}

__global__ void __launch_bounds__(256)
wta_finish_kernel(const float* __restrict__ max_prob, const float* __restrict__ exp_sum, int HW,
                  float* __restrict__ prob) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < HW) prob[i] = max_prob[i] / (exp_sum[i] + 1e-7f);     // model.py:749-751
}

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

// Small-channel 3x3 convolution over [xa | xb] for ConvGRU cells 2 and 3 (20 -> 8/4, 6 -> 4/2
// channels).  A workgroup owns a 16 x 16 pixel tile: the 18 x 18 neighbourhood of the concatenation is
// staged once in LDS (MODE 1 folds the reset gate into xb there: xb = sigmoid(LayerNorm(g_r)) * h,
// convgru.py:97,101,107 -- evaluated once per staged element, not once per tap), every thread then
// reads its 9 taps from LDS; weights come through the scalar cache.
// MODE 2: what the previous plane's blend needs, evaluated while staging xb (see conv2d_small_kernel)
constexpr int MAXV = 8;          // reference views per sweep launch (mvs_gru_wta_batch_f32)
// View v of a multi-view launch: every tensor of the sweep lives `vstride` bytes after view v-1's (one workspace block per
// view); weights, biases and LayerNorm parameters are shared.
template <class T> __device__ __forceinline__ T* view_ptr(T* p, size_t vo) { return p ? (T*)((char*)p + vo) : p; }
template <class T> __device__ __forceinline__ const T* view_ptr(const T* p, size_t vo) { return p ? (const T*)((const char*)p + vo) : p; }

struct BlendIn {
    const float* c; const float* g;                   // previous plane: raw candidate (H,W,F) and gate (H,W,2F) convolutions
    const double* stats_c; const double* stats_u;     // their LayerNorm moments [sum, sumsq]
    const float *og, *ob, *ug, *ub;                   // candidate / update LayerNorm gamma, beta
    float* h_out;                                     // receives the blended state (the tile's own pixels)
    // cell 3 only (pw != null): prob_conv + exp + winner-take-all update of the PREVIOUS plane, whose final state is the
    // state just formed in the tile (model.py:701-703, 721-731): one launch less per plane
    const float* pw; const float* pbias; float depth_value[MAXV];      // per view: the planes' depths differ between views
    float *max_prob, *depth_image, *exp_sum;
};

// MODE 2 folds the PREVIOUS plane's blend into the staging of xb: xb holds the state that entered the previous plane
// and the state entering this one, u*h + (1-u)*tanh(LN c) (convgru.py:98,102,114-120), is formed on load (halo
// positions are recomputed by the neighbouring tiles), written out for the tile's own pixels -- the candidate
// convolution, the next cell and the WTA update read it -- and convolved: one launch less per plane and cell.
struct SmallArgs {
    const float* xa; const float* xb; const float* g; const double* g_stats; const float* r_gamma; const float* r_beta;
    const float* w; const float* bias; int H, W; float* y; double* stats; int groups; BlendIn bl;
    size_t vstride;                                   // blockIdx.y = view
};

template <int CA, int CB, int CO, int MODE, bool MATRIX>
__device__ __forceinline__ void conv2d_small_body(const SmallArgs& sa, const int bid) {
    const int view = blockIdx.y;
    const size_t vo = (size_t)view * sa.vstride;
    const float* __restrict__ xa = view_ptr(sa.xa, vo); const float* __restrict__ xb = view_ptr(sa.xb, vo);
    const float* __restrict__ g = view_ptr(sa.g, vo); const double* __restrict__ g_stats = view_ptr(sa.g_stats, vo);
    const float* __restrict__ r_gamma = sa.r_gamma; const float* __restrict__ r_beta = sa.r_beta;
    const float* __restrict__ w = sa.w; const float* __restrict__ bias = sa.bias;
    const int H = sa.H, W = sa.W, groups = sa.groups;
    float* __restrict__ y = view_ptr(sa.y, vo); double* __restrict__ stats = view_ptr(sa.stats, vo);
    // (no local copy of sa.bl: its per-view depth array would be indexed dynamically in registers, i.e. put into scratch memory)
    const BlendIn& bl = sa.bl;
    const float* __restrict__ bl_c = view_ptr(bl.c, vo); const float* __restrict__ bl_g = view_ptr(bl.g, vo);
    const double* __restrict__ bl_stats_c = view_ptr(bl.stats_c, vo); const double* __restrict__ bl_stats_u = view_ptr(bl.stats_u, vo);
    float* __restrict__ bl_h_out = view_ptr(bl.h_out, vo); float* __restrict__ bl_max_prob = view_ptr(bl.max_prob, vo);
    float* __restrict__ bl_depth_image = view_ptr(bl.depth_image, vo); float* __restrict__ bl_exp_sum = view_ptr(bl.exp_sum, vo);
    constexpr int CT = CA + CB;
    constexpr int TS = 16, PS = TS + 2;
    typedef const __attribute__((address_space(4))) float cfloat;      // wave-uniform -> s_load into SGPRs
    cfloat* wsh = (cfloat*)w;
    __shared__ __attribute__((aligned(16))) float tile[PS * PS * CT];
    __shared__ float red[4][2][2];
    // Round 4: the convolution itself on v_mfma_f32_4x4x1_16B_f32 -- 16 blocks of (4 output channels x 4 pixels), K = 1: a lane
    // stays one pixel, its B operand is the staged input value it already holds, its four result registers are 4 output
    // channels of that pixel, and the A operand (lane m: weight of output channel m & 3) is one LDS read per (tap, channel).
    // Exactly the fused multiply-add chain of the vector form, in the same order: the same bits (tools/small_cell_probe.hip
    // checks it).  The probe (profiles/r04_small_cell_probe.txt) measured the 20 -> 8 convolution at 82.7 TFLOP/s on this
    // form against 48.0 on v_pk_fma_f32 with SGPR weights -- the packed FMA issues at half rate on gfx950, so the matrix
    // instruction is twice the vector ALU's real fp32 rate, without padding for 4- and 8-channel outputs (2 channels: half).
    // MATRIX is chosen by the launcher: the matrix form for ONE reference view per sweep (same-box A/B, c3: 22.49 -> 22.06 ms),
    // the vector form for several views per launch (72.4 against 73.1 ms per 4-view sweep: there the kernels are bound by memory
    // and the sweep by the matrix pipe cell 1 keeps busy; the 5.8 KB weight table costs the 20-channel instances one of their
    // six workgroups per CU).  Both forms give the same bits.
    constexpr bool MFMA44 = MATRIX && (CO == 8 || CO == 4 || CO == 2);
    constexpr int NGRP = CO > 4 ? 2 : 1;               // matrix instructions per (tap, input channel)
    __shared__ __attribute__((aligned(16))) float wl[MFMA44 ? 9 * CT * 4 * NGRP : 4];      // [tap * CT + ci][m & 3][group]
    if (MFMA44) {
        for (int i = threadIdx.x; i < 9 * CT * 4 * NGRP; i += 256) {
            const int g_ = i % NGRP, r_ = (i / NGRP) & 3, k_ = i / (4 * NGRP), co = r_ + 4 * g_;
            wl[i] = co < CO ? w[k_ * CO + co] : 0.f;     // visible to the convolution after the staging barrier below
        }
    }
    // LayerNorm moments -> (mean, 1 / sqrt(var + eps)) ONCE per workgroup (two lanes, then LDS): every thread used to run the
    // float64 divisions and square roots itself, per channel -- ~1000 instruction slots per wave at the head of a kernel whose
    // own work is a few hundred (round 3: the four small-cell kernels 43 / 30 / 27 / 15 us per 4-view plane before)
    // Round 4: the per-channel (scale, shift) pairs too -- CB (MODE 2: 2 * CB) lanes do the float64 arithmetic, everybody reads
    // the float results back as LDS broadcasts (every thread used to redo 4 * CB float64 multiplies and conversions)
    __shared__ float lna[2][CB][2];
    if (MODE != 0) {
        if (threadIdx.x < (MODE == 2 ? 2 * CB : CB)) {
            const int k = threadIdx.x / CB, f = threadIdx.x - k * CB;       // k = 0: reset (MODE 1) / update (MODE 2) gate, 1: candidate
            const double cnt = (double)H * W * CB;
            const double* st = MODE == 1 ? g_stats : (k == 0 ? bl_stats_u : bl_stats_c);
            const double mean = st[0] / cnt;
            double var = st[1] / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            const double rstd = 1.0 / sqrt(var + 1e-12);              // tf.contrib.layers.layer_norm, eps 1e-12
            const float gamma = MODE == 1 ? r_gamma[f] : (k == 0 ? bl.ug[f] : bl.og[f]);
            const float beta = MODE == 1 ? r_beta[f] : (k == 0 ? bl.ub[f] : bl.ob[f]);
            const double inv = (double)gamma * rstd;
            lna[k][f][0] = (float)inv; lna[k][f][1] = (float)((double)beta - mean * inv);
        }
        __syncthreads();
    }
    float ra[CB], rb[CB];
    if (MODE == 1) {
#pragma unroll
        for (int f = 0; f < CB; ++f) { ra[f] = lna[0][f][0]; rb[f] = lna[0][f][1]; }
    }
    float ua[CB], ub_[CB], ca[CB], cb_[CB];
    if (MODE == 2) {
#pragma unroll
        for (int f = 0; f < CB; ++f) { ua[f] = lna[0][f][0]; ub_[f] = lna[0][f][1]; ca[f] = lna[1][f][0]; cb_[f] = lna[1][f][1]; }
    }
    const int tiles_x = (W + TS - 1) / TS;
    const int ty = bid / tiles_x, tx = bid - ty * tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    for (int f = threadIdx.x; f < PS * PS; f += 256) {
        const int r = f / PS, c = f - r * PS;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        float va[CA], vb[CB];
#pragma unroll
        for (int i = 0; i < CA; ++i) va[i] = 0.f;
#pragma unroll
        for (int i = 0; i < CB; ++i) vb[i] = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t p = (size_t)gy * W + gx;
            load_vec<CA>(xa + p * CA, va);
            load_vec<CB>(xb + p * CB, vb);
            if (MODE == 1) {
                float gr[CB];
                load_vec<CB>(g + p * 2 * CB, gr);
#pragma unroll
                for (int i = 0; i < CB; ++i) vb[i] *= mvs_sigmoid_fast(gr[i] * ra[i] + rb[i]);
            }
            if (MODE == 2) {
                float cv[CB], gu[CB];
                load_vec<CB>(bl_c + p * CB, cv);
                load_vec<CB>(bl_g + p * 2 * CB + CB, gu);
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    const float uu = mvs_sigmoid_fast(gu[i] * ua[i] + ub_[i]);
                    vb[i] = uu * vb[i] + (1.0f - uu) * mvs_tanh_fast(cv[i] * ca[i] + cb_[i]);
                }
                if (r >= 1 && r <= TS && c >= 1 && c <= TS) {
#pragma unroll
                    for (int i = 0; i < CB; ++i) bl_h_out[p * CB + i] = vb[i];
                }
            }
        }
        float* d = tile + f * CT;
#pragma unroll
        for (int i = 0; i < CA; ++i) d[i] = va[i];
#pragma unroll
        for (int i = 0; i < CB; ++i) d[CA + i] = vb[i];
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int py = y0 + ly, px = x0 + lx;
    const bool valid = py < H && px < W;
    if (MODE == 2 && bl.pw != nullptr && valid) {     // as prob_wta_kernel: taps outside the image are staged zeros
        float pacc = bl.pbias ? bl.pbias[0] : 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int gy = py + kh - 1, gx = px + kw - 1;
                if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
                const float* tp = tile + ((ly + kh) * PS + lx + kw) * CT + CA;
#pragma unroll
                for (int ci = 0; ci < CB; ++ci) pacc += tp[ci] * bl.pw[(kh * 3 + kw) * CB + ci];
            }
        const float pr = expf(pacc);
        const int pix = py * W + px;
        const float mp = bl_max_prob[pix];
        if (mp < pr) { bl_max_prob[pix] = pr; bl_depth_image[pix] = bl.depth_value[view]; }     // kernarg array, uniform index: s_load
        bl_exp_sum[pix] += pr;
    }
    float acc[CO];
#pragma unroll
    for (int j = 0; j < CO; ++j) acc[j] = bias ? bias[j] : 0.f;
    if constexpr (MFMA44) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (j < CO) a0[j] = acc[j]; if (NGRP == 2) a1[j] = acc[CO > 4 ? 4 + j : 0]; }
        const float* wa = wl + (threadIdx.x & 3) * NGRP;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v[CT];
                load_vec<CT>(tile + ((ly + kh) * PS + lx + kw) * CT, v);
                const float* wt = wa + (kh * 3 + kw) * CT * 4 * NGRP;
#pragma unroll
                for (int ci = 0; ci < CT; ++ci) {
                    if constexpr (NGRP == 2) {
                        const float2 wv = *(const float2*)(wt + ci * 8);
                        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, v[ci], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, v[ci], a1, 0, 0, 0);
                    } else {
                        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wt[ci * 4], v[ci], a0, 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (j < CO) acc[j] = a0[j]; if (NGRP == 2) acc[CO > 4 ? 4 + j : 0] = a1[j]; }
    } else {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v[CT];
                load_vec<CT>(tile + ((ly + kh) * PS + lx + kw) * CT, v);
                cfloat* wt = wsh + (kh * 3 + kw) * CT * CO;
#pragma unroll
                for (int ci = 0; ci < CT; ++ci)
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j] = __builtin_fmaf(v[ci], wt[ci * CO + j], acc[j]);
                // (explicitly fused: written as acc += v * w the 20 -> 4 instance came out with part of its products on
                // v_pk_mul_f32 + v_add_f32, i.e. rounded twice -- legal under -ffp-contract=fast, but not the matrix form's
                // multiply-add chain, and the two forms must give the same bits)
            }
        }
    }
    if (valid) {
        float* dst = y + ((size_t)py * W + px) * CO;
#pragma unroll
        for (int j = 0; j < CO; ++j) dst[j] = acc[j];
    }
    if (stats) {
        const int per = CO / groups;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        for (int gi = 0; gi < groups; ++gi) {
            float sv = 0.f, qv = 0.f;
            if (valid) {
#pragma unroll
                for (int j = 0; j < CO; ++j)      // explicit fused multiply-add: left to -ffp-contract the matrix and the vector
                    if (j / per == gi) { sv += acc[j]; qv = __builtin_fmaf(acc[j], acc[j], qv); }      // form of this kernel came out differently (1 ulp in the moments)
            }
            sv = wave_sum(sv); qv = wave_sum(qv);
            if (lane == 0) { red[wv][gi][0] = sv; red[wv][gi][1] = qv; }
        }
        __syncthreads();
        if (threadIdx.x < groups * 2) {
            int gi = threadIdx.x >> 1, k = threadIdx.x & 1;
            double t = (double)red[0][gi][k] + (double)red[1][gi][k] + (double)red[2][gi][k] + (double)red[3][gi][k];
            atomicAdd(&stats[gi * 2 + k], t);
        }
    }
}

template <int CA, int CB, int CO, int MODE, bool MATRIX>
__global__ void __launch_bounds__(256)
conv2d_small_kernel(SmallArgs a) { conv2d_small_body<CA, CB, CO, MODE, MATRIX>(a, blockIdx.x); }

// cells 2 / 3: gate conv then candidate conv (reset gate folded in); false if the shape has no instance
// `prev`: the previous plane's blend has not been launched -- its inputs; h then RECEIVES the state entering this plane
// (formed from h_before, the state that entered the previous plane) in the gate convolution's staging
struct PrevPlane { const float* h_before; const float* g; const double* sg; const double* so; };
// `wta` (cell 3, with `prev`): prob_conv + winner-take-all update of the previous plane inside the gate convolution
struct WtaFold { const float* pw; const float* pbias; float depth_value[MAXV]; float *max_prob, *depth_image, *exp_sum; };
struct Views { int n; size_t stride; };              // views per launch, byte stride between their tensors
template <int CA, int F>
bool launch_small_cell(const float* xin, float* h, const float* const* p, int H, int W, float* g, float* c,
                       double* sg, double* so, const PrevPlane* prev, Views vw, hipStream_t st, const WtaFold* wta = nullptr) {
    const dim3 grid(((H + 15) / 16) * ((W + 15) / 16), vw.n);     // 16 x 16 pixel tiles x views
    BlendIn none = {};
    const bool mg = vw.n == 1, mc = vw.n == 1;                      // matrix form for one view per sweep, vector form for several (conv2d_small_body)
    if (prev) {
        BlendIn bl = {c, prev->g, prev->so, prev->sg + 2, p[8], p[9], p[4], p[5], h, nullptr, nullptr, {}, nullptr, nullptr, nullptr};
        if (wta) {
            bl.pw = wta->pw; bl.pbias = wta->pbias; bl.max_prob = wta->max_prob; bl.depth_image = wta->depth_image; bl.exp_sum = wta->exp_sum;
            for (int v = 0; v < MAXV; ++v) bl.depth_value[v] = wta->depth_value[v];
        }
        const SmallArgs ga{xin, prev->h_before, nullptr, nullptr, nullptr, nullptr, p[0], p[1], H, W, g, sg, 2, bl, vw.stride};
        if (mg) conv2d_small_kernel<CA, F, 2 * F, 2, true><<<grid, 256, 0, st>>>(ga);
        else conv2d_small_kernel<CA, F, 2 * F, 2, false><<<grid, 256, 0, st>>>(ga);
    } else {
        const SmallArgs ga{xin, h, nullptr, nullptr, nullptr, nullptr, p[0], p[1], H, W, g, sg, 2, none, vw.stride};
        if (mg) conv2d_small_kernel<CA, F, 2 * F, 0, true><<<grid, 256, 0, st>>>(ga);
        else conv2d_small_kernel<CA, F, 2 * F, 0, false><<<grid, 256, 0, st>>>(ga);
    }
    const SmallArgs ca{xin, h, g, sg, p[2], p[3], p[6], p[7], H, W, c, so, 1, none, vw.stride};
    if (mc) conv2d_small_kernel<CA, F, F, 1, true><<<grid, 256, 0, st>>>(ca);
    else conv2d_small_kernel<CA, F, F, 1, false><<<grid, 256, 0, st>>>(ca);
    return true;
}

// blend with the update gate evaluated in place: h = u*h + (1-u)*tanh(LN(c)), u = sigmoid(LN(g_u))
// (convgru.py:98,102,114-120); g holds the raw gate convolution (reset | update).  One thread handles
// VEC consecutive channels of a pixel (VEC | F), so the float64 LayerNorm statistics are folded into
// (scale, shift) once per thread instead of once per element.
template <int VEC>
__global__ void __launch_bounds__(256)
gru_blend_fused_kernel(const float* __restrict__ c, const double* __restrict__ stats_c,
                       const float* __restrict__ og, const float* __restrict__ ob,
                       const float* __restrict__ g, const double* __restrict__ stats_u,
                       const float* __restrict__ ug, const float* __restrict__ ub, int HW, int F,
                       const float* h, float* h_out,         // may alias (non-pipelined sweep)
                       size_t vstride) {                     // blockIdx.y = view
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    const size_t vo = (size_t)blockIdx.y * vstride;
    c = view_ptr(c, vo); stats_c = view_ptr(stats_c, vo); g = view_ptr(g, vo); stats_u = view_ptr(stats_u, vo);
    h = view_ptr(h, vo); h_out = view_ptr(h_out, vo);
    // the (scale, shift) pairs of both LayerNorms once per workgroup (2F lanes in float64, then LDS broadcasts): every thread
    // used to run two float64 square roots and 4 * VEC float64 multiplies (F <= 64: the launcher checks)
    __shared__ float aff[2][64][2];
    if (threadIdx.x < 2 * F) {
        const int k = threadIdx.x / F, f = threadIdx.x - k * F;             // k = 0: update gate, 1: candidate
        const double n = (double)HW * F;
        const double* st = k == 0 ? stats_u : stats_c;
        const double mean = st[0] / n;
        double var = st[1] / n - mean * mean; if (var < 0.0) var = 0.0;
        const double inv = (double)(k == 0 ? ug[f] : og[f]) * (1.0 / sqrt(var + 1e-12));
        aff[k][f][0] = (float)inv; aff[k][f][1] = (float)((double)(k == 0 ? ub[f] : ob[f]) - mean * inv);
    }
    __syncthreads();
    if (i >= (long long)HW * F) return;
    const int f = (int)(i % F);
    const long long pix = i / F;
    float cv[VEC], gv[VEC], hv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { cv[k] = c[i + k]; gv[k] = g[pix * 2 * F + F + f + k]; hv[k] = h[i + k]; }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const float uu = mvs_sigmoid_fast(gv[k] * aff[0][f + k][0] + aff[0][f + k][1]);
        const float yv = mvs_tanh_fast(cv[k] * aff[1][f + k][0] + aff[1][f + k][1]);
        h_out[i + k] = uu * hv[k] + (1.0f - uu) * yv;
    }
}

// prob_conv (3x3, F3 -> 1, bias) + exp + winner-take-all update in one pass
// (model.py:701-703, 721-731); strict '<' keeps the first maximum.
struct DepthVals { float v[MAXV]; };
template <int F3>
__global__ void __launch_bounds__(256)
prob_wta_kernel(const float* __restrict__ h3, const float* __restrict__ w, const float* __restrict__ bias,
                DepthVals dv, int H, int W, float* __restrict__ max_prob,
                float* __restrict__ depth_image, float* __restrict__ exp_sum, size_t vstride) {
    int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= H * W) return;
    const float depth_value = dv.v[blockIdx.y];
    const size_t vo = (size_t)blockIdx.y * vstride;
    h3 = view_ptr(h3, vo); max_prob = view_ptr(max_prob, vo); depth_image = view_ptr(depth_image, vo); exp_sum = view_ptr(exp_sum, vo);
    int py = pix / W, px = pix - py * W;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        int iy = py + kh - 1; if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            int ix = px + kw - 1; if (ix < 0 || ix >= W) continue;
            const float* p = h3 + ((size_t)iy * W + ix) * F3;
#pragma unroll
            for (int ci = 0; ci < F3; ++ci) acc += p[ci] * w[(kh * 3 + kw) * F3 + ci];
        }
    }
    float pr = expf(acc);
    float mp = max_prob[pix];
    if (mp < pr) { max_prob[pix] = pr; depth_image[pix] = depth_value; }
    exp_sum[pix] += pr;
}

int launch_conv2d(const float* xa, int Ca, const float* xb, int Cb, const float* w,
                  const float* bias, int H, int W, int Cout, float* y, double* stats, int groups,
                  hipStream_t st) {
    int grid = mvs_cdiv((long long)H * W, 256);
#define MVS_C2D(CO) case CO: conv2d_cat_kernel<CO><<<grid, 256, 0, st>>>(xa, Ca, xb, Cb, w, bias, H, W, y, stats, groups); break;
    switch (Cout) {
        MVS_C2D(1) MVS_C2D(2) MVS_C2D(4) MVS_C2D(8) MVS_C2D(16) MVS_C2D(32)
        default: return MVS_E_SHAPE;
    }
#undef MVS_C2D
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mvs_conv2d_cat_f32(const float* xa, int Ca, const float* xb, int Cb, const float* w,
                                  const float* bias, int H, int W, int Cout, float* y,
                                  double* stats, int groups, void* stream) {
    MVS_CHECK_ARG(xa && w && y && Ca > 0 && Cb >= 0 && H > 0 && W > 0 && Cout > 0);
    MVS_CHECK_ARG(Cb == 0 || xb);
    if (stats) { MVS_CHECK_ARG(groups == 1 || groups == 2); if (Cout % groups) return MVS_E_SHAPE; }
    else groups = 1;
    return launch_conv2d(xa, Ca, xb, Cb, w, bias, H, W, Cout, y, stats, groups, mvs_stream(stream));
}

extern "C" int mvs_gru_gates_f32(const float* g, const double* stats, const float* reset_gamma,
                                 const float* reset_beta, const float* update_gamma,
                                 const float* update_beta, const float* h, int H, int W, int F,
                                 float* rh, float* u, void* stream) {
    MVS_CHECK_ARG(g && stats && reset_gamma && reset_beta && update_gamma && update_beta && h && rh && u);
    MVS_CHECK_ARG(H > 0 && W > 0 && F > 0);
    long long n = (long long)H * W * F;
    gru_gates_kernel<<<mvs_cdiv(n, 256), 256, 0, mvs_stream(stream)>>>(
        g, stats, reset_gamma, reset_beta, update_gamma, update_beta, h, H * W, F, rh, u);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_gru_blend_f32(const float* c, const double* stats, const float* out_gamma,
                                 const float* out_beta, const float* u, int H, int W, int F,
                                 float* h, void* stream) {
    MVS_CHECK_ARG(c && stats && out_gamma && out_beta && u && h && H > 0 && W > 0 && F > 0);
    long long n = (long long)H * W * F;
    gru_blend_kernel<<<mvs_cdiv(n, 256), 256, 0, mvs_stream(stream)>>>(c, stats, out_gamma, out_beta,
                                                                     u, H * W, F, h);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_wta_update_f32(const float* reg, float depth_value, int H, int W,
                                  float* max_prob, float* depth_image, float* exp_sum,
                                  void* stream) {
    MVS_CHECK_ARG(reg && max_prob && depth_image && exp_sum && H > 0 && W > 0);
    wta_update_kernel<<<mvs_cdiv((long long)H * W, 256), 256, 0, mvs_stream(stream)>>>(
        reg, depth_value, H * W, max_prob, depth_image, exp_sum);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_wta_finish_f32(const float* max_prob, const float* exp_sum, int H, int W,
                                  float* prob_out, void* stream) {
    MVS_CHECK_ARG(max_prob && exp_sum && prob_out && H > 0 && W > 0);
    wta_finish_kernel<<<mvs_cdiv((long long)H * W, 256), 256, 0, mvs_stream(stream)>>>(
        max_prob, exp_sum, H * W, prob_out);
    MVS_LAUNCH_RET();
}

// ---- composed recurrent sweep -----------------------------------------------------------------

namespace {
struct GruWs {
    float *x, *g[3], *g2[3], *c[3], *rh, *u, *h[3][8 * 4], *reg, *max_prob, *exp_sum, *depth;   // h: ring of RG*PG states (RG <= 8); g2: the gate buffer of odd planes (cells whose blend is folded into the next plane's gate convolution)
    float *px, *wx, *wgh, *woh;        // hoisted x-part of cell 1: (2, XB, H, W, 3*f1) and its prepared weights
    float *wfg, *wfo;                  // cell 1 unhoisted: prepared weights of the full 48-channel convolutions
    float *wsg, *wsc;                  // fused sweep: small-cell tables of the gates / output launch (gru_fused.hip)
    double* fstats;                    // fused sweep: GRU_FUSED_RING planes x GRU_FUSED_SLOTS copies x 3 cells x 6 LayerNorm sums
    double* stats;     // per plane of a batch: 3 cells x (gates: 2 groups x 2, out: 1 x 2) = 3 x 6 doubles
    size_t bytes;
};
// Planes per cost-volume batch of the recurrent sweep: the -variance slices of XB consecutive planes
// come from ONE depth-sweep launch (register tap reuse along depth, cost_volume.hip) into a ring of
// XB slices, instead of one single-plane launch per step (26 -> ~6 us per plane at 400 x 300).
constexpr int XB = 16;
// Planes per synchronisation group of the wavefront (see mvs_gru_wta_batch_f32); the state ring holds RG groups of PG planes.
// (round 4, same box: PG = 2 / 4 / 8 measured 23.40 / 22.47 / 22.49 ms at one view and 72.65 / 71.52 / 70.79 ms per 4-view sweep)
constexpr int PG = 4;
// Ring depth in groups.  A cell may run RG groups ahead of the cell that consumes its states.  Round 1 used 2: the kernel
// trace showed every stream stalling ~100-200 us at EVERY group boundary -- cell k can start group j only when cell k+1
// has finished group j-2, i.e. one group time + two cross-stream signal latencies after cell k finished it, and the signal
// latency (tens of microseconds) exceeds the slack.  With 4 groups the wait is already satisfied when it is reached.
constexpr int RG = 4;
constexpr int SB = 4;          // LayerNorm-sum ring in batches of XB planes: cell 3 lags cell 1 by fewer than 2 * RG groups <= SB batches
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
GruWs carve(char* base, int H, int W, int C, int f1, int f2, int f3) {
    size_t hw = (size_t)H * W, off = 0;
    auto take = [&](size_t nfloat) { char* p = base ? base + off : nullptr; off += align256(nfloat * 4); return (float*)p; };
    GruWs w;
    const int F[3] = {f1, f2, f3};
    int fmax = f1 > f2 ? (f1 > f3 ? f1 : f3) : (f2 > f3 ? f2 : f3);
    w.x = take(hw * C * XB);
    // every cell has its own gate / candidate buffers and a ring of states: the three cells of
    // consecutive planes run concurrently (see mvs_gru_wta_batch_f32)
    for (int k = 0; k < 3; ++k) {
        w.g[k] = take(hw * 2 * F[k]); w.g2[k] = take(hw * 2 * F[k]); w.c[k] = take(hw * F[k]);
        for (int r = 0; r < RG * PG; ++r) w.h[k][r] = take(hw * F[k]);
    }
    w.rh = take(hw * fmax); w.u = take(hw * fmax);
    w.reg = take(hw); w.max_prob = take(hw); w.exp_sum = take(hw); w.depth = take(hw);
    w.px = take((size_t)2 * XB * hw * 3 * f1);
    w.wx = take((size_t)9 * C * 3 * f1); w.wgh = take((size_t)9 * f1 * 2 * f1); w.woh = take((size_t)9 * f1 * f1);
    w.wfg = take((size_t)9 * (C + f1) * 2 * f1); w.wfo = take((size_t)9 * (C + f1) * f1);
    w.stats = (double*)(base ? base + off : nullptr); off += align256((size_t)(SB + 1) * XB * 18 * 8);   // SB + 1 batches deep
    w.wsg = take(2 * 720 + 288 + 20); w.wsc = take(720 + 288);
    w.fstats = (double*)(base ? base + off : nullptr); off += align256((size_t)GRU_FUSED_RING * GRU_FUSED_ROW * 8);
    w.bytes = off;
    return w;
}

// ---- the sweep's side streams -----------------------------------------------------------------------------------------
// Cell 1 (the recurrent chain) runs on the CALLER's stream; cell 2, cell 3 (+ WTA) and the per-batch producer (cost slices +
// hoisted x-part of cell 1) run on three library-owned streams.  WHICH streams matters (round 3, profiles/r03_gru_bisect*.log,
// r03_pipe_probe.txt): the runtime binds a stream to a hardware queue on its first use, hardware queues are dealt round-robin
// over the FOUR compute pipes of the command processor in creation order (queue ids k and k + 4 share a pipe), and a queue
// that is stalled on an event wait -- or busy with the chain's ~8 dispatches per plane -- slows the dispatches of the other
// queue of its pipe 3-19x (a chain of 200 dependent empty kernels: 0.31 ms alone, 0.59 ms with any other queue stalled,
// 1.6-5.8 ms with the stalled queue on the same pipe).  Round 2 created three side streams on first use and took whatever
// queue ids came: with the caller's queue created first and nothing else in the process they were k+1..k+3 (23 ms per c3
// depth map); with ONE unrelated stream used in between (any torch.cuda.Stream that ran a kernel) the producer or cell 3
// landed on the chain's pipe and the same sweep took 44 ms.  Now: eight candidate streams per caller stream, four of the high
// and four of the low priority class, hardware queues created back to back (ids k..k+7: the candidates of a class sit on four
// different pipes, high[m] and low[m] on the same one), and ONE calibration on first use finds the candidate pipe the
// caller's queue lives on by measurement (pipe_of_caller): cells 2 / 3 take two high-priority candidates and the producer a
// low-priority one on the three OTHER pipes.
__global__ void gru_probe_empty_kernel() {}
__global__ void gru_probe_spin_kernel(long long ticks) {       // bounded: leaves after `ticks` of the 100 MHz wall clock or 2^26 polls
    const long long t0 = wall_clock64();
    for (int i = 0; i < (1 << 26); ++i)
        if (wall_clock64() - t0 > ticks) break;
}

struct GruStreams { hipStream_t cand[8], s[3]; int pipe_of_caller; float probe_us[8];
                    hipEvent_t fork, join[3], ready[2][RG], read[2][RG], xready[2], xdone[2]; };

// Index m (0..3) of the candidate pair (high[m], low[m]) that shares a compute pipe with `caller`, or -1.  While the caller
// waits on an event (as it does at the end of every sweep) a chain of 100 dependent empty kernels runs on each candidate in
// turn: the candidate on the caller's pipe takes several times as long as the others.  One-off, ~10 ms, synchronises.
int pipe_of_caller(hipStream_t caller, GruStreams& g) {
    hipEvent_t t0, t1, gate;
    if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess ||
        hipEventCreateWithFlags(&gate, hipEventDisableTiming) != hipSuccess) return -1;
    auto slowest = [&](const float* t) {                         // the one of four that stands out (> 1.7 x the median), or -1
        int m = 0;
        for (int i = 1; i < 4; ++i) if (t[i] > t[m]) m = i;
        float o[3]; int n = 0;
        for (int i = 0; i < 4; ++i) if (i != m) o[n++] = t[i];
        const float med = o[0] > o[1] ? (o[1] > o[2] ? o[1] : (o[0] > o[2] ? o[2] : o[0])) : (o[0] > o[2] ? o[0] : (o[1] > o[2] ? o[2] : o[1]));
        return t[m] > 1.7f * med ? m : -1;
    };
    // high[m] and low[m] share a pipe by construction, so the two classes must name the same m: a measurement disturbed by other
    // work on the GPU (another process, the application's own streams) is repeated, up to three times
    int mh = -1, ml = -1;
    bool ok = true;
    for (int attempt = 0; attempt < 3 && ok; ++attempt) {
        ok = hipStreamSynchronize(caller) == hipSuccess;         // the caller must be idle, or it would not be stalled on OUR wait
        for (int j = 0; ok && j < 8; ++j) {
            hipStream_t sj = g.cand[j], sg = g.cand[(j + 1) & 7];    // the gate holds the chain back while the host enqueues it
            gru_probe_spin_kernel<<<1, 64, 0, sg>>>(60000);          // 0.6 ms
            ok = ok && hipEventRecord(gate, sg) == hipSuccess && hipStreamWaitEvent(sj, gate, 0) == hipSuccess &&
                 hipEventRecord(t0, sj) == hipSuccess;
            for (int k = 0; k < 100; ++k) gru_probe_empty_kernel<<<1, 64, 0, sj>>>();
            ok = ok && hipEventRecord(t1, sj) == hipSuccess && hipStreamWaitEvent(caller, t1, 0) == hipSuccess &&
                 hipEventSynchronize(t1) == hipSuccess && hipStreamSynchronize(sg) == hipSuccess;
            float ms = 0.f;
            ok = ok && hipEventElapsedTime(&ms, t0, t1) == hipSuccess;
            g.probe_us[j] = ms * 1e3f;
        }
        if (!ok) break;
        mh = slowest(g.probe_us); ml = slowest(g.probe_us + 4);
        if (mh == ml && mh >= 0) break;                          // both classes agree
    }
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1); (void)hipEventDestroy(gate);
    if (!ok) return -1;
    if (mh >= 0 && ml >= 0 && mh != ml) return g.probe_us[mh] / g.probe_us[(mh + 1) & 3] > g.probe_us[4 + ml] / g.probe_us[4 + ((ml + 1) & 3)] ? mh : ml;
    return mh >= 0 ? mh : ml;                                    // the same pipe by construction; either measurement will do
}

// One set per (device, caller stream) -- sweeps of different reference views in flight on different caller streams must not
// share side streams, or they would serialise behind each other.  Sets are created and calibrated by mvs_gru_prepare() ONLY
// (round 4: the sweep itself used to do this on first use, i.e. create streams and synchronise inside an entry point whose
// header promises neither, and invalidate a hipGraph capture it was first called under); the sweep looks its set up and never
// creates one; mvs_gru_release() gives a slot back.
struct GruSlot { int dev; hipStream_t caller; GruStreams g; int state; };      // state: 0 free, 1 ready
constexpr int GRU_SLOTS = 16;
GruSlot g_slots[GRU_SLOTS];
std::mutex g_slots_mu;

GruStreams* gru_find(hipStream_t caller) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_slots_mu);
    for (int i = 0; i < GRU_SLOTS; ++i)
        if (g_slots[i].state == 1 && g_slots[i].dev == dev && g_slots[i].caller == caller) return &g_slots[i].g;
    return nullptr;
}

void gru_destroy(GruStreams& g) {                       // whatever of a set exists (also a half-built one)
    for (int i = 0; i < 8; ++i) if (g.cand[i]) { (void)hipStreamSynchronize(g.cand[i]); (void)hipStreamDestroy(g.cand[i]); g.cand[i] = nullptr; }
    auto ev = [](hipEvent_t& e) { if (e) { (void)hipEventDestroy(e); e = nullptr; } };
    ev(g.fork);
    for (int i = 0; i < 2; ++i) { ev(g.xready[i]); ev(g.xdone[i]); for (int j = 0; j < RG; ++j) { ev(g.ready[i][j]); ev(g.read[i][j]); } }
    for (int i = 0; i < 3; ++i) { ev(g.join[i]); g.s[i] = nullptr; }
}

// Creates and calibrates the set of `caller` on the current device (idempotent).  Synchronises `caller` and the new streams.
int gru_prepare(hipStream_t caller) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(caller, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return MVS_E_NOT_PREPARED;   // prepare synchronises
    std::lock_guard<std::mutex> lock(g_slots_mu);
    int free_slot = -1;
    for (int i = 0; i < GRU_SLOTS; ++i) {
        if (g_slots[i].state == 1 && g_slots[i].dev == dev && g_slots[i].caller == caller) return 0;
        if (g_slots[i].state == 0 && free_slot < 0) free_slot = i;
    }
    if (free_slot < 0) return MVS_E_NO_SLOT;               // GRU_SLOTS caller streams hold a set: release one first
    GruSlot& sl = g_slots[free_slot];
    sl = GruSlot{};
    sl.dev = dev; sl.caller = caller;
    GruStreams& g = sl.g;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;   // lo = least urgent: the batch producer yields to the cells
    e = hipSuccess;
    for (int i = 0; e == hipSuccess && i < 8; ++i) e = hipStreamCreateWithPriority(&g.cand[i], hipStreamNonBlocking, i < 4 ? hi : lo);
    // first use = hardware queue creation: touch the eight candidates now, in order, with nothing in between
    for (int i = 0; e == hipSuccess && i < 8; ++i) {
        gru_probe_empty_kernel<<<1, 64, 0, g.cand[i]>>>();
        e = hipStreamSynchronize(g.cand[i]);
    }
    auto ev = [&](hipEvent_t* ep) { if (e == hipSuccess) e = hipEventCreateWithFlags(ep, hipEventDisableTiming); };
    ev(&g.fork);
    for (int i = 0; i < 2; ++i) { ev(&g.xready[i]); ev(&g.xdone[i]); for (int j = 0; j < RG; ++j) { ev(&g.ready[i][j]); ev(&g.read[i][j]); } }
    for (int i = 0; i < 3; ++i) ev(&g.join[i]);
    if (e != hipSuccess) { gru_destroy(g); return (int)e; }
    g.pipe_of_caller = pipe_of_caller(caller, g);
    if (g.pipe_of_caller < 0) {
        // no candidate pipe stood out (other work on the GPU during the ~10 ms measurement, or a runtime that deals queues
        // differently): the sweep still runs as a wavefront, but one of its side streams may share the caller's compute
        // pipe -- the 2x slow layout of round 2 (profiles/r03_gru_bisect*.log).  Say so, once per process.
        static bool told = false;
        if (!told) { told = true; fprintf(stderr, "mvsnet_hip: mvs_gru_prepare: the stream-layout calibration was inconclusive (chains %.0f %.0f %.0f %.0f | %.0f %.0f %.0f %.0f us); "
                                                  "the recurrent sweep may run up to 2x slower on this stream -- call mvs_gru_release + mvs_gru_prepare again on an idle GPU\n",
                                          g.probe_us[0], g.probe_us[1], g.probe_us[2], g.probe_us[3], g.probe_us[4], g.probe_us[5], g.probe_us[6], g.probe_us[7]); }
    }
    int pick[3], n = 0;                                  // the three candidate pipes the caller's queue is NOT on
    for (int m = 0; m < 4 && n < 3; ++m) if (m != g.pipe_of_caller) pick[n++] = m;
    g.s[0] = g.cand[pick[0]]; g.s[1] = g.cand[pick[1]]; g.s[2] = g.cand[4 + pick[2]];
    for (int i = 0; i < 8; ++i)                          // the five candidates that lost go back (their hardware queues with them)
        if (g.cand[i] != g.s[0] && g.cand[i] != g.s[1] && g.cand[i] != g.s[2]) { (void)hipStreamDestroy(g.cand[i]); g.cand[i] = nullptr; }
    sl.state = 1;
    return 0;
}

int gru_release(hipStream_t caller) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lock(g_slots_mu);
    for (int i = 0; i < GRU_SLOTS; ++i)
        if (g_slots[i].state == 1 && g_slots[i].dev == dev && g_slots[i].caller == caller) {
            gru_destroy(g_slots[i].g);                   // waits for the side streams' work
            g_slots[i].state = 0;
            return 0;
        }
    return MVS_E_BADARG;
}

__global__ void __launch_bounds__(256)
zero_views_kernel(float4* p, size_t n4, size_t vstride) {       // blockIdx.y = view
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) view_ptr(p, (size_t)blockIdx.y * vstride)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void __launch_bounds__(256)
wta_finish_views_kernel(const float* max_prob, const float* exp_sum, const float* depth, int HW, size_t vstride,
                        float* __restrict__ depth_out, float* __restrict__ prob_out) {      // outputs (views, H, W)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const size_t vo = (size_t)blockIdx.y * vstride, o = (size_t)blockIdx.y * HW + i;
    prob_out[o] = view_ptr(max_prob, vo)[i] / (view_ptr(exp_sum, vo)[i] + 1e-7f);          // model.py:749-751
    depth_out[o] = view_ptr(depth, vo)[i];
}
}  // namespace

// One side stream (and a fork / join event pair) of the set mvs_gru_prepare made for `caller`, for other users of the library's
// stream sets (regnet.hip: a branch layer beside the low-resolution chain); false without a set.
bool mvs_stream_set_side(hipStream_t caller, hipStream_t* side, hipEvent_t* fork, hipEvent_t* join) {
    GruStreams* g = gru_find(caller);
    if (!g) return false;
    *side = g->s[0]; *fork = g->fork; *join = g->join[0];
    return true;
}

// shape part of the routing decision of mvs_gru_wta*_f32 (exported for the CPU-side routing test; tests/test_abi_and_io.py)
extern "C" int mvs_gru_fused_route(int C, int f1, int f2, int f3, size_t view_block_bytes) {
    return C == 32 && f1 == 16 && f2 == 4 && f3 == 2 && view_block_bytes < ((size_t)1 << 31);
}

extern "C" size_t mvs_gru_workspace_bytes(int H, int W, int C, int f1, int f2, int f3) {
    return carve(nullptr, H, W, C, f1, f2, f3).bytes;
}

// Formulation of cell 1 on the MFMA kernels: 1 = hoisted x-part (its own batched launches on the producer stream + per-plane
// kernels over the 16 state channels), 2 = full 48-channel per-plane kernels, 0 = by view count (hoisted for one view: the
// chain's latency paces a single sweep; full for two or more: B x 950 tiles per launch hide it, and the hoisted form's px
// tensor -- 46 MB of traffic per plane and view -- and its producer competing for the matrix pipes are what is left to save).
// Both give the same bits (gru_mfma.hip, SPLIT accumulators).
static std::atomic<int> g_gru_form{0};                 // read ONCE per sweep (a sweep in flight keeps the formulation it started with)
extern "C" int mvs_gru_set_formulation(int form) {
    if (form < 0 || form > 3) return MVS_E_BADARG;
    g_gru_form.store(form);
    return 0;
}

extern "C" int mvs_gru_prepare(void* stream) { return gru_prepare(mvs_stream(stream)); }
extern "C" int mvs_gru_release(void* stream) { return gru_release(mvs_stream(stream)); }

extern "C" int mvs_gru_stream_layout(void* stream, int* pipe_of_caller_out, float* probe_us_out) {
    GruStreams* gs = gru_find(mvs_stream(stream));
    if (!gs) return MVS_E_NOT_PREPARED;
    if (pipe_of_caller_out) *pipe_of_caller_out = gs->pipe_of_caller;
    if (probe_us_out) for (int i = 0; i < 8; ++i) probe_us_out[i] = gs->probe_us[i];
    return 0;
}

extern "C" int mvs_gru_wta_batch_f32(const float* const* ref, const float* const* src, const float* const* transforms,
                                     int views, int view_num, int depth_num, int H, int W, int C, int f1, int f2,
                                     int f3, const float* const* params, const float* depth_values,
                                     void* workspace, size_t workspace_bytes, float* depth_out,
                                     float* prob_out, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && params && depth_values && workspace && depth_out && prob_out);
    MVS_CHECK_ARG(views >= 1 && views <= MAXV);
    MVS_CHECK_ARG(view_num >= 2 && depth_num >= 1 && H > 0 && W > 0 && C > 0 && f1 > 0 && f2 > 0 && f3 > 0);
    if (f1 > 64 || f2 > 64 || f3 > 64) return MVS_E_SHAPE;     // gru_blend_fused_kernel keeps 2 x F LayerNorm affines in LDS ('fat': 32)
    for (int v = 0; v < views; ++v) MVS_CHECK_ARG(ref[v] && src[v] && transforms[v]);
    GruWs ws = carve((char*)workspace, H, W, C, f1, f2, f3);      // view 0's block; view v's tensors are v * ws.bytes further
    const size_t vstride = ws.bytes;
    if (workspace_bytes < vstride * (size_t)views) return MVS_E_WORKSPACE;
    const Views vw = {views, vstride};
    const hipStream_t st = mvs_stream(stream);           // the caller's stream carries cell 1, the recurrent chain
    const size_t hw = (size_t)H * W;
    hipError_t e;
    const int F[3] = {f1, f2, f3};
    auto vp = [&](auto* p, int v) { return (decltype(p))((char*)p + (size_t)v * vstride); };     // host-side view pointer
    auto zero = [&](float* p, size_t nfloat) -> int {    // the same tensor of every view (sizes are multiples of 4 floats: 256-byte carving)
        const size_t n4 = (nfloat + 3) / 4;
        zero_views_kernel<<<dim3(mvs_cdiv((long long)n4, 256), views), 256, 0, st>>>((float4*)p, n4, vstride);
        return (int)hipGetLastError();
    };
    int rc;
    // zero initial states and WTA accumulators (model.py:649-654, 737-739)
    for (int k = 0; k < 3; ++k) if ((rc = zero(ws.h[k][0], hw * F[k]))) return rc;
    if ((rc = zero(ws.max_prob, hw)) || (rc = zero(ws.exp_sum, hw)) || (rc = zero(ws.depth, hw))) return rc;

    // cell 1 (90 % of the MACs) runs on the fp32-MFMA kernels when its shape fits their tiling
    bool mfma1 = (mvs_get_conv_impl() != MVS_CONV_IMPL_SCALAR) && C == 32 && f1 == 16;
    const int form = g_gru_form.load();
    // The fused sweep (gru_fused.hip): the reference's filter counts (model.py:641-660, 'normal' mode) at 32 feature channels --
    // all three cells, prob_conv and the winner-take-all update in two launches per plane on the caller's stream alone.
    // Formulation 0 takes it whenever the shape fits -- also for ONE eager view on a stream that has a stream set, where the round-4
    // wavefront with the hoisted x-part (formulation 1) is still 3-4 % faster (c3, same box: 21.8 against 22.6 ms): a batch of views
    // must give the single view's bits, and from two views per sweep on, and under hipGraph capture (23.4 against 37 ms), the fused
    // sweep is the faster or equal one (profiles/r05_gru_ab_forms.txt).
    // The fused kernels address a whole view's workspace block through one buffer resource with 32-bit byte offsets: blocks of
    // 2 GiB and more (~10.2 KB per pixel: feature maps above ~210 k pixels, e.g. 576 x 384) take the wavefront route, which
    // addresses every tensor by itself -- decided HERE, before anything is enqueued.
    const bool fused = mvs_gru_fused_route(C, f1, f2, f3, vstride) && mfma1 && (form == 0 || form == 3);
    if (fused) {
        GruFusedWs fw;
        fw.base = (char*)workspace; fw.x = ws.x;
        for (int k = 0; k < 3; ++k) { fw.S[k][0] = ws.h[k][0]; fw.S[k][1] = ws.h[k][1]; fw.G[k][0] = ws.g[k]; fw.G[k][1] = ws.g2[k]; fw.Cb[k] = ws.c[k]; }
        fw.stats = ws.fstats; fw.max_prob = ws.max_prob; fw.depth = ws.depth; fw.exp_sum = ws.exp_sum;
        fw.w1g = ws.wfg; fw.w1c = ws.wfo; fw.wsg = ws.wsg; fw.wsc = ws.wsc;
        if ((rc = mvs_gru1_full_weights(params[0], params[6], C, f1, ws.wfg, ws.wfo, st))) return rc;
        if ((rc = mvs_gru_fused_prepare_weights(params, fw, st))) return rc;
        for (int k = 0; k < 3; ++k) if ((rc = zero(ws.h[k][1], hw * F[k]))) return rc;      // s(-1) = 0 lives in S[k][1]; S[k][0] zeroed above
        if ((rc = zero((float*)ws.fstats, (size_t)GRU_FUSED_RING * GRU_FUSED_ROW * 2))) return rc;
        // The -variance cost slices (model.py:680-693,698) come in batches of XB planes from one depth-sweep launch per view.  With a
        // stream set (mvs_gru_prepare) the producer runs ONE BATCH AHEAD on the set's low-priority stream, into the other half of
        // a two-batch buffer (the px tensor of the wavefront formulations, unused here): its waves fill the issue slots the
        // recurrent launches leave idle instead of standing in line with them (31 of 287 us per plane at four views).  Without a
        // set, and under hipGraph capture, everything stays on the caller's stream.
        hipStreamCaptureStatus cs0 = hipStreamCaptureStatusNone;
        const bool capturing0 = hipStreamIsCapturing(st, &cs0) == hipSuccess && cs0 != hipStreamCaptureStatusNone;
        GruStreams* pg = (!capturing0 && depth_num > XB && !mvs_hook(MVS_HOOK_GRU_ONE_STREAM)) ? gru_find(st) : nullptr;
        const hipStream_t sx = pg ? pg->s[2] : st;
        const int small_wg = mvs_hook(MVS_HOOK_GRU_PRODUCER_THREADS);      // A/B hook, validated by mvs_set_test_hook (64 / 128 / 192 / 256)
        auto xhalf = [&](int bidx) -> float* { return pg ? ws.px + (size_t)(bidx & 1) * XB * hw * C : ws.x; };
        auto produce = [&](int bidx, hipStream_t s) -> int {
            const int t0 = bidx * XB, nb = depth_num - t0 < XB ? depth_num - t0 : XB;
            for (int v = 0; v < views; ++v) {
                // (on the producer stream: 128-thread workgroups, 16 KB of LDS -- they fit beside a fused workgroup)
                const int r2 = mvs_cost_volume_threads_f32(ref[v], src[v], transforms[v], view_num, depth_num, t0, nb, H, W, C, /*variant*/ 1,
                                                           /*negate*/ 1, /*border*/ 0, vp(xhalf(bidx), v), (pg && s == sx) ? small_wg : 256, s);
                if (r2) return r2;
            }
            return 0;
        };
        bool forked0 = false;
        auto run = [&]() -> int {
            hipError_t e0;
            if (pg) {
                if ((e0 = hipEventRecord(pg->fork, st)) != hipSuccess) return (int)e0;
                forked0 = true;
                if ((e0 = hipStreamWaitEvent(sx, pg->fork, 0)) != hipSuccess) return (int)e0;
                if ((rc = produce(0, sx))) return rc;
                if ((e0 = hipEventRecord(pg->xready[0], sx)) != hipSuccess) return (int)e0;
            }
            for (int t = 0; t < depth_num + 3; ++t) {
                if (t % XB == 0) {
                    const int bidx = t / XB;
                    if (t < depth_num) {
                        if (!pg) { if ((rc = produce(bidx, st))) return rc; }
                        else {
                            if ((e0 = hipStreamWaitEvent(st, pg->xready[bidx & 1], 0)) != hipSuccess) return (int)e0;
                            if ((bidx + 1) * XB < depth_num) {       // the next batch into the other half, once its readers (batch bidx - 1) are done
                                if (bidx >= 1 && ((e0 = hipEventRecord(pg->xdone[(bidx + 1) & 1], st)) != hipSuccess ||
                                                  (e0 = hipStreamWaitEvent(sx, pg->xdone[(bidx + 1) & 1], 0)) != hipSuccess)) return (int)e0;
                                if ((rc = produce(bidx + 1, sx))) return rc;
                                if ((e0 = hipEventRecord(pg->xready[(bidx + 1) & 1], sx)) != hipSuccess) return (int)e0;
                            }
                        }
                    }
                    // LayerNorm-sum rows of planes t + XB .. t + 2 XB - 1 (their previous users, planes 64 earlier, are long done)
                    if (t > 0 && (rc = zero((float*)(ws.fstats + (size_t)((t + XB) % GRU_FUSED_RING) * GRU_FUSED_ROW), (size_t)XB * GRU_FUSED_ROW * 2))) return rc;
                }
                const int tx = t < depth_num ? t : depth_num - 1;
                if ((rc = mvs_gru_fused_step(fw, params, t, depth_num, xhalf(tx / XB) + (size_t)(tx % XB) * hw * C, H, W, views, vstride, depth_values, st))) return rc;
            }
            return 0;
        };
        rc = run();
        if (forked0) {                                   // join on every exit after the fork
            hipError_t e1 = hipEventRecord(pg->join[2], sx);
            if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, pg->join[2], 0);
            if (e1 != hipSuccess && rc == 0) rc = (int)e1;
        }
        if (rc) return rc;
        wta_finish_views_kernel<<<dim3(mvs_cdiv((long long)hw, 256), views), 256, 0, st>>>(ws.max_prob, ws.exp_sum, ws.depth, H * W, vstride,
                                                                                         depth_out, prob_out);
        return (int)hipGetLastError();
    }
    const bool hoist = mfma1 && (form == 1 || (form == 0 && views == 1));
    if (mfma1) {                                         // prepared weights, shared by the views
        if (hoist) rc = mvs_gru1_split_weights(params[0], params[6], C, f1, ws.wx, ws.wgh, ws.woh, st);
        else rc = mvs_gru1_full_weights(params[0], params[6], C, f1, ws.wfg, ws.wfo, st);
        if (rc) return rc;
    }
    // which kernels each cell gets; the generic conv + gates route shares rh / u and stays on one stream
    const int cins[3] = {C, f1, f2};
    int route[3];                                    // 0 generic, 1 MFMA (cell 1), 2 small-cell kernels
    for (int k = 0; k < 3; ++k) {
        const int ci = cins[k], f = F[k];
        route[k] = (k == 0 && mfma1) ? 1
                 : ((ci == 16 && f == 4) || (ci == 4 && f == 2) || (ci == 8 && f == 2) || (ci == 2 && f == 1)) ? 2 : 0;
    }
    // Wavefront over (plane, cell): cell k of plane d needs cell k-1 of plane d and cell k of plane d-1, so
    // the three cells run on three streams, cell 1 of the next planes alongside cell 2 of these and cell 3 of
    // the previous ones; the small kernels of cells 2 / 3 (launch-latency bound, a few workgroups per CU) then
    // fill the machine under cell 1's kernels instead of serialising behind them.  A cross-stream dependency
    // costs tens of microseconds of signal latency, as much as a cell's kernels for one plane, so the streams
    // synchronise per GROUP of PG planes: states live in a ring of RG*PG planes (plane d reads h[k][d % ring], writes
    // h[k][(d+1) % ring]); per group j, ready[k][j % RG] = cell k has written its states of group j, read[k][j % RG] =
    // cell k+1 is done reading them (cell k may overwrite those ring slots in group j+RG).
    // Several reference views (views > 1) ride in the SAME launches: every kernel of the sweep takes a view index from its
    // grid, so the ~7 launches per plane, their ~5 us floors and the cross-stream waits are shared by `views` depth maps.
    const int ring = RG * PG;
    const bool wavefront = route[0] && route[1] && route[2] && depth_num > 2 * PG &&
                           !mvs_hook(MVS_HOOK_GRU_ONE_STREAM);      // test hook (parity of the one-stream sweep), read per sweep
    // Under hipGraph capture the WAVEFRONT formulations go to the caller's stream alone.  Root cause of the round-4 host crash
    // (profiles/r05_capture_wavefront_root_cause.txt): hip::Stream::EndCapture() of the HIP runtime the PyTorch wheel bundles
    // (libamdhip64 7.0.70002, the runtime every Python process of this library runs on) recurses without bound once two captured
    // streams wait on each other's events in BOTH directions -- which the ring's backward waits (`read`) do from the fifth group on;
    // tools/capture_wavefront_repro.hip reproduces it with empty kernels (stack overflow from 40 planes on with that runtime, clean
    // with /opt/rocm 7.2's; stream flags, priorities, event re-use and capture mode do not matter).  Nothing in this event graph is
    // illegal.  The default formulation (gru_fused.hip) has no cross-stream pattern and is captured at full speed.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    // (tools/capture_wavefront_repro.hip is the standalone evidence; the library has no switch that re-enables the crashing path)
    GruStreams* gs = (wavefront && !capturing) ? gru_find(st) : nullptr;
    if (wavefront && !capturing && !gs) {
        // no side streams for this caller stream: this entry point creates none (mvs_gru_prepare does) -- run the sweep on the
        // caller's stream alone (same results, ~1.7x the time at 400 x 300) and say so once.
        static std::atomic<bool> told{false};
        if (!told.exchange(true))
            fprintf(stderr, "mvsnet_hip: mvs_gru_wta*_f32 on a stream without mvs_gru_prepare(): the recurrent sweep runs on this stream "
                            "alone (same results, slower); call mvs_gru_prepare(stream) once per caller stream\n");
    }
    // (cells 2 and 3 sharing ONE side stream, round 4: 28.6 against 22.5 ms at one view, 74.4 against 71.5 ms per 4-view sweep)
    hipStream_t sk[3] = {st, gs ? gs->s[0] : st, gs ? gs->s[1] : st};
    const long long hw_ll = (long long)H * W;
    bool forked = false;
    auto sweep = [&]() -> int {
    if (gs) {
        if ((e = hipEventRecord(gs->fork, st)) != hipSuccess) return (int)e;
        forked = true;
        for (int i = 0; i < 3; ++i) if ((e = hipStreamWaitEvent(gs->s[i], gs->fork, 0)) != hipSuccess) return (int)e;
    }

    // The cost slices of the batch that holds plane d.  Hoisted form: one buffer -- the x-part launches that read it follow the
    // slices' launch on the producer stream, and what the chain reads (px) has two halves.  Full form: the chain reads the
    // slices themselves while the producer writes the next batch, so they alternate between two halves too (the memory of
    // the unused px tensor: 2 * XB * 48 floats per pixel against the 2 * XB * C needed).
    auto xbatch = [&](int d) -> float* {
        return (mfma1 && !hoist && gs) ? ws.px + (size_t)((d / XB) & 1) * XB * hw * C : ws.x;
    };
    // start of a batch of XB planes (on cell 1's stream, before its first plane of the batch)
    auto batch_start = [&](int d) -> int {
        const int half = (d / XB) & 1;
        // LayerNorm sums of this batch (other slots of the ring may still be in use by cells 2 / 3 of earlier planes: cell 3
        // lags cell 1 by fewer than 2 * RG groups = at most SB batches)
        int r;
        if ((r = zero((float*)(ws.stats + (size_t)((d / XB) % (SB + 1)) * XB * 18), (size_t)XB * 18 * 2))) return r;
        // Per batch: x = -variance cost slices (model.py:680-693,698) and, for the MFMA cell 1, the x halves of
        // its two convolutions for the whole batch (gru_mfma.hip: x-part hoisting).
        auto produce = [&](int d0, hipStream_t s) -> int {
            const int nb = depth_num - d0 < XB ? depth_num - d0 : XB;
            for (int v = 0; v < views; ++v) {
                int r2 = mvs_cost_volume_f32(ref[v], src[v], transforms[v], view_num, depth_num, d0, nb, H, W, C, /*variant*/ 1,
                                             /*negate*/ 1, /*border*/ 0, vp(xbatch(d0), v), s);
                if (r2) return r2;
                if (hoist && (r2 = mvs_gru1_xpart_mfma(vp(xbatch(d0), v), ws.wx, ws.wx + (size_t)9 * C * 2 * f1, params[1], params[7], H, W, nb,
                                                       vp(ws.px, v) + (size_t)((d0 / XB) & 1) * XB * hw * 3 * f1, s))) return r2;
            }
            return 0;
        };
        if (!(gs && mfma1)) return produce(d, st);
        // the producer runs one batch ahead on its own (low-priority) stream: 2/3 of cell 1's MACs leave the
        // recurrent chain
        hipStream_t sx = gs->s[2];
        if (d == 0) {
            if ((r = produce(0, sx))) return r;
            if ((e = hipEventRecord(gs->xready[0], sx)) != hipSuccess) return (int)e;
        }
        if ((e = hipStreamWaitEvent(st, gs->xready[half], 0)) != hipSuccess) return (int)e;
        if (d + XB < depth_num) {                     // next batch into the other half, once its readers are done
            if (d >= XB && (e = hipStreamWaitEvent(sx, gs->xdone[half ^ 1], 0)) != hipSuccess) return (int)e;
            if ((r = produce(d + XB, sx))) return r;
            if ((e = hipEventRecord(gs->xready[half ^ 1], sx)) != hipSuccess) return (int)e;
        }
        return 0;
    };

    // cell k of plane d on stream s: gate conv, candidate conv, blend (+ prob / WTA after cell 3)
    auto cell_plane = [&](int k, int d, hipStream_t s) -> int {
        const int slot = d % XB, half = (d / XB) & 1;
        const float* const* p = params + 10 * k;
        double* sg = ws.stats + (size_t)(((d / XB) % (SB + 1)) * XB + slot) * 18 + 6 * k;
        double* so = sg + 4;
        float* hp_w = ws.h[k][d % ring];
        const float* hp = hp_w;
        float* hn = ws.h[k][(d + 1) % ring];
        // On the wavefront: the blend of a plane is folded into the NEXT plane's gate convolution (its
        // staging forms the state it convolves) except at the end of a synchronisation group, where the next cell
        // (and the WTA update) are about to read the state: 3 of 4 blend launches disappear.  The gate buffer
        // alternates, a folded gate convolution still reads the previous plane's update gate while it writes its own.
        const bool fold = route[k] != 0 && gs != nullptr;           // (the MFMA cell 1 folds its blend the same way)
        const bool fused_in = fold && d % PG != 0;               // plane d-1 left its blend to this plane
        const bool blend_now = !fold || d % PG == PG - 1 || d == depth_num - 1;
        float* gcur = (fold && (d & 1)) ? ws.g2[k] : ws.g[k];
        PrevPlane prev = {nullptr, nullptr, nullptr, nullptr};
        if (fused_in) {
            const int dp = d - 1;
            double* sgp = ws.stats + (size_t)(((dp / XB) % (SB + 1)) * XB + dp % XB) * 18 + 6 * k;
            prev = {ws.h[k][dp % ring], (dp & 1) ? ws.g2[k] : ws.g[k], sgp, sgp + 4};
        }
        const PrevPlane* pv = fused_in ? &prev : nullptr;
        const float* xin = k == 0 ? xbatch(d) + (size_t)slot * hw * C : ws.h[k - 1][(d + 1) % ring];
        const float* px_d = ws.px + ((size_t)half * XB + slot) * hw * 3 * f1;
        const int cin = cins[k];
        int r;
        // cell 3: the previous plane's prob_conv + winner-take-all update rides in this plane's gate convolution, which forms
        // that plane's final state in its tile anyway
        WtaFold wfold = {params[30], params[31], {}, ws.max_prob, ws.depth, ws.exp_sum};
        for (int v = 0; v < views; ++v) wfold.depth_value[v] = d >= 1 ? depth_values[(size_t)v * depth_num + d - 1] : 0.f;
        const WtaFold* wf = (k == 2 && fused_in) ? &wfold : nullptr;
        bool wta_folded = false;
        if (route[k] == 1 && !hoist) {
            if (fused_in)
                r = mvs_gru1_gates_full_blend_mfma(xin, prev.h_before, ws.c[k], prev.g, prev.so, prev.sg + 2, p[8], p[9], p[4], p[5], hp_w,
                                                   ws.wfg, p[1], H, W, gcur, sg, views, vstride, s);
            else
                r = mvs_gru1_gates_full_mfma(xin, hp, ws.wfg, p[1], H, W, gcur, sg, views, vstride, s);
            if (r) return r;
            if ((r = mvs_gru1_out_full_mfma(xin, hp, gcur, sg, p[2], p[3], ws.wfo, p[7], H, W, ws.c[k], so, views, vstride, s))) return r;
            if (gs && (slot == XB - 1 || d == depth_num - 1) && (e = hipEventRecord(gs->xdone[half], s)) != hipSuccess) return (int)e;
        } else if (route[k] == 1) {
            if (fused_in)
                r = mvs_gru1_gates_h_blend_mfma(prev.h_before, ws.c[k], prev.g, prev.so, prev.sg + 2, p[8], p[9], p[4], p[5], hp_w,
                                                ws.wgh, px_d, H, W, gcur, sg, views, vstride, s);
            else
                r = mvs_gru1_gates_h_mfma(hp, ws.wgh, px_d, H, W, gcur, sg, views, vstride, s);
            if (r) return r;
            if ((r = mvs_gru1_out_h_mfma(hp, gcur, sg, p[2], p[3], ws.woh, px_d, H, W, ws.c[k], so, views, vstride, s))) return r;
            if (gs && (slot == XB - 1 || d == depth_num - 1) && (e = hipEventRecord(gs->xdone[half], s)) != hipSuccess) return (int)e;
        } else if (cin == 16 && F[k] == 4 && launch_small_cell<16, 4>(xin, hp_w, p, H, W, gcur, ws.c[k], sg, so, pv, vw, s, wf)) {
            wta_folded = wf != nullptr;
        } else if (cin == 4 && F[k] == 2 && launch_small_cell<4, 2>(xin, hp_w, p, H, W, gcur, ws.c[k], sg, so, pv, vw, s, wf)) {
            wta_folded = wf != nullptr;
        } else if (cin == 8 && F[k] == 2 && launch_small_cell<8, 2>(xin, hp_w, p, H, W, gcur, ws.c[k], sg, so, pv, vw, s, wf)) {
            wta_folded = wf != nullptr;
        } else if (cin == 2 && F[k] == 1 && launch_small_cell<2, 1>(xin, hp_w, p, H, W, gcur, ws.c[k], sg, so, pv, vw, s, wf)) {
            wta_folded = wf != nullptr;
        } else {
            for (int v = 0; v < views; ++v) {            // shape-generic route: one view per launch
                if ((r = launch_conv2d(vp(xin, v), cin, vp(hp, v), F[k], p[0], p[1], H, W, 2 * F[k], vp(ws.g[k], v), vp(sg, v), 2, s))) return r;
                if ((r = mvs_gru_gates_f32(vp(ws.g[k], v), vp(sg, v), p[2], p[3], p[4], p[5], vp(hp, v), H, W, F[k], vp(ws.rh, v), vp(ws.u, v), s))) return r;
                if ((r = launch_conv2d(vp(xin, v), cin, vp(ws.rh, v), F[k], p[6], p[7], H, W, F[k], vp(ws.c[k], v), vp(so, v), 1, s))) return r;
            }
        }
        // prob_conv + exp + winner-take-all update (model.py:701-731) of plane `dd`, whose final state is `hs`
        auto prob_wta = [&](const float* hs, int dd) -> int {
            const dim3 grid(mvs_cdiv(hw_ll, 256), views);
            DepthVals dv = {};
            for (int v = 0; v < views; ++v) dv.v[v] = depth_values[(size_t)v * depth_num + dd];
            int r2;
            switch (f3) {
                case 1: prob_wta_kernel<1><<<grid, 256, 0, s>>>(hs, params[30], params[31], dv, H, W, ws.max_prob, ws.depth, ws.exp_sum, vstride); break;
                case 2: prob_wta_kernel<2><<<grid, 256, 0, s>>>(hs, params[30], params[31], dv, H, W, ws.max_prob, ws.depth, ws.exp_sum, vstride); break;
                case 4: prob_wta_kernel<4><<<grid, 256, 0, s>>>(hs, params[30], params[31], dv, H, W, ws.max_prob, ws.depth, ws.exp_sum, vstride); break;
                default:
                    for (int v = 0; v < views; ++v) {
                        if ((r2 = launch_conv2d(vp(hs, v), f3, nullptr, 0, params[30], params[31], H, W, 1, vp(ws.reg, v), nullptr, 1, s))) return r2;
                        if ((r2 = mvs_wta_update_f32(vp(ws.reg, v), dv.v[v], H, W, vp(ws.max_prob, v), vp(ws.depth, v), vp(ws.exp_sum, v), s))) return r2;
                    }
            }
            return (int)hipGetLastError();
        };
        if (k == 2 && fused_in && !wta_folded && (r = prob_wta(hp, d - 1))) return r;     // plane d-1's state exists since this plane's gate convolution
        if (blend_now) {
            const int vec = F[k] % 4 == 0 ? 4 : F[k] % 2 == 0 ? 2 : 1;
            const dim3 grid(mvs_cdiv(hw_ll * F[k] / vec, 256), views);
            if (vec == 4)
                gru_blend_fused_kernel<4><<<grid, 256, 0, s>>>(ws.c[k], so, p[8], p[9], gcur, sg + 2, p[4], p[5], H * W, F[k], hp, hn, vstride);
            else if (vec == 2)
                gru_blend_fused_kernel<2><<<grid, 256, 0, s>>>(ws.c[k], so, p[8], p[9], gcur, sg + 2, p[4], p[5], H * W, F[k], hp, hn, vstride);
            else
                gru_blend_fused_kernel<1><<<grid, 256, 0, s>>>(ws.c[k], so, p[8], p[9], gcur, sg + 2, p[4], p[5], H * W, F[k], hp, hn, vstride);
            if ((r = (int)hipGetLastError())) return r;
            if (k == 2 && (r = prob_wta(hn, d))) return r;
        }
        return (int)hipGetLastError();
    };

    if (!gs) {
        // one stream: planes in order, cells in order
        for (int d = 0; d < depth_num; ++d) {
            if (d % XB == 0 && (rc = batch_start(d))) return rc;
            for (int k = 0; k < 3; ++k) if ((rc = cell_plane(k, d, st))) return rc;
        }
    } else {
        for (int j = 0, d0 = 0; d0 < depth_num; ++j, d0 += PG) {
            const int d1 = d0 + PG < depth_num ? d0 + PG : depth_num, jp = j % RG;
            for (int k = 0; k < 3; ++k) {
                hipStream_t s = sk[k];
                // the states cell k wrote in this group ...
                if (k > 0 && (e = hipStreamWaitEvent(s, gs->ready[k - 1][jp], 0)) != hipSuccess) return (int)e;
                // ... and the ring slots this cell is about to overwrite were read by cell k+1 RG groups ago
                if (k < 2 && j >= RG && (e = hipStreamWaitEvent(s, gs->read[k][jp], 0)) != hipSuccess) return (int)e;
                for (int d = d0; d < d1; ++d) {
                    if (k == 0 && d % XB == 0 && (rc = batch_start(d))) return rc;
                    if ((rc = cell_plane(k, d, s))) return rc;
                }
                if (k < 2 && (e = hipEventRecord(gs->ready[k][jp], s)) != hipSuccess) return (int)e;
                if (k > 0 && (e = hipEventRecord(gs->read[k - 1][jp], s)) != hipSuccess) return (int)e;
            }
        }
    }
    return 0;
    };       // sweep
    rc = sweep();
    // Join on EVERY exit after the fork: also when a launch failed half-way the caller's stream must stay ordered
    // after whatever the side streams were given -- the caller frees or re-uses the feature maps and the workspace in stream order.
    if (forked)
        for (int i = 0; i < 3; ++i) {
            hipError_t e1 = hipEventRecord(gs->join[i], gs->s[i]);
            if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, gs->join[i], 0);
            if (e1 != hipSuccess && rc == 0) rc = (int)e1;
        }
    if (rc) return rc;
    wta_finish_views_kernel<<<dim3(mvs_cdiv(hw_ll, 256), views), 256, 0, st>>>(ws.max_prob, ws.exp_sum, ws.depth, H * W, vstride,
                                                                             depth_out, prob_out);
    return (int)hipGetLastError();
}

extern "C" int mvs_gru_wta_f32(const float* ref, const float* src, const float* transforms,
                               int view_num, int depth_num, int H, int W, int C, int f1, int f2,
                               int f3, const float* const* params, const float* depth_values,
                               void* workspace, size_t workspace_bytes, float* depth_out,
                               float* prob_out, void* stream) {
    return mvs_gru_wta_batch_f32(&ref, &src, &transforms, 1, view_num, depth_num, H, W, C, f1, f2, f3, params, depth_values,
                                 workspace, workspace_bytes, depth_out, prob_out, stream);
}
