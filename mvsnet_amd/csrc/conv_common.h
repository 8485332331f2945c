// Shared pieces of the fp32-MFMA 3D convolution kernels (conv3d_mfma.hip, conv3d_s2_mfma.hip,
// deconv3d_mfma.hip, conv3d_out.hip).
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Producer BatchNorm given as raw float64 sums: the consumer folds them into (scale, shift) itself
// (same arithmetic as bn_finalize_kernel), which removes one launch per layer.
struct BnSrc {
    const double* stats;      // (2,C) [sum, sum of squares] or null
    const float* gamma; const float* beta;
    double count; float eps; int C;
    int nslot;                // stats is (nslot, 2, C): partial rows the consumer adds up (0 / 1: a single row)
};

// mean and variance in float64 (E[x^2] - E[x]^2 cancels), the reciprocal square root in float32 (v_rsq_f32, 1 ulp) as
// TensorFlow's fused batch norm takes it: the float64 divisions and square root of a literal transcription are ~600
// double-rate instructions per thread in EVERY workgroup's prologue (round 2: the consumers' prologues showed up as ~1.6 us
// per 240-workgroup layer).
__device__ __forceinline__ void bn_affine_finish(double s1, double s2, double inv_count, float gamma, float beta, float eps,
                                                 float& scale, float& shift) {
    const double mean = s1 * inv_count;
    double var = s2 * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float inv = gamma * __builtin_amdgcn_rsqf((float)var + eps);
    scale = inv;
    shift = beta - (float)mean * inv;
}

__device__ __forceinline__ void bn_affine4(const BnSrc& b, int c0, float4& sc, float4& sh) {
    float s[4], t[4];
    const double ic = 1.0 / b.count;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double s1 = b.stats[c0 + k], s2 = b.stats[b.C + c0 + k];
        for (int sl = 1; sl < b.nslot; ++sl) { s1 += b.stats[sl * 2 * b.C + c0 + k]; s2 += b.stats[sl * 2 * b.C + b.C + c0 + k]; }
        bn_affine_finish(s1, s2, ic, b.gamma[c0 + k], b.beta[c0 + k], b.eps, s[k], t[k]);
    }
    sc = make_float4(s[0], s[1], s[2], s[3]);
    sh = make_float4(t[0], t[1], t[2], t[3]);
}

// The same in two halves, for kernels that want the float64 sums in flight early: bn_sums4 only loads (and adds the partial
// rows), bn_affine4_from turns the sums into (scale, shift).
struct BnSums4 { double s1[4], s2[4]; float g[4], b[4]; };
__device__ __forceinline__ BnSums4 bn_sums4(const BnSrc& b, int c0) {
    BnSums4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double s1 = b.stats[c0 + k], s2 = b.stats[b.C + c0 + k];
        for (int sl = 1; sl < b.nslot; ++sl) { s1 += b.stats[sl * 2 * b.C + c0 + k]; s2 += b.stats[sl * 2 * b.C + b.C + c0 + k]; }
        r.s1[k] = s1; r.s2[k] = s2; r.g[k] = b.gamma[c0 + k]; r.b[k] = b.beta[c0 + k];
    }
    return r;
}
__device__ __forceinline__ void bn_affine4_from(const BnSrc& b, const BnSums4& r, float4& sc, float4& sh) {
    float s[4], t[4];
    const double ic = 1.0 / b.count;
#pragma unroll
    for (int k = 0; k < 4; ++k) bn_affine_finish(r.s1[k], r.s2[k], ic, r.g[k], r.b[k], b.eps, s[k], t[k]);
    sc = make_float4(s[0], s[1], s[2], s[3]);
    sh = make_float4(t[0], t[1], t[2], t[3]);
}

struct ConvArgs {
    const float* x; const float* xs; const float* xb;      // input + producer BN affine (or null)
    const float* x2; const float* x2s; const float* x2b;   // optional additive skip input
    const float* w;                                         // TensorFlow kernel layout
    float* y;                                               // raw (pre-BN) output
    double* stats;                                          // (2,CoutTotal) float64 sums or null
    int D, H, W, cout_total, planes_per_wg;
    int pd, ph, pw;                                         // SAME pad_before per axis (stride 2)
    BnSrc bn, bn2;                                          // alternative to xs/xb, x2s/x2b (stats given)
    const float* wprep;                                     // weights already in the kernel's LDS order, or null
    const unsigned short* wprep_bf;                         // bf16 hi|lo split weights (opt-in bf16x3 path), or null
    int stats_slots;                                        // > 1: stats is (slots, 2, CoutTotal) partial rows, see conv_stats_row
};

// BatchNorm sums go to memory-side float64 atomics.  With every workgroup of a layer adding into the same 2*C doubles the
// atomics cost ~5 us at the end of each 240..960-workgroup layer (measured by leaving them out: the low-resolution chain
// went 140 -> 120 us; they are native global_atomic_add_f64, one instruction of 32..128 lanes per workgroup).  A layer's sums
// can be spread over `stats_slots` partial rows, workgroup w adds into row w % slots, and the consumer's bn_affine4 /
// bn_sums4 add the rows up (BnSrc::nslot); regnet.hip uses 2 rows (more rows cost the consumers more than they save).
__device__ __forceinline__ double* conv_stats_row(const ConvArgs& a) {
    const int slots = a.stats_slots > 1 ? a.stats_slots : 1;
    return a.stats + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % slots) * 2 * a.cout_total;
}

constexpr int CONV_TW = 16;      // voxels per MFMA column tile (along w)

// floats per staged position: channel count + 8 so that (bytes/16) = 2 mod 4, which makes the
// 16-lane ds_read_b128 groups of a (col = lane&15, k-quad = lane>>4) access hit 16 distinct slots
template <int CIN> struct SlabGeom { static constexpr int S = CIN + 8; };

__device__ __forceinline__ float4 bn_relu4(float4 v, float4 s, float4 b, bool aff) {
    if (aff) {
        v.x = relu(v.x * s.x + b.x); v.y = relu(v.y * s.y + b.y);
        v.z = relu(v.z * s.z + b.z); v.w = relu(v.w * s.w + b.w);
    }
    return v;
}

// Final reduction of per-lane BatchNorm partial sums.  Lane (kq = lane>>4, n = lane&15) holds the
// sums of channels 4*cq .. 4*cq+3 (cq = channel-quad index given by the caller) over its voxels.
// `red` is >= 4*2*16 floats of LDS that is dead by now; one f64 atomic per channel per workgroup.
template <int COUT>
__device__ __forceinline__ void stats_commit(const float (&st_s)[4], const float (&st_q)[4],
                                             bool fold32, float* red_raw, double* stats,
                                             int cout_total, int co_base) {
    float (*red)[2][16] = reinterpret_cast<float (*)[2][16]>(red_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    float s[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float sv = st_s[k], qv = st_q[k];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
        if (fold32) { sv += __shfl_xor(sv, 32, 64); qv += __shfl_xor(qv, 32, 64); }
        s[k] = sv; q[k] = qv;
    }
    constexpr int ngrp = COUT / 4;
    if (n == 0 && kq < ngrp) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[wave][0][4 * kq + k] = s[k]; red[wave][1][4 * kq + k] = q[k]; }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
        int k = tid / COUT, c = tid - k * COUT;
        double t = (double)red[0][k][c] + (double)red[1][k][c] + (double)red[2][k][c] + (double)red[3][k][c];
        atomicAdd(&stats[(size_t)k * cout_total + co_base + c], t);
    }
}

// Planes per workgroup.  The MFMA kernels are compute-bound, so a CU's throughput does not grow with
// the number of resident workgroups: time ~ ceil(workgroups / 256 CUs) * (planes + halo planes + a
// start-up cost of about one plane for the weight upload).  Exhaustive over the chunk length.
// `slots` > 256 is for latency-bound kernels whose co-resident workgroups do overlap.
static inline int conv_pick_planes(int D, long long wgs_per_chunk, int halo, int slots = 256) {
    int best = D;
    long long best_cost = 1LL << 60;
    for (int dr = 1; dr <= D; ++dr) {
        long long chunks = (D + dr - 1) / dr;
        long long wgs = wgs_per_chunk * chunks;
        long long rounds = (wgs + slots - 1) / slots;
        long long cost = rounds * (dr + halo + 1);
        if (cost < best_cost) { best_cost = cost; best = dr; }
    }
    return best;
}

// Output channels per workgroup chosen by the dispatchers; the weight pre-layout uses the same rule.
// kind: 0 = conv stride 1, 1 = conv stride 2, 2 = transposed conv.  0 = shape not covered by MFMA.
static inline int conv_coutg(int kind, int Cin, int Cout) {
    if (kind == 0) {
        if (Cin == 32 && Cout == 8) return 8;
        if (Cin == 16 && Cout % 16 == 0) return 16;
        if (Cin == 32 && Cout % 16 == 0) return 16;
        if (Cin == 64 && Cout % 8 == 0) return 8;
        if (Cin == 16 && Cout == 8) return 8;
        return 0;
    }
    if (kind == 1) return (Cout % 16 == 0 && (Cin == 32 || Cin == 16)) ? 16 : 0;
    if (Cin == 16 && Cout == 8) return 8;
    if (Cin == 32 && Cout % 16 == 0) return 16;
    if (Cin == 64 && Cout % 8 == 0) return 8;
    if (Cin == 16 && Cout % 16 == 0) return 16;
    return 0;
}

// Copies a workgroup's pre-laid-out weights (mvs_regnet_prepare_f32) into LDS: coalesced 16-B lanes,
// eight loads in flight per thread (a plain load -> store loop serialises on the global-load latency:
// 14 round trips for a 55 KB weight set, ~10 us at the head of every workgroup).
__device__ __forceinline__ void copy_weights_to_lds(float* wl, const float* src, int w_floats) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(wl);
    const int n4 = w_floats / 4;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * 256) {
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = i0 + 256 * k; t[k] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = i0 + 256 * k; if (i < n4) d4[i] = t[k]; }
    }
}
__device__ __forceinline__ void load_prepared_weights(float* wl, const float* wprep, int w_floats) {
    copy_weights_to_lds(wl, wprep + (size_t)blockIdx.y * w_floats, w_floats);
}

// launchers implemented in the kernel files; MVS_E_SHAPE when the shape is outside their tiling
constexpr int MVS_BN_SLOTS_MAX = 8;      // partial rows per BatchNorm layer in the regulariser's workspace
int mvs_conv3d_s2_mfma(const ConvArgs& a, int Cin, int Cout, hipStream_t st);
int mvs_deconv3d_mfma_launch(const ConvArgs& a, int Cin, int Cout, hipStream_t st);
int mvs_conv3d_out_launch(const ConvArgs& a, int Cin, hipStream_t st);
// one-channel side layers of the training path (conv3d_c1.hip)
int mvs_conv3d_in1_launch(const ConvArgs& a, int Cout, hipStream_t st);
int mvs_conv3d_k8_launch(const ConvArgs& a, int Cout, hipStream_t st);       // 8 -> 32, taps paired along w (conv3d_k8.hip)
int mvs_wgrad_c1_plan(int D, int H, int W, int CB, int* rows, int* planes_per_wg);
int mvs_wgrad_c1_launch(const float* big, const float* small, int D, int H, int W, int CB, float* partial, hipStream_t st);
int mvs_deconv3d_c8_launch(const ConvArgs& a, int Cin, int Cout, hipStream_t st);   // 8 couts per workgroup, packed tiles
int mvs_conv3d_c8_launch(const ConvArgs& a, hipStream_t st);      // 32 -> 8 stride 1 (conv3d_c8.hip)
// same + the 32 -> 16 stride-2 consumer of the same input in one pass
// slots1 / slots2 > 1: the BatchNorm sums go to (slots, 2, C) partial rows (workgroup id modulo slots), see BnSrc::nslot
int mvs_conv3d_c8_s2_launch(const ConvArgs& a, const float* w2, float* y2, double* stats2, hipStream_t st,
                            int slots1 = 1, int slots2 = 1);
// output-stationary block kernels of the low-resolution levels (conv3d_os.hip); kind: 0 stride 1, 1 stride 2, 2 transposed
bool mvs_conv3d_os_covers(int kind, int Cin, int Cout);
int mvs_conv3d_os_weight_layout(const float* w, int kind, int Cin, int Cout, float* out, hipStream_t st);
int mvs_conv3d_os_launch(const ConvArgs& a, int kind, int Cin, int Cout, hipStream_t st);
// a chain layer of the 1/8 level (3dconv3_0 / 3_1 / 4_0) + blocks [b_first, b_first + b_count) of the 32 -> 32 stride-1 layer b
// (3dconv2_1) in one launch; mvs_conv3d_os_filler_blocks = how many blocks that layer has at (D, H, W)
int mvs_conv3d_os_filler_blocks(int D, int H, int W);
int mvs_conv3d_os_filled_launch(const ConvArgs& a, int kind, int Cin, int Cout, const ConvArgs& b, int b_first, int b_count,
                                hipStream_t st);
// opt-in split-precision stride-1 path (conv3d_bf16x3.hip)
bool mvs_conv3d_bf16x3_supported(int Cin, int Cout);
int mvs_conv3d_s1_bf16x3(const ConvArgs& a, int Cin, int Cout, hipStream_t st);
int mvs_conv_weight_split(const float* w, int Cin, int Cout, unsigned short* out, hipStream_t st);
