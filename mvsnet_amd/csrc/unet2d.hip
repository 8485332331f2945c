// 2D convolutions of the UNetDS2GN feature extractor (mvsnet/cnn_wrapper/
// mvsnetworks.py:53-115, Network.conv_gn / deconv_gn network.py:217-276,350-409) as fp32-MFMA
// implicit GEMMs with the producer's GroupNorm (+ReLU) folded into the consumer's load -- the same
// fusion the 3D stack uses for BatchNorm:
//
//   y_raw = conv(k x k, stride s, SAME, no bias)( concat_c[ act1(gn1(x1)), act2(gn2(x2)) ] )
//   stats[v][g] += (sum, sum of squares) of y_raw over the 8-channel group g of view v
//
// A consumer turns a producer's raw group sums into per-channel (scale, shift) itself, so a layer is
// ONE launch and activations are stored once, raw.  Views are independent samples (GroupNorm is per
// sample), so the N towers of mvsnet/model.py:392-406 run as one batched launch.
//
// GEMM roles as in conv3d_mfma.hip: rows = 16*MT output channels, columns = 16 output pixels along w,
// K = (kh, kw, ci) walked in chunks of CK input channels (the staged patch and the chunk's weights
// live in LDS; v_mfma_f32_16x16x4_f32, exact fp32).  CG = channels per operand read: 16 (ds_read_b128,
// 4 MFMAs), 8 (b64, 2 MFMAs) or 4 (b32, 1 MFMA) for the 3/4-, 8- and >=16-channel layers.
// Transposed convolutions (4 small layers) are a VALU gather kernel.
#include "unet2d_common.h"
#include <type_traits>

namespace {

constexpr int TH = 8, TW = 16;
constexpr int NSLOT = GN_NSLOT;

__device__ __forceinline__ void gn_affine4(const GnSrc& s, int view, int c0, float4& sc, float4& sh) {
    sc = make_float4(1.f, 1.f, 1.f, 1.f); sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!s.stats) return;
    const double* st = s.stats + ((size_t)view * (s.C / 8) + c0 / 8) * (NSLOT * 2);   // 4 channels share a group
    double sum = 0.0, sq = 0.0;
#pragma unroll 8
    for (int i = 0; i < NSLOT; ++i) { sum += st[2 * i]; sq += st[2 * i + 1]; }
    const double mean = sum / s.count;
    double var = sq / s.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + 1e-5);                                   // network.py:55,254
    float a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = (float)((double)s.gamma[c0 + k] * inv);
        b[k] = (float)((double)s.beta[c0 + k] - mean * (double)s.gamma[c0 + k] * inv);
    }
    sc = make_float4(a[0], a[1], a[2], a[3]); sh = make_float4(b[0], b[1], b[2], b[3]);
}

// DECONV: the 3x3 stride-2 transposed convolution as a 2x2-tap convolution over the coarse input with
// the four output parity classes stacked as 4*Cout "channels" (class (ph,pw) of coarse pixel (m,n) is
// output pixel (2m+ph, 2n+pw); tap offsets 0 / -1 carry kernel index 0|1 / 2 per axis, absent
// combinations have zero weights), stored depth-to-space.  KS = 2, STRIDE = 1, pad 1 in front.
template <int KS, int STRIDE, int CG, int MT, bool DECONV = false>
__global__ void __launch_bounds__(256)
conv2d_gn_kernel(Conv2dArgs p) {
    constexpr int CK = CG;                          // channels per chunk
    constexpr int CQ = CK / 4;
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int NPOS = IH * IW;
    constexpr int S = (CG == 16) ? 24 : (CG == 8 ? 10 : 5);       // slab pitch (floats): conflict-free operand reads
    constexpr int COUT_T = 16 * MT;
    constexpr int WCH = KS * KS * CQ * COUT_T * 4;  // weight floats per chunk
    constexpr int V2 = 2;                           // column tiles (output rows) per wave
    extern __shared__ __attribute__((aligned(16))) float smem2d[];
    float* slab = smem2d;                           // [NPOS][S]
    float* wl = smem2d + ((NPOS * S + 3) & ~3);     // [KS*KS][CQ][COUT_T][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int tiles_w = (p.Wo + TW - 1) / TW, tiles_h = (p.Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int view = bid / (tiles_h * tiles_w); bid -= view * tiles_h * tiles_w;
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int oh0 = tile_h * TH, ow0 = tile_w * TW;
    const int ih0 = oh0 * STRIDE - p.pad_h, iw0 = ow0 * STRIDE - p.pad_w;
    const int cog = blockIdx.y;
    const int Ctot = p.a.C + (p.b.x ? p.b.C : 0);
    const int nchunks = Ctot / CK;

    f32x4 acc[MT][V2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < V2; ++v) acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int b_off[V2];
#pragma unroll
    for (int v = 0; v < V2; ++v) b_off[v] = ((V2 * wave + v) * STRIDE * IW + n * STRIDE) * S + (CG / 4) * kq;
    const int a_off = (kq * COUT_T + n) * 4;

    // GroupNorm (scale, shift) of every input channel of this view -> LDS, once per workgroup.  The
    // NSLOT partial sums of a group are added up by 8 threads (4 slots each, all loads in flight at
    // once, then a 3-step shuffle): a few threads walking 64 doubles serially cost ~7 us per workgroup.
    __shared__ __attribute__((aligned(16))) float aff_s[256], aff_b[256];
    __shared__ double g_mean[32], g_inv[32];
    {
        const int g = tid >> 3, part = tid & 7;             // up to 32 groups of 8 channels in [a | b]
        const int ga = p.a.C / 8;
        const bool in_a = g < ga;
        const GnSrc& src = in_a ? p.a : p.b;
        const int gl = in_a ? g : g - ga;
        double sum = 0.0, sq = 0.0;
        const bool live = g < Ctot / 8 && src.stats != nullptr;
        if (live) {
            const double* st = src.stats + (((size_t)view * (src.C / 8) + gl) * NSLOT + part * (NSLOT / 8)) * 2;
#pragma unroll
            for (int i = 0; i < NSLOT / 8; ++i) { sum += st[2 * i]; sq += st[2 * i + 1]; }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
        if (part == 0 && g < 32) {
            double mean = 0.0, inv = 1.0;                    // identity when the source has no GroupNorm
            if (live) {
                mean = sum / src.count;
                double var = sq / src.count - mean * mean;
                if (var < 0.0) var = 0.0;
                inv = 1.0 / sqrt(var + 1e-5);               // network.py:55,254
            }
            g_mean[g] = mean; g_inv[g] = inv;
        }
        __syncthreads();
        if (tid < Ctot / 4) {
            const int c = 4 * tid;
            const bool ca = c < p.a.C;
            const GnSrc& s2 = ca ? p.a : p.b;
            const int cl = ca ? c : c - p.a.C;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
            if (s2.stats) {
                const double mean = g_mean[c / 8], inv = g_inv[c / 8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double a = (double)s2.gamma[cl + k] * inv;
                    sc[k] = (float)a; sh[k] = (float)((double)s2.beta[cl + k] - mean * a);
                }
            }
            *(float4*)(aff_s + c) = make_float4(sc[0], sc[1], sc[2], sc[3]);
            *(float4*)(aff_b + c) = make_float4(sh[0], sh[1], sh[2], sh[3]);
        }
    }

    // Chunk ch+1's patch and weights are requested from global memory (into registers) before the
    // MFMAs of chunk ch and written to LDS after them: the load latency hides under the matrix work.
    constexpr int NIN = (NPOS * CQ + 255) / 256, NWT = (WCH / 4 + 255) / 256;
    float4 pin[NIN], pwt[NWT];
    const int q = tid % CQ;                         // 256 % CQ == 0: a thread keeps one channel quad
    auto fetch = [&](int ch) __attribute__((always_inline)) {
        const int c0 = ch * CK;
        const bool from_b = c0 >= p.a.C;
        const GnSrc& src = from_b ? p.b : p.a;
        const int cs = from_b ? c0 - p.a.C : c0;    // first channel inside its source
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int f = tid + 256 * i;
            const int pos = f / CQ;
            const int r = pos / IW, c = pos - r * IW;
            const int gh = ih0 + r, gw = iw0 + c;
            const bool ok = f < NPOS * CQ && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
            pin[i] = ok ? *(const float4*)(src.x + (((size_t)view * p.H + gh) * p.W + gw) * src.C + cs + 4 * q)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4* w4 = reinterpret_cast<const float4*>(p.wprep + ((size_t)cog * nchunks + ch) * WCH);
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int k = tid + 256 * i;
            pwt[i] = k < WCH / 4 ? w4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // registers -> LDS: GroupNorm affine (+ReLU) on the way, zeros outside the image (SAME padding)
    auto stage = [&](int ch) __attribute__((always_inline)) {
        const int c0 = ch * CK;
        const bool relu_on = (c0 >= p.a.C ? p.b.relu : p.a.relu) != 0;
        const float4 sc = *(const float4*)(aff_s + c0 + 4 * q), sh = *(const float4*)(aff_b + c0 + 4 * q);
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int f = tid + 256 * i;
            if (f >= NPOS * CQ) break;
            const int pos = f / CQ;
            const int r = pos / IW, c = pos - r * IW;
            const int gh = ih0 + r, gw = iw0 + c;
            float4 v = pin[i];
            if (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) {
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (relu_on) { v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w); }
            }
            if (CG == 4) { float* d = slab + pos * S; d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
            else if (CG == 8) { *(f32x2*)(slab + pos * S + 4 * q) = (f32x2){v.x, v.y}; *(f32x2*)(slab + pos * S + 4 * q + 2) = (f32x2){v.z, v.w}; }
            else *(float4*)(slab + pos * S + 4 * q) = v;
        }
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int k = tid + 256 * i;
            if (k < WCH / 4) reinterpret_cast<float4*>(wl)[k] = pwt[i];
        }
    };

    fetch(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                            // previous chunk's operands are dead (first trip: affine table written)
        stage(ch);
        __syncthreads();
        if (ch + 1 < nchunks) fetch(ch + 1);
        // ---- MFMAs: all taps of the chunk
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int tap = kh * KS + kw;
                float bq[V2][4], aq[MT][4];
#pragma unroll
                for (int v = 0; v < V2; ++v) {
                    const float* bp = slab + b_off[v] + (kh * IW + kw) * S;
                    if (CG == 16) { f32x4 t = *(const f32x4*)bp; bq[v][0] = t[0]; bq[v][1] = t[1]; bq[v][2] = t[2]; bq[v][3] = t[3]; }
                    else if (CG == 8) { f32x2 t = *(const f32x2*)bp; bq[v][0] = t[0]; bq[v][1] = t[1]; }
                    else bq[v][0] = *bp;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    // A rows: [tap][ciq][co][4]; this lane supplies ci = (CG/4)*kq + j of the chunk
                    if (CG == 16) { f32x4 t = *(const f32x4*)(wl + a_off + m * 64 + tap * CQ * COUT_T * 4); aq[m][0] = t[0]; aq[m][1] = t[1]; aq[m][2] = t[2]; aq[m][3] = t[3]; }
                    else if (CG == 8) {
                        // ci = 2kq + j -> ci-quad kq>>1, element 2(kq&1) + j
                        f32x2 t = *(const f32x2*)(wl + (((tap * CQ + (kq >> 1)) * COUT_T + m * 16 + n) * 4 + 2 * (kq & 1)));
                        aq[m][0] = t[0]; aq[m][1] = t[1];
                    } else aq[m][0] = wl[((tap * COUT_T) + m * 16 + n) * 4 + kq];
                }
#pragma unroll
                for (int j = 0; j < CG / 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int v = 0; v < V2; ++v)
                            acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m][j], bq[v][j], acc[m][v], 0, 0, 0);
            }
        }
    }

    // ---- store raw outputs, GroupNorm sums per 8-channel group ----------------------------------------
    float gs[MT] , gq[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { gs[m] = 0.f; gq[m] = 0.f; }
    const int co_base = cog * COUT_T;
    const int cout_rows = DECONV ? 4 * p.Cout : p.Cout;     // GEMM rows in total
#pragma unroll
    for (int v = 0; v < V2; ++v) {
        const int oh = oh0 + V2 * wave + v, ow = ow0 + n;
        if (oh < p.Ho && ow < p.Wo) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co = co_base + m * 16 + 4 * kq;
                if (co < cout_rows) {
                    f32x4 r = acc[m][v];
                    float* dst;
                    if (DECONV) {
                        const int cls = co / p.Cout, c = co - cls * p.Cout;
                        dst = p.y + (((size_t)view * 2 * p.Ho + 2 * oh + (cls >> 1)) * (2 * p.Wo) + 2 * ow + (cls & 1)) * p.Cout + c;
                    } else dst = p.y + (((size_t)view * p.Ho + oh) * p.Wo + ow) * p.Cout + co;
                    *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
                    gs[m] += (r[0] + r[1]) + (r[2] + r[3]);
                    gq[m] += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                }
            }
        }
    }
    if (p.stats) {
        // lane (kq, n) holds channels 4kq..4kq+3 of tile m: group 2m + (kq >> 1); fold n and the kq pair
        __shared__ float red[4][MT][2][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s = gs[m], q = gq[m];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
            if (n == 0 && (kq & 1) == 0) { red[wave][m][kq >> 1][0] = s; red[wave][m][kq >> 1][1] = q; }
        }
        __syncthreads();
        if (tid < MT * 4) {
            const int m = tid >> 2, h = (tid >> 1) & 1, k = tid & 1;
            const int row8 = co_base + m * 16 + 8 * h;                   // first GEMM row of this 8-channel half
            const int g = (DECONV ? row8 % p.Cout : row8) / 8;
            if (row8 < cout_rows) {
                double t = (double)red[0][m][h][k] + (double)red[1][m][h][k] + (double)red[2][m][h][k] + (double)red[3][m][h][k];
                atomicAdd(&p.stats[(((size_t)view * (p.Cout / 8) + g) * NSLOT + (blockIdx.x & (NSLOT - 1))) * 2 + k], t);
            }
        }
    }
}

// TensorFlow conv2d kernel (k,k,Cin,Cout) -> [cout group][chunk][tap][CK/4][COUT_T][4], zero padded
// flipT: `w` is the FORWARD kernel (k,k,Cout,Cin) of the layer whose input gradient this convolution computes -- the tap is
// mirrored and the channel roles swapped while reading (what w.flip(0,1).permute(0,1,3,2).contiguous() materialised before)
// (`CinSrc`: channels `w` really has -- the image layer's kernel (k,k,3,Cout) is laid out for 4 channels, the fourth zero)
__device__ __forceinline__ float conv2d_layout_value(const float* __restrict__ w, int KS, int Cin, int CinSrc, int Cout, int CK,
                                                     int COUT_T, int CinPad, int flipT, long long i) {
    const int nch = CinPad / CK, CQ = CK / 4;
    long long r = i;
    const int j = r & 3; r >>= 2;
    const int co = r % COUT_T; r /= COUT_T;
    const int ciq = r % CQ; r /= CQ;
    const int tap = r % (KS * KS); r /= (KS * KS);
    const int ch = r % nch; const int g = r / nch;
    const int ci = ch * CK + ciq * 4 + j, cout = g * COUT_T + co;
    if (!(ci < Cin && ci < CinSrc && cout < Cout)) return 0.f;
    return flipT ? w[((size_t)(KS * KS - 1 - tap) * Cout + cout) * CinSrc + ci] : w[((size_t)tap * CinSrc + ci) * Cout + cout];
}

__global__ void conv2d_weight_layout_kernel(const float* __restrict__ w, int KS, int Cin, int Cout, int CK,
                                            int COUT_T, int CinPad, float* __restrict__ out, int flipT) {
    const int nch = CinPad / CK, CQ = CK / 4, groups = (Cout + COUT_T - 1) / COUT_T;
    const long long total = (long long)groups * nch * KS * KS * CQ * COUT_T * 4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = conv2d_layout_value(w, KS, Cin, Cin, Cout, CK, COUT_T, CinPad, flipT, i);
}

// TensorFlow conv2d_transpose kernel (3,3,Cout,Cin) -> the stacked 2x2-tap form described at
// conv2d_gn_kernel: [group][chunk][tap 2x2][CK/4][COUT_T][4], rows = (class, cout)
__device__ __forceinline__ float deconv2d_layout_value(const float* __restrict__ w, int Cin, int Cout, int CK, int COUT_T, long long i) {
    const int nch = Cin / CK, CQ = CK / 4, rows = 4 * Cout;
    long long r = i;
    const int j = r & 3; r >>= 2;
    const int co = r % COUT_T; r /= COUT_T;
    const int ciq = r % CQ; r /= CQ;
    const int tap = r % 4; r /= 4;
    const int ch = r % nch; const int g = r / nch;
    const int ci = ch * CK + ciq * 4 + j, row = g * COUT_T + co;
    float v = 0.f;
    if (row < rows) {
        const int cls = row / Cout, c = row - cls * Cout, ph = cls >> 1, pw = cls & 1;
        const int dh = 1 - (tap >> 1), dw = 1 - (tap & 1);        // staged tap 0 is the (-1) neighbour
        // class parity 0: offset 0 -> k = 0, offset -1 -> k = 2;  parity 1: offset 0 -> k = 1, offset -1 -> none
        const int kh = ph ? (dh ? -1 : 1) : (dh ? 2 : 0), kw = pw ? (dw ? -1 : 1) : (dw ? 2 : 0);
        if (kh >= 0 && kw >= 0) v = w[((size_t)(kh * 3 + kw) * Cout + c) * Cin + ci];
    }
    return v;
}

__global__ void deconv2d_weight_layout_kernel(const float* __restrict__ w, int Cin, int Cout, int CK, int COUT_T,
                                              float* __restrict__ out) {
    const int nch = Cin / CK, CQ = CK / 4, rows = 4 * Cout, groups = (rows + COUT_T - 1) / COUT_T;
    const long long total = (long long)groups * nch * 4 * CQ * COUT_T * 4;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    out[i] = deconv2d_layout_value(w, Cin, Cout, CK, COUT_T, i);
}

// Many kernels in one launch (the training towers lay out every layer's forward AND input-gradient kernel once per step: ~60
// launches of a step the launching thread bounds); the jobs ride in the kernel arguments, blockIdx.y = job.
constexpr int PREP_MAX = 56;
struct PrepJob { const float* w; float* out; long long total; int deconv, KS, Cin, CinSrc, Cout, CK, COUT_T, CinPad, flipT; };
struct PrepJobs { PrepJob j[PREP_MAX]; };

__global__ __launch_bounds__(256) void weight_layout_many_kernel(PrepJobs jobs) {
    const PrepJob J = jobs.j[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < J.total; i += (long long)gridDim.x * 256)
        J.out[i] = J.deconv ? deconv2d_layout_value(J.w, J.Cin, J.Cout, J.CK, J.COUT_T, i)
                            : conv2d_layout_value(J.w, J.KS, J.Cin, J.CinSrc, J.Cout, J.CK, J.COUT_T, J.CinPad, J.flipT, i);
}

// ---- transposed convolution k3 s2 SAME (out = 2n, cropped at the end), bias-free, VALU gather --------------
// (reference implementation of the kernel above; used for shapes outside the MFMA tiling)
// out[2i + k] += in[i] * W[k][co][ci] per axis (weight layout (3,3,Cout,Cin), TF conv2d_transpose).
struct Deconv2dArgs {
    GnSrc a;
    const float* w;           // (3,3,Cout,Cin)
    float* y;                 // (V,2H,2W,Cout) raw
    double* stats;            // (V, Cout/8, 2) or null
    int V, H, W, Cout;
};

__global__ void __launch_bounds__(256)
deconv2d_gn_kernel(Deconv2dArgs p) {
    const int Ho = 2 * p.H, Wo = 2 * p.W, CQo = p.Cout / 4, Cin = p.a.C;
    const long long total = (long long)Ho * Wo * CQo;            // items of one view (grid.y = view)
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool live = idx < total;
    int coq = 0; const int view = blockIdx.y; long long pix = 0;
    // GroupNorm affine of all input channels of this view, once per block
    __shared__ float aff_s[256], aff_b[256];
    {
        if ((int)threadIdx.x < Cin / 4) {
            float4 sc, sh;
            gn_affine4(p.a, view, 4 * threadIdx.x, sc, sh);
            *(float4*)(aff_s + 4 * threadIdx.x) = sc; *(float4*)(aff_b + 4 * threadIdx.x) = sh;
        }
        __syncthreads();
    }
    if (live) {
        coq = (int)(idx % CQo); pix = idx / CQo;
        const int ow = (int)(pix % Wo); const int oh = (int)(pix / Wo);
        pix += (long long)view * Ho * Wo;
        // taps: even o = 2m: (i = m, k = 0), (i = m-1, k = 2); odd o = 2m+1: (i = m, k = 1)
        int ih[2], kh[2], nh = 0, iw[2], kw[2], nw = 0;
        if (oh & 1) { ih[0] = oh >> 1; kh[0] = 1; nh = 1; } else { ih[0] = oh >> 1; kh[0] = 0; nh = 1; if (oh >= 2) { ih[1] = (oh >> 1) - 1; kh[1] = 2; nh = 2; } }
        if (ow & 1) { iw[0] = ow >> 1; kw[0] = 1; nw = 1; } else { iw[0] = ow >> 1; kw[0] = 0; nw = 1; if (ow >= 2) { iw[1] = (ow >> 1) - 1; kw[1] = 2; nw = 2; } }
        for (int a = 0; a < nh; ++a)
            for (int b = 0; b < nw; ++b) {
                const float* xin = p.a.x + (((size_t)view * p.H + ih[a]) * p.W + iw[b]) * Cin;
                const float* wt = p.w + ((size_t)(kh[a] * 3 + kw[b]) * p.Cout + 4 * coq) * Cin;
                for (int c = 0; c < Cin; c += 4) {
                    const float4 sc = *(const float4*)(aff_s + c), sh = *(const float4*)(aff_b + c);
                    float4 v = *(const float4*)(xin + c);
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    if (p.a.relu) { v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w); }
                    const float4 w0 = *(const float4*)(wt + c), w1 = *(const float4*)(wt + Cin + c);
                    const float4 w2 = *(const float4*)(wt + 2 * Cin + c), w3 = *(const float4*)(wt + 3 * Cin + c);
                    acc.x += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
                    acc.y += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
                    acc.z += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
                    acc.w += v.x * w3.x + v.y * w3.y + v.z * w3.z + v.w * w3.w;
                }
            }
        *(float4*)(p.y + pix * p.Cout + 4 * coq) = acc;
    }
    if (p.stats) {
        // a wave covers consecutive (pixel, quad) items of one view, lane % CQo = quad;
        // quads q and q^1 form a group of 8 channels: lanes l and l^1
        float s = live ? (acc.x + acc.y) + (acc.z + acc.w) : 0.f;
        float q = live ? (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w) : 0.f;
        s += __shfl_xor(s, 1, 64); q += __shfl_xor(q, 1, 64);
        // fold the lanes that share a group: same (lane mod CQo) >> 1; CQo is a power of two <= 32
        for (int o = CQo; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        const int lane = threadIdx.x & 63;
        if (live && lane < CQo && (lane & 1) == 0) {
            const int g = (coq >> 1);
            double* dst = p.stats + (((size_t)view * (p.Cout / 8) + g) * NSLOT + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & (NSLOT - 1))) * 2;
            atomicAdd(dst, (double)s);
            atomicAdd(dst + 1, (double)q);
        }
    }
}

template <int KS, int STRIDE, int CG, int MT, bool DECONV = false>
int launch_conv2d(const Conv2dArgs& p, hipStream_t st) {
    const int tiles = ((p.Ho + TH - 1) / TH) * ((p.Wo + TW - 1) / TW);
    dim3 grid(p.V * tiles, ((DECONV ? 4 : 1) * p.Cout + 16 * MT - 1) / (16 * MT));
    constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    constexpr int S = (CG == 16) ? 24 : (CG == 8 ? 10 : 5);
    constexpr size_t smem = (size_t)(((IH * IW * S + 3) & ~3) + KS * KS * CG * 16 * MT) * sizeof(float);
    static bool attr_done = false;       // per template instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_gn_kernel<KS, STRIDE, CG, MT, DECONV>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv2d_gn_kernel<KS, STRIDE, CG, MT, DECONV><<<grid, 256, smem, st>>>(p);
    return (int)hipGetLastError();
}

// operand-read width and row tiles per workgroup chosen from the layer shape (the weight pre-layout uses the same rule)
void conv2d_tiling(int Cin_total, int Cout, int c1, int& CG, int& MT) {
    CG = (Cin_total % 16 == 0 && c1 % 16 == 0) ? 16 : (Cin_total % 8 == 0 && c1 % 8 == 0 ? 8 : 4);
    MT = (Cout >= 32) ? 2 : 1;          // (64 couts per workgroup measured slower: too few workgroups on the low-resolution layers)
}

}  // namespace

extern "C" int mvs_gn_stat_slots(void) { return NSLOT; }

static size_t conv2d_plain_floats(int ks, int cin1, int cin2, int cout) {
    int CG, MT; conv2d_tiling(cin1 + cin2, cout, cin1, CG, MT);
    const int cin = cin1 + cin2, cpad = (cin + CG - 1) / CG * CG, ct = 16 * MT;
    return (size_t)((cout + ct - 1) / ct) * (cpad / CG) * ks * ks * (CG / 4) * ct * 4;
}

extern "C" size_t mvs_conv2d_prepared_floats(int ks, int cin1, int cin2, int cout) {
    int CG, MT; conv2d_tiling(cin1 + cin2, cout, cin1, CG, MT);
    const int cin = cin1 + cin2, cpad = (cin + CG - 1) / CG * CG, ct = 16 * MT;
    return conv2d_plain_floats(ks, cin1, cin2, cout) + ((cin1 % CG) || (cin2 % CG) ? 0 : mvs_conv2d_pair_floats(ks, 1, cin, cout));
}

extern "C" int mvs_conv2d_prepare_f32(const float* w, int ks, int cin1, int cin2, int cout, float* prepared, void* stream) {
    MVS_CHECK_ARG(w && prepared && (ks == 3 || ks == 5) && cin1 > 0 && cin2 >= 0 && cout > 0);
    int CG, MT; conv2d_tiling(cin1 + cin2, cout, cin1, CG, MT);
    const int cin = cin1 + cin2, cpad = (cin + CG - 1) / CG * CG;
    const size_t total = conv2d_plain_floats(ks, cin1, cin2, cout);
    conv2d_weight_layout_kernel<<<mvs_cdiv((long long)total, 256), 256, 0, mvs_stream(stream)>>>(w, ks, cin, cout, CG, 16 * MT, cpad, prepared, 0);
    if (mvs_conv2d_prepared_floats(ks, cin1, cin2, cout) > total)      // 8-cout 3 x 3 layers: the pixel-pair layout of the persistent kernel behind it
        return mvs_conv2d_pair_prepare(w, cin, cout, CG, prepared + total, mvs_stream(stream), 0);
    MVS_LAUNCH_RET();
}

// Prepared weights of the stride-1 convolution that computes a layer's INPUT gradient, straight from the layer's forward kernel
// w (k,k,cin_fwd,cout_fwd): the gradient convolution maps cout_fwd -> cin_fwd channels with the mirrored, transposed kernel.
// Same buffer size / consumer as mvs_conv2d_prepare_f32(k, cout_fwd, 0, cin_fwd) (training towers: one launch instead of
// flip + permute + copy + prepare).
extern "C" int mvs_conv2d_prepare_dgrad_f32(const float* w, int ks, int cin_fwd, int cout_fwd, float* prepared, void* stream) {
    MVS_CHECK_ARG(w && prepared && (ks == 3 || ks == 5) && cin_fwd > 0 && cout_fwd > 0);
    const int cin = cout_fwd, cout = cin_fwd;            // roles in the gradient convolution
    int CG, MT; conv2d_tiling(cin, cout, cin, CG, MT);
    const int cpad = (cin + CG - 1) / CG * CG;
    const size_t total = conv2d_plain_floats(ks, cin, 0, cout);
    conv2d_weight_layout_kernel<<<mvs_cdiv((long long)total, 256), 256, 0, mvs_stream(stream)>>>(w, ks, cin, cout, CG, 16 * MT, cpad, prepared, 1);
    if (mvs_conv2d_prepared_floats(ks, cin, 0, cout) > total)
        return mvs_conv2d_pair_prepare(w, cin, cout, CG, prepared + total, mvs_stream(stream), 1);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_conv2d_gn_f32(const float* x1, const double* stats1, const float* gamma1, const float* beta1, int c1, int relu1,
                                 const float* x2, const double* stats2, const float* gamma2, const float* beta2, int c2, int relu2,
                                 const float* prepared, int V, int H, int W, int cout, int ks, int stride,
                                 float* y, double* stats_out, void* stream) {
    MVS_CHECK_ARG(x1 && prepared && y && V > 0 && H > 0 && W > 0 && c1 > 0 && cout > 0 && c2 >= 0);
    MVS_CHECK_ARG((stats1 == nullptr) == (gamma1 == nullptr) && (c2 == 0) == (x2 == nullptr));
    if ((c1 % 4) || (c2 % 4) || (cout % 8) || (stats1 && c1 % 8) || (stats2 && c2 % 8) || c1 + c2 > 256) return MVS_E_SHAPE;
    int CG, MT; conv2d_tiling(c1 + c2, cout, c1, CG, MT);
    if ((c1 % CG) || (c2 % CG)) return MVS_E_SHAPE;
    Conv2dArgs p;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    auto pad_before = [&](int n, int o) { int t = (o - 1) * stride + ks - n; return t < 0 ? 0 : t / 2; };
    p.a = GnSrc{x1, stats1, gamma1, beta1, (double)H * W * 8, c1, relu1};
    p.b = GnSrc{x2, stats2, gamma2, beta2, (double)H * W * 8, c2, relu2};
    p.wprep = prepared; p.y = y; p.stats = stats_out;
    p.wpair = (stride == 1 && mvs_conv2d_prepared_floats(ks, c1, c2, cout) > conv2d_plain_floats(ks, c1, c2, cout)) ? prepared + conv2d_plain_floats(ks, c1, c2, cout) : nullptr;
    p.V = V; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.Cout = cout;
    p.pad_h = pad_before(H, Ho); p.pad_w = pad_before(W, Wo);
    hipStream_t st = mvs_stream(stream);
    // persistent form where an instance exists (unet2d_p.hip; the same prepared weights)
    if (const int inst = mvs_conv2d_p_find(ks, stride, c1 + c2, CG, MT, cout); inst >= 0) {
        const int rc = mvs_conv2d_p_run(inst, p, st);
        if (rc != MVS_E_SHAPE) return rc;
    }
#define CASE(K, S_, G, M) if (ks == K && stride == S_ && CG == G && MT == M) return launch_conv2d<K, S_, G, M>(p, st);
    CASE(3, 1, 4, 1) CASE(3, 1, 8, 1) CASE(3, 1, 16, 1) CASE(3, 1, 16, 2) CASE(3, 1, 8, 2) CASE(3, 1, 4, 2)
    CASE(3, 2, 4, 1) CASE(3, 2, 8, 1) CASE(3, 2, 16, 1) CASE(3, 2, 16, 2) CASE(3, 2, 8, 2) CASE(3, 2, 4, 2)
    CASE(5, 2, 8, 1) CASE(5, 2, 16, 1) CASE(5, 2, 16, 2) CASE(5, 2, 8, 2)
#undef CASE
    return MVS_E_SHAPE;
}

extern "C" size_t mvs_deconv2d_prepared_floats(int cin, int cout) {
    if (cin % 16 || cout % 8) return 0;
    const int ct = 32;                              // 4*cout rows in tiles of 32
    return (size_t)((4 * cout + ct - 1) / ct) * (cin / 16) * 4 * 4 * ct * 4;
}

extern "C" int mvs_deconv2d_prepare_f32(const float* w, int cin, int cout, float* prepared, void* stream) {
    MVS_CHECK_ARG(w && prepared && cin > 0 && cout > 0);
    const size_t total = mvs_deconv2d_prepared_floats(cin, cout);
    if (!total) return MVS_E_SHAPE;
    deconv2d_weight_layout_kernel<<<mvs_cdiv((long long)total, 256), 256, 0, mvs_stream(stream)>>>(w, cin, cout, 16, 32, prepared);
    MVS_LAUNCH_RET();
}

// n preparations in ONE launch (+ one small launch per 8-cout 3 x 3 layer for its pixel-pair layout).  Job i:
//   kind 0  mvs_conv2d_prepare_f32(w, ks, c1, c2, cout) -- with cin_src channels in `w` (the image layer: c1 = 4, cin_src = 3;
//           otherwise c1 + c2)
//   kind 1  mvs_conv2d_prepare_dgrad_f32(w, ks, cin_fwd = c1, cout_fwd = cout)
//   kind 2  mvs_deconv2d_prepare_f32(w, cin = c1, cout)
// into prepared[i] (sized by the matching *_prepared_floats).
extern "C" int mvs_unet_prepare_many_f32(int n, const int* kind, const float* const* w, const int* ks, const int* c1, const int* c2,
                                         const int* cin_src, const int* cout, float* const* prepared, void* stream) {
    MVS_CHECK_ARG(n > 0 && kind && w && ks && c1 && c2 && cin_src && cout && prepared);
    for (int i = 0; i < n; ++i) {
        MVS_CHECK_ARG(w[i] && prepared[i] && c1[i] > 0 && c2[i] >= 0 && cout[i] > 0 && kind[i] >= 0 && kind[i] <= 2);
        if (kind[i] != 2) MVS_CHECK_ARG(ks[i] == 3 || ks[i] == 5);
        else if (!mvs_deconv2d_prepared_floats(c1[i], cout[i])) return MVS_E_SHAPE;
        if (kind[i] == 0) MVS_CHECK_ARG(cin_src[i] > 0 && cin_src[i] <= c1[i] + c2[i]);
    }
    hipStream_t st = mvs_stream(stream);
    for (int first = 0; first < n; first += PREP_MAX) {
        const int m = n - first < PREP_MAX ? n - first : PREP_MAX;
        PrepJobs jobs;
        long long largest = 1;
        for (int k = 0; k < m; ++k) {
            const int i = first + k;
            PrepJob& J = jobs.j[k];
            J.w = w[i]; J.out = prepared[i];
            if (kind[i] == 2) {
                J.deconv = 1; J.KS = 3; J.Cin = c1[i]; J.CinSrc = c1[i]; J.Cout = cout[i]; J.CK = 16; J.COUT_T = 32; J.CinPad = c1[i]; J.flipT = 0;
                J.total = (long long)mvs_deconv2d_prepared_floats(c1[i], cout[i]);
            } else {
                const int cin = kind[i] == 1 ? cout[i] : c1[i] + c2[i], co = kind[i] == 1 ? c1[i] : cout[i];      // roles in THIS convolution
                int CG, MT; conv2d_tiling(cin, co, kind[i] == 1 ? cin : c1[i], CG, MT);
                J.deconv = 0; J.KS = ks[i]; J.Cin = cin; J.CinSrc = kind[i] == 0 ? cin_src[i] : cin; J.Cout = co; J.CK = CG; J.COUT_T = 16 * MT;
                J.CinPad = (cin + CG - 1) / CG * CG; J.flipT = kind[i] == 1;
                J.total = (long long)(kind[i] == 1 ? conv2d_plain_floats(ks[i], cin, 0, co) : conv2d_plain_floats(ks[i], c1[i], c2[i], co));
            }
            if (J.total > largest) largest = J.total;
        }
        const int gx = (int)((largest + 2047) / 2048 < 32 ? (largest + 2047) / 2048 : 32);
        hipLaunchKernelGGL(weight_layout_many_kernel, dim3(gx, m), dim3(256), 0, st, jobs);
    }
    for (int i = 0; i < n; ++i) {                          // the pixel-pair layouts behind the plain ones
        if (kind[i] == 2) continue;
        const int cin = kind[i] == 1 ? cout[i] : c1[i] + c2[i], co = kind[i] == 1 ? c1[i] : cout[i];
        const int a1 = kind[i] == 1 ? cin : c1[i], a2 = kind[i] == 1 ? 0 : c2[i];
        const size_t plain = conv2d_plain_floats(ks[i], a1, a2, co);
        if (mvs_conv2d_prepared_floats(ks[i], a1, a2, co) > plain) {
            int CG, MT; conv2d_tiling(cin, co, a1, CG, MT);
            int rc = mvs_conv2d_pair_prepare(w[i], cin, co, CG, prepared[i] + plain, st, kind[i] == 1, kind[i] == 0 ? cin_src[i] : cin);
            if (rc) return rc;
        }
    }
    MVS_LAUNCH_RET();
}

// `prepared` != NULL (from mvs_deconv2d_prepare_f32; cin a multiple of 16): MFMA path; else the VALU gather on `w`
extern "C" int mvs_deconv2d_gn_f32(const float* x, const double* stats, const float* gamma, const float* beta, int cin, int relu,
                                   const float* w, const float* prepared, int V, int H, int W, int cout, float* y,
                                   double* stats_out, void* stream) {
    MVS_CHECK_ARG(x && (w || prepared) && y && V > 0 && H > 0 && W > 0 && cin > 0 && cout > 0);
    MVS_CHECK_ARG((stats == nullptr) == (gamma == nullptr));
    if (prepared) {
        if ((cin % 16) || (cout % 8)) return MVS_E_SHAPE;
        Conv2dArgs q;
        q.a = GnSrc{x, stats, gamma, beta, (double)H * W * 8, cin, relu};
        q.b = GnSrc{nullptr, nullptr, nullptr, nullptr, 1.0, 0, 0};
        q.wprep = prepared; q.wpair = nullptr; q.y = y; q.stats = stats_out;
        q.V = V; q.H = H; q.W = W; q.Ho = H; q.Wo = W; q.Cout = cout; q.pad_h = 1; q.pad_w = 1;
        return launch_conv2d<2, 1, 16, 2, true>(q, mvs_stream(stream));
    }
    MVS_CHECK_ARG(w);
    const int cqo = cout / 4;
    if ((cin % 4) || cin > 256 || (cout % 8) || (stats && cin % 8) || cqo > 32 || (cqo & (cqo - 1))) return MVS_E_SHAPE;
    Deconv2dArgs p{GnSrc{x, stats, gamma, beta, (double)H * W * 8, cin, relu}, w, y, stats_out, V, H, W, cout};
    const long long per_view = (long long)4 * H * W * cqo;
    deconv2d_gn_kernel<<<dim3(mvs_cdiv(per_view, 256), V), 256, 0, mvs_stream(stream)>>>(p);
    MVS_LAUNCH_RET();
}
