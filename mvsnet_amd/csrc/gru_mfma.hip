// fp32-MFMA 3x3 convolution over the channel concatenation [xa | xb] for the first ConvGRU cell of the
// R-MVSNet sweep (90 % of the recurrent path's MACs): the gate convolution (48 -> 32) and the
// candidate convolution (48 -> 16) of mvsnet/convgru.py:89-93,107-111, with bias.
//
// Same GEMM roles as conv3d_mfma.hip: rows = cout, columns = 16 pixels along w, K = (kh, kw, ci),
// v_mfma_f32_16x16x4_f32, one ds_read_b128 per operand per 4 k-steps, slab positions padded to
// Cin+8 floats (conflict-free 16-lane b128 groups).  Workgroups are persistent over 8x16 pixel tiles
// (weights stay in LDS, the next tile's global loads are issued before the current tile's sweep).
// MODE 1 fuses the reset gate into the staging of xb:  xb = sigmoid(LayerNorm(g_r)) * h
// (convgru.py:97,101,107), so r*h is never materialised.  LayerNorm moments of the output are
// accumulated per row tile (tile 0 = reset | tile 1 = update for the gate convolution).
#include "conv_common.h"

namespace {

struct Gru2dArgs {
    const float* xa;            // (H,W,CA)
    const float* xb;            // (H,W,CB): h
    const float* g;             // MODE 1: raw gate conv output (H,W,2*CB); reset gate = channels [0,CB)
    const double* g_stats;      // MODE 1: (2,2) [sum,sumsq] of reset | update groups
    const float* r_gamma; const float* r_beta;     // MODE 1
    const float* wprep;         // [tap9][(CA+CB)/4][COUT][4]
    const float* bias;          // (COUT)
    float* y;                   // (H,W,COUT)
    double* stats;              // (COUT/16 groups, 2) [sum, sumsq]
    int H, W, tiles_h, tiles_w;
};

constexpr int TH2 = 8, TW2 = 16, PW2 = TW2 + 2;

template <int CA, int CB, int COUT, int MODE>
__global__ void __launch_bounds__(256, 1)
conv2d_cat_mfma_kernel(Gru2dArgs a) {
    constexpr int CT = CA + CB;
    constexpr int S = CT + 8;
    constexpr int NPOS = (TH2 + 2) * PW2;
    constexpr int CQ = CT / 4, CQA = CA / 4;
    constexpr int NF4 = NPOS * CQ;
    constexpr int STG = (256 / CQ) * CQ;            // staging threads: a thread keeps one channel quad
    constexpr int NIT = (NF4 + STG - 1) / STG;
    constexpr int MT = COUT / 16;
    constexpr int V = 2;
    constexpr int WROW = COUT * 4;
    constexpr int W_FLOATS = 9 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(CA % 16 == 0 && CB % 16 == 0 && COUT % 16 == 0, "tiling");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;
    float* slab = smem + W_FLOATS;                  // [2][NPOS][S]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int ntiles = a.tiles_h * a.tiles_w;

    {   // weights: coalesced copy of the prepared layout
        const float4* s4 = reinterpret_cast<const float4*>(a.wprep);
        for (int i = tid; i < W_FLOATS / 4; i += 256) reinterpret_cast<float4*>(wl)[i] = s4[i];
    }

    // this thread's channel quad of the concatenation; quads >= CQA come from xb
    const int c4 = tid % CQ;
    const bool from_b = c4 >= CQA;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;      // MODE 1: LayerNorm affine of the reset gate
    if (MODE == 1 && from_b) {
        const int f0 = 4 * (c4 - CQA);
        const double cnt = (double)a.H * a.W * CB;
        double mean = a.g_stats[0] / cnt;
        double var = a.g_stats[1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        float s[4], t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double inv = (double)a.r_gamma[f0 + k] / sqrt(var + 1e-12);
            s[k] = (float)inv; t[k] = (float)((double)a.r_beta[f0 + k] - mean * inv);
        }
        ra = make_float4(s[0], s[1], s[2], s[3]); rb = make_float4(t[0], t[1], t[2], t[3]);
    }

    float4 pre[NIT], preg[MODE == 1 ? NIT : 1];
    auto issue_loads = [&](int tile) __attribute__((always_inline)) {
        const int th = tile / a.tiles_w, tw = tile - th * a.tiles_w;
        const int h0 = th * TH2, w0 = tw * TW2;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int f = tid + STG * i;
            int pos = f / CQ;
            int r = pos / PW2, c = pos - r * PW2;
            int gh = h0 - 1 + r, gw = w0 - 1 + c;
            bool ok = (tid < STG) && (f < NF4) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            size_t pix = (size_t)gh * a.W + gw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), vg = v;
            if (ok) {
                if (!from_b) v = *(const float4*)(a.xa + pix * CA + 4 * c4);
                else {
                    v = *(const float4*)(a.xb + pix * CB + 4 * (c4 - CQA));
                    if (MODE == 1) vg = *(const float4*)(a.g + pix * (2 * CB) + 4 * (c4 - CQA));
                }
            }
            pre[i] = v;
            if (MODE == 1) preg[i] = vg;
        }
    };
    auto sig = [](float x) { return 1.0f / (1.0f + expf(-x)); };
    auto write_slab = [&](int tile, float* buf) __attribute__((always_inline)) {
        const int th = tile / a.tiles_w, tw = tile - th * a.tiles_w;
        const int h0 = th * TH2, w0 = tw * TW2;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int f = tid + STG * i;
            if (tid >= STG || f >= NF4) continue;
            int pos = f / CQ;
            int r = pos / PW2, c = pos - r * PW2;
            int gh = h0 - 1 + r, gw = w0 - 1 + c;
            bool ok = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            float4 v = pre[i];
            if (MODE == 1 && from_b && ok) {       // xb = sigmoid(LN(g_r)) * h ; zero outside the image
                float4 gq = preg[i];
                v.x *= sig(gq.x * ra.x + rb.x); v.y *= sig(gq.y * ra.y + rb.y);
                v.z *= sig(gq.z * ra.z + rb.z); v.w *= sig(gq.w * ra.w + rb.w);
            }
            *(float4*)(buf + pos * S + 4 * c4) = v;
        }
    };

    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = ((V * wave + v) * PW2 + n) * S + 4 * kq;
    const int a_off = (kq * COUT + n) * 4;
    float bias4[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) bias4[m][k] = a.bias ? a.bias[m * 16 + 4 * kq + k] : 0.f;
    float st_s[MT], st_q[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { st_s[m] = 0.f; st_q[m] = 0.f; }

    int tile = blockIdx.x;
    if (tile < ntiles) issue_loads(tile);
    if (tile < ntiles) write_slab(tile, slab);
    __syncthreads();
    int it = 0;
    for (; tile < ntiles; tile += gridDim.x, ++it) {
        const float* cur = slab + (it & 1) * SLAB_FLOATS;
        float* nxt = slab + ((it + 1) & 1) * SLAB_FLOATS;
        const int next_tile = tile + gridDim.x;
        const bool more = next_tile < ntiles;
        if (more) issue_loads(next_tile);

        f32x4 acc[MT][V];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {   // software-pipelined sweep (see conv3d_mfma.hip): group g+1's LDS reads are interleaved
            // with group g's MFMAs; a lone wave per SIMD never idles the matrix pipe on LDS latency
            constexpr int NG = 9 * (CT / 16), NR = V + MT, NM = 4 * MT * V;
            f32x4 bv[2][V], av[2][MT];
            auto load_one = [&](int g, int r, f32x4 (&b)[V], f32x4 (&aop)[MT]) __attribute__((always_inline)) {
                const int tap = g / (CT / 16), s = g % (CT / 16);
                const int kh = tap / 3, kw = tap % 3;
                if (r < V) b[r] = *(const f32x4*)(cur + b_off[r] + (kh * PW2 + kw) * S + 16 * s);
                else aop[r - V] = *(const f32x4*)(wl + a_off + (r - V) * 64 + (tap * CQ + 4 * s) * WROW);
            };
#pragma unroll
            for (int r = 0; r < NR; ++r) load_one(0, r, bv[0], av[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (g + 1 < NG) load_one(g + 1, r, bv[(g + 1) & 1], av[(g + 1) & 1]);
#pragma unroll
                    for (int i = (r * NM) / NR; i < ((r + 1) * NM) / NR; ++i) {
                        const int j = i / (MT * V), m = (i / V) % MT, v = i % V;
                        acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][j], bv[g & 1][v][j], acc[m][v], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // store (+bias) and LayerNorm moments
        {
            const int th = tile / a.tiles_w, tw = tile - th * a.tiles_w;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                int h = th * TH2 + V * wave + v, w = tw * TW2 + n;
                if (h < a.H && w < a.W) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        f32x4 r = acc[m][v];
                        float4 o = make_float4(r[0] + bias4[m][0], r[1] + bias4[m][1], r[2] + bias4[m][2], r[3] + bias4[m][3]);
                        *(float4*)(a.y + ((size_t)h * a.W + w) * COUT + m * 16 + 4 * kq) = o;
                        st_s[m] += (o.x + o.y) + (o.z + o.w);
                        st_q[m] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                    }
                }
            }
        }
        if (more) write_slab(next_tile, nxt);
        __syncthreads();
    }

    if (a.stats) {
        float* red = slab;                          // dead now: [4 waves][MT][2]
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s = wave_sum(st_s[m]), q = wave_sum(st_q[m]);
            if (lane == 0) { red[(wave * MT + m) * 2] = s; red[(wave * MT + m) * 2 + 1] = q; }
        }
        __syncthreads();
        if (tid < MT * 2) {
            int m = tid >> 1, k = tid & 1;
            double t = 0.0;
            for (int wv = 0; wv < 4; ++wv) t += (double)red[(wv * MT + m) * 2 + k];
            atomicAdd(&a.stats[m * 2 + k], t);
        }
    }
}

// TensorFlow conv2d kernel (3,3,CT,COUT) -> [tap9][CT/4][COUT][4]
__global__ void gru_weight_layout_kernel(const float* __restrict__ w, int CT, int COUT, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * CT * COUT) return;
    int j = i & 3, r = i >> 2;
    int co = r % COUT; r /= COUT;
    int CQ = CT / 4;
    int ciq = r % CQ, tap = r / CQ;
    out[i] = w[((size_t)tap * CT + ciq * 4 + j) * COUT + co];
}

template <int CA, int CB, int COUT, int MODE>
int launch_gru2d(const Gru2dArgs& a0, hipStream_t st) {
    Gru2dArgs a = a0;
    a.tiles_h = (a.H + TH2 - 1) / TH2;
    a.tiles_w = (a.W + TW2 - 1) / TW2;
    const int ntiles = a.tiles_h * a.tiles_w;
    const int grid = ntiles < 256 ? ntiles : 256;
    constexpr int CT = CA + CB;
    size_t smem = (size_t)(9 * CT * COUT + 2 * (TH2 + 2) * PW2 * (CT + 8)) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_cat_mfma_kernel<CA, CB, COUT, MODE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv2d_cat_mfma_kernel<CA, CB, COUT, MODE><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_gru_weight_layout(const float* w, int CT, int COUT, float* out, hipStream_t st) {
    gru_weight_layout_kernel<<<mvs_cdiv(9 * CT * COUT, 256), 256, 0, st>>>(w, CT, COUT, out);
    return (int)hipGetLastError();
}

// gate convolution of cell 1: [x | h] -> 2F raw gates (+bias), LayerNorm moments of reset | update
int mvs_gru1_gates_mfma(const float* x, const float* h, const float* wprep, const float* bias, int H,
                        int W, int CA, int F, float* g, double* stats, hipStream_t st) {
    if (!(CA == 32 && F == 16)) return MVS_E_SHAPE;
    Gru2dArgs a{x, h, nullptr, nullptr, nullptr, nullptr, wprep, bias, g, stats, H, W, 0, 0};
    return launch_gru2d<32, 16, 32, 0>(a, st);
}

// candidate convolution of cell 1: [x | sigmoid(LN(g_r)) * h] -> F (+bias), LayerNorm moments
int mvs_gru1_out_mfma(const float* x, const float* h, const float* g, const double* g_stats,
                      const float* r_gamma, const float* r_beta, const float* wprep,
                      const float* bias, int H, int W, int CA, int F, float* c, double* stats,
                      hipStream_t st) {
    if (!(CA == 32 && F == 16)) return MVS_E_SHAPE;
    Gru2dArgs a{x, h, g, g_stats, r_gamma, r_beta, wprep, bias, c, stats, H, W, 0, 0};
    return launch_gru2d<32, 16, 16, 1>(a, st);
}
