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
    constexpr int MT = COUT / 16;
    constexpr int V = 2;
    constexpr int WROW = COUT * 4;
    constexpr int W_FLOATS = 9 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(CA % 16 == 0 && CB % 16 == 0 && COUT % 16 == 0, "tiling");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;
    float* slab = smem + W_FLOATS;                  // [2][NPOS][S]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int ntiles = a.tiles_h * a.tiles_w;

    copy_weights_to_lds(wl, a.wprep, W_FLOATS);      // prepared layout, eight loads in flight per thread

    // ---- staging, in pieces that are issued between the MFMAs of the sweep (branch-free) -------------
    // Piece i of a tile: global -> registers (load_piece), registers -> LDS (stage_piece).  The xa
    // channels (pieces 0..NA-1) and the xb channels (pieces NA..NA+NB-1) are staged by separate pieces,
    // so that the reset-gate sigmoid of MODE 1 is only evaluated where it applies.  Addresses are
    // 32-bit buffer offsets: rows above / below the image fall outside the buffer and read 0, only
    // the horizontal wrap needs a test.
    constexpr int CQB = CB / 4;
    constexpr int NFA = NPOS * CQA, NFB = NPOS * CQB;
    constexpr int NA = (NFA + 255) / 256, NB = (NFB + 255) / 256, NIT = NA + NB;
    static_assert(256 % CQA == 0 && 256 % CQB == 0 && NFA >= 256 && NFB >= 256, "staging map");
    const int qb = tid % CQB;                        // this thread's channel quad in the xb pieces
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;      // MODE 1: LayerNorm affine of the reset gate
    if (MODE == 1) {
        const double cnt = (double)a.H * a.W * CB;
        double mean = a.g_stats[0] / cnt;
        double var = a.g_stats[1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        float s[4], t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double inv = (double)a.r_gamma[4 * qb + k] / sqrt(var + 1e-12);
            s[k] = (float)inv; t[k] = (float)((double)a.r_beta[4 * qb + k] - mean * inv);
        }
        ra = make_float4(s[0], s[1], s[2], s[3]); rb = make_float4(t[0], t[1], t[2], t[3]);
    }
    const int bytes_a = a.H * a.W * CA * 4, bytes_b = a.H * a.W * CB * 4;
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.xa, 0, bytes_a, 0x00020000);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.xb, 0, bytes_b, 0x00020000);
    const auto rsrc_g = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 1 ? a.g : a.xb), 0, MODE == 1 ? 2 * bytes_b : bytes_b, 0x00020000);
    // per piece: byte offset relative to the tile's (h0-1, w0-1) pixel, staged column, LDS float offset
    int poff[NIT], pcol[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const bool isb = i >= NA;
        const int cq = isb ? CQB : CQA, nf = isb ? NFB : NFA;
        int f = tid + 256 * (isb ? i - NA : i);
        if (f >= nf) f -= 256;                       // spare threads of a family's last piece redo their previous one
        const int pos = f / cq, q = f % cq;
        const int r = pos / PW2, c = pos - r * PW2;
        pcol[i] = c;
        poff[i] = ((r * a.W + c) * (isb ? CB : CA) + 4 * q) * 4;
        loff[i] = pos * S + (isb ? CA : 0) + 4 * q;
    }
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    auto ldb = [](auto rsrc, int voff) __attribute__((always_inline)) {
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    float4 pre[NIT], preg[MODE == 1 ? NB : 1];
    auto load_piece = [&](int i, int tile) __attribute__((always_inline)) {
        const int tl = tile < ntiles ? tile : 0;     // past the end: a harmless reload of tile 0
        const int th = tl / a.tiles_w, h0 = th * TH2, w0 = (tl - th * a.tiles_w) * TW2;
        const int base = (h0 - 1) * a.W + (w0 - 1);  // may be negative: such offsets are out of range as unsigned
        const bool ok = (unsigned)(w0 - 1 + pcol[i]) < (unsigned)a.W;
        if (i < NA) pre[i] = ldb(rsrc_a, ok ? base * (CA * 4) + poff[i] : (int)0x80000000);
        else {
            pre[i] = ldb(rsrc_b, ok ? base * (CB * 4) + poff[i] : (int)0x80000000);
            // the reset gate is channels [0, CB) of the (H, W, 2*CB) gate tensor: pixel stride doubles
            if (MODE == 1) preg[i - NA] = ldb(rsrc_g, ok ? base * (2 * CB * 4) + 2 * poff[i] - 16 * qb : (int)0x80000000);
        }
    };
    auto sig = [](float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); };
    auto stage_piece = [&](int i, float* buf) __attribute__((always_inline)) {
        float4 v = pre[i];                           // zeros outside the image (SAME padding)
        if (MODE == 1 && i >= NA) {                  // xb = sigmoid(LN(g_r)) * h (convgru.py:97,101,107)
            const float4 gq = preg[i - NA];
            v.x *= sig(gq.x * ra.x + rb.x); v.y *= sig(gq.y * ra.y + rb.y);
            v.z *= sig(gq.z * ra.z + rb.z); v.w *= sig(gq.w * ra.w + rb.w);
        }
        *(float4*)(buf + loff[i]) = v;
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

    // Tile t is swept while tile t+1 moves registers -> LDS (first half of the operand groups) and
    // tile t+2 is requested from memory (second half): with one wave per SIMD everything that is not
    // issued under the MFMAs -- above all the global-load latency of a 3 us tile -- would be exposed.
    const int stride = gridDim.x;
    int tile = blockIdx.x;
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, tile);
#pragma unroll
    for (int i = 0; i < NIT; ++i) stage_piece(i, slab);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, tile + stride);
    __syncthreads();
    int it = 0;
    for (; tile < ntiles; tile += stride, ++it) {
        const float* cur = slab + (it & 1) * SLAB_FLOATS;
        float* nxt = slab + ((it + 1) & 1) * SLAB_FLOATS;

        f32x4 acc[MT][V];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {   // operand reads of group g+1 are issued before the MFMAs of group g (register double buffer)
            constexpr int NG = 9 * (CT / 16);
            f32x4 bv[2][V], av[2][MT];
            auto load_grp = [&](int g, f32x4 (&b)[V], f32x4 (&aop)[MT]) __attribute__((always_inline)) {
                const int tap = g / (CT / 16), s = g % (CT / 16);
                const int kh = tap / 3, kw = tap % 3;
#pragma unroll
                for (int v = 0; v < V; ++v) b[v] = *(const f32x4*)(cur + b_off[v] + (kh * PW2 + kw) * S + 16 * s);
#pragma unroll
                for (int m = 0; m < MT; ++m) aop[m] = *(const f32x4*)(wl + a_off + m * 64 + (tap * CQ + 4 * s) * WROW);
            };
            load_grp(0, bv[0], av[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_grp(g + 1, bv[(g + 1) & 1], av[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NIT; ++i) {
                    if ((i * (NG / 2)) / NIT == g) stage_piece(i, nxt);
                    if (NG / 2 + (i * (NG - NG / 2)) / NIT == g) load_piece(i, tile + 2 * stride);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int v = 0; v < V; ++v)
                            acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][j], bv[g & 1][v][j], acc[m][v], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // store (+bias) and LayerNorm moments
        {
            const int th = tile / a.tiles_w, h0 = th * TH2, w0 = (tile - th * a.tiles_w) * TW2;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                int h = h0 + V * wave + v, w = w0 + n;
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
