// fp32-MFMA kernel for the two consumers of the cost volume: the full-resolution 32 -> 8 stride-1
// convolution (3dconv0_1) and, fused into the same pass, the 32 -> 16 stride-2 convolution
// (3dconv1_0) (mvsnet/cnn_wrapper/mvsnetworks.py:130-134; tf.layers.conv3d SAME, network.py:203-215).
// 3dconv0_1 alone is 60 % of the regulariser's FLOPs, and both layers read the same 503 MB volume:
// fused, the volume is staged through LDS once instead of being streamed from HBM twice.
//
// Same input-stationary plane march as conv3d_mfma.hip (a workgroup owns an 8 x 16 (h x w) column,
// stages every input plane into LDS once, and each staged plane feeds the output planes q+1, q, q-1
// through the (kd, cout) weight rows), with three changes:
//
//   * Row packing (stride-1 part).  (kd, cout) is 24 rows = 1.5 MFMA row tiles; the generic kernel
//     pads the third 8-row block to a full tile (75 % useful).  Here a wave owns two adjacent output
//     rows r, r+1 and the third block of BOTH rows shares one tile: against the staged row r-1+i
//     (i = 0..3) tile rows 0-7 carry the weights of kh = i (output row r) and rows 8-15 those of
//     kh = i-1 (output row r+1) -- the same B operand is tap kh = i of row r and tap kh = i-1 of
//     row r+1.  Per (kw, ci-group) that is 6 + 4 = 10 MFMA tiles instead of 12 (90 % useful rows).
//   * XOR-swizzled slab.  Positions are stored at a 128-B pitch (no padding) with the 16-B channel
//     slot XORed by (column >> 1) & 7: the (col = lane&15, k-quad = lane>>4) ds_read_b128 groups stay
//     bank-conflict free and the slab shrinks by 11 %, which keeps two workgroups per CU.
//   * Stride-2 part.  The workgroup's slab covers a 4 x 8 patch of stride-2 outputs (two 16-voxel
//     column tiles).  Its 55 KB of weights do not fit in LDS next to the slab, so K = (kh, kw, ci)
//     is split four ways by ci: wave w keeps the A fragments of ci 8w..8w+7 for all 27 taps x 16
//     couts in 54 registers, accumulates its quarter of every output, and the four partial tiles
//     are summed through 6 KB of LDS when an output plane completes (every second input plane).
//
// Operand reads of the next (kw, ci-group, row) step are issued between the MFMAs of the current one.
#include "conv_common.h"
#include <cstdlib>
#include <cstdio>

namespace {

constexpr int TW = CONV_TW;
constexpr int TH = 8;
constexpr int PW = TW + 2;
constexpr int CIN = 32, COUT = 8, CQ = CIN / 4;
constexpr int COUT2 = 16;                      // stride-2 layer
constexpr int NPOS = (TH + 2) * PW;            // 180 staged positions
constexpr int NF4 = NPOS * CQ;                 // 1440 float4
constexpr int NIT = (NF4 + 255) / 256;         // 6
constexpr int NROWS = 3 * COUT;                // 24
constexpr int WROW = NROWS * 4;                // floats per (tap, ci-quad)
constexpr int W_FLOATS = 9 * CQ * WROW;        // 6912
constexpr int SP = CIN;                        // slab pitch (floats), swizzled
constexpr int ROWP = PW * SP;                  // floats per staged row
constexpr int SLAB_FLOATS = NPOS * SP;         // 5760
constexpr int KH_FLOATS = 3 * CQ * WROW;       // weight floats per kh
constexpr int RED_FLOATS = 3 * 2 * 64 * 4;     // partial stride-2 tiles of waves 1..3
constexpr int LDS_FLOATS = 2 * SLAB_FLOATS + W_FLOATS + 4;   // + zero block

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;

// float offset of 16-B channel slot `slot` at staged (row, col)
__device__ __forceinline__ int slab_off(int row, int col, int slot) {
    return (row * PW + col) * SP + ((slot ^ ((col >> 1) & 7)) << 2);
}

struct FuseArgs {
    const float* w2;      // stride-2 weights, TensorFlow layout (3,3,3,32,16)
    float* y2;            // (D/2, H/2, W/2, 16) raw output
    double* stats2;       // (2,16) float64 sums or null
    // The 240 workgroups of a metric-size launch (one per CU) all finish together: their float64 atomics on the 48
    // sum addresses serialised into a ~19 us tail.  With slots they go to (slots, 2, C) partial rows that the
    // consumers' bn_affine4 adds up (regnet.hip: 8 rows for 3dconv0_1, 4 for 3dconv1_0 -- what fits the layer's
    // 128-double statistics slab).
    int slots1, slots2;
    // SPAN schedule (conv3d_c8_kernel<.., SPAN = true>): groups of `span_g` tiles are cut into `span_m` plane ranges over the
    // tiles' concatenated depth; range k of a group is planes [span_b[k], span_b[k+1]) of that sequence and may cross ONE tile
    // boundary (two segments).  See launch_c8.
    int span_g, span_m;
    int span_b[17];
};

template <bool FUSE, bool AFF, bool SPAN = false>
__global__ void __launch_bounds__(256, 2)
conv3d_c8_kernel(ConvArgs a, FuseArgs fa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* slab = smem;                            // [2][NPOS][SP]
    float* wl = smem + 2 * SLAB_FLOATS;            // [9 taps][CQ][3 kd][8 co][4], then 4 zeros
    float* red = wl + W_FLOATS + 4;                // FUSE: [3 waves][2 tiles][64 lanes][4]
    constexpr int ZOFF = W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar
    const int n = lane & 15, kq = lane >> 4;

    const int tiles_w = (a.W + TW - 1) / TW;
    // What this workgroup marches: one (tile, plane range), or with SPAN up to two -- the end of one tile's depth and the
    // start of the next tile's.
    int seg_tile[2], seg_d0[2], seg_d1[2], nseg = 1;
    if (SPAN) {
        // workgroup w -> XCD w % 8 (round-robin dispatch): an XCD works through consecutive groups, i.e. neighbouring tiles
        const int per_xcd = gridDim.x >> 3, j = blockIdx.x >> 3, x = blockIdx.x & 7;
        const int lin = (gridDim.x & 7) ? (int)blockIdx.x : x * per_xcd + j;
        const int grp = lin / fa.span_m, k = lin - grp * fa.span_m;
        const int lo = fa.span_b[k], hi_ = fa.span_b[k + 1];
        const int t0 = lo / a.D, t1 = (hi_ - 1) / a.D;
        seg_tile[0] = grp * fa.span_g + t0; seg_d0[0] = lo - t0 * a.D; seg_d1[0] = min(hi_, (t0 + 1) * a.D) - t0 * a.D;
        seg_tile[1] = grp * fa.span_g + t1; seg_d0[1] = 0; seg_d1[1] = hi_ - t1 * a.D;
        nseg = t1 != t0 ? 2 : 1;
    } else {
        seg_tile[0] = xcd_swizzle(blockIdx.x, gridDim.x);
        seg_d0[0] = blockIdx.z * a.planes_per_wg;
        seg_d1[0] = min(seg_d0[0] + a.planes_per_wg, a.D);
        seg_tile[1] = 0; seg_d0[1] = 0; seg_d1[1] = 0;
    }
    int h0 = (seg_tile[0] / tiles_w) * TH, w0 = (seg_tile[0] % tiles_w) * TW;
    int d0 = seg_d0[0], d1 = seg_d1[0];
    int T = d1 - d0 + 2;                           // input planes d0-1 .. d1

    if (a.wprep) load_prepared_weights(wl, a.wprep, W_FLOATS);
    else for (int i = tid; i < W_FLOATS; i += 256) {
        int j = i & 3;
        int r = (i >> 2) % NROWS;
        int g = (i >> 2) / NROWS;
        int ciq = g % CQ, tap = g / CQ;
        int kd = r / COUT, co = r - kd * COUT;
        wl[i] = a.w[(((size_t)(kd * 9 + tap)) * CIN + ciq * 4 + j) * a.cout_total + co];
    }
    if (tid < 4) wl[ZOFF + tid] = 0.f;

    // ---- staging ---------------------------------------------------------------------------------
    const int c4 = tid % CQ;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AFF && a.xs) { sc = *(const float4*)(a.xs + 4 * c4); sh = *(const float4*)(a.xb + 4 * c4); }
    else if (AFF && a.bn.stats) bn_affine4(a.bn, 4 * c4, sc, sh);

    float4 pre[NIT];
    int goff[NIT], loff[NIT];
    auto set_goff = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int f = tid + 256 * i;
            if (f >= NF4) f -= 256;                // spare threads of the last piece redo their previous one
            int pos = f / CQ;
            int r = pos / PW, c = pos - r * PW;
            int gh = h0 - 1 + r, gw = w0 - 1 + c;
            bool inb = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
            goff[i] = inb ? ((gh * a.W + gw) * CIN + 4 * c4) * 4 : OOB;      // byte offset inside a plane
            loff[i] = slab_off(r, c, c4);
        }
    };
    set_goff();
    // Buffer addressing (the launcher guarantees < 2 GB tensors): 32-bit byte offsets, and an offset
    // with bit 31 set is out of range -> loads return 0 (= SAME padding), stores are dropped.
    const int plane_bytes = a.H * a.W * CIN * 4;
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.D * plane_bytes, 0x00020000);
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.D * a.H * a.W * COUT * 4, 0x00020000);

    // piece i of a plane's staging: global -> registers (load_piece), registers -> LDS (stage_piece)
    // (branch-free: these run between the MFMAs of the sweep)
    auto load_piece = [&](int i, int q) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, goff[i] | (plane_ok ? 0 : OOB),
                                                           plane_ok ? q * plane_bytes : 0, 0);
        pre[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    auto stage_piece = [&](int i, int q, float* buf) __attribute__((always_inline)) {
        float4 v = pre[i];
        if (AFF) {                                       // SAME pads the NORMALISED input with 0
            const bool ok = (q >= 0) && (q < a.D) && goff[i] >= 0;
            v = bn_relu4(v, sc, sh, true);
            v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        }
        *(float4*)(buf + loff[i]) = v;
    };

    // ---- stride-1 accumulators: [0],[1] = blocks 0|1 (rows 0-7 | 8-15) of output rows r, r+1;
    // [2] = block 2 of row r (rows 0-7) and of row r+1 (rows 8-15).  Block b carries kd = (P-b) mod 3.
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    // B operand: staged row 2*wave + i (i = 0..3, an immediate), column n + kw, slot kq (s = 1 is ^16)
    int bcol[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) bcol[kw] = slab_off(2 * wave, n + kw, kq);
    const int hi = n >> 3;                         // 0: tile rows 0-7, 1: rows 8-15
    const int a_lane = (kq * NROWS + (n & 7)) * 4;

    // `extra(G)` is called once per step: the plane march hangs the next plane's staging on it so
    // that those VALU / LDS-write / global-load instructions issue in the shadow of the MFMAs
    auto sweep = [&](auto Pc, const float* buf, auto&& extra) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int KD_B0 = P % 3, KD_B1 = (P + 2) % 3, KD_B2 = (P + 1) % 3;
        const int a0 = a_lane + (hi ? KD_B1 : KD_B0) * COUT * 4;          // blocks 0|1
        const int ap = a_lane + KD_B2 * COUT * 4 - hi * KH_FLOATS;        // block 2: kh = i - hi
        f32x4 bv[2], pv[2], av[3];
        // step G = ((kw*2 + s)*4 + i): staged row i against the weights of (kw, ci group s)
        auto load_grp = [&](int G, int r) __attribute__((always_inline)) {
            const int i = G & 3, kw = G >> 3, s = (G >> 2) & 1;
            const int wconst = (kw * CQ + 4 * s) * WROW;
            if (r == 0) bv[G & 1] = *(const f32x4*)(buf + (bcol[kw] ^ (16 * s)) + i * ROWP);
            else if (r == 1) {
                int off = ap + i * KH_FLOATS + wconst;
                if (i == 0) off = hi ? ZOFF : off;       // rows 8-15 would need kh = -1
                if (i == 3) off = hi ? off : ZOFF;       // rows 0-7 would need kh = 3
                pv[G & 1] = *(const f32x4*)(wl + off);
            } else if (i < 3) av[i] = *(const f32x4*)(wl + a0 + i * KH_FLOATS + wconst);
        };
        constexpr int NG = 24;
#pragma unroll
        for (int r = 0; r < 3; ++r) load_grp(0, r);
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            const int i = G & 3;
            const int NC = (i == 0 || i == 3) ? 2 : 3;      // MFMA tiles of this step
            // the next step's operands are requested a whole step (8-12 MFMAs) ahead of their use
            if (G + 1 < NG) {
#pragma unroll
                for (int r = 0; r < 3; ++r) load_grp(G + 1, r);
            }
            __builtin_amdgcn_sched_barrier(0);
            extra(G);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    // i = 0: row r kh 0 | packed;  i = 1,2: row r+1 kh i-1 | row r kh i | packed;  i = 3: row r+1 kh 2 | packed
                    int t; float aval;
                    if (c == NC - 1) { t = 2; aval = pv[G & 1][j]; }
                    else if (i == 0) { t = 0; aval = av[0][j]; }
                    else if (c == 0) { t = 1; aval = av[i - 1][j]; }
                    else { t = 0; aval = av[i][j]; }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, bv[G & 1][j], acc[t], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // store + zero the block that has just received kd = 2 (block (P+1)%3), output plane o
    // per-lane byte offsets inside an output plane: rows r, r+1 (blocks 0|1) and r + (kq>>1) (block 2)
    int yoff[3];
    auto set_yoff = [&]() __attribute__((always_inline)) {
        const int w = w0 + n, co = 4 * (kq & 1);
        const int hs[3] = {h0 + 2 * wave, h0 + 2 * wave + 1, h0 + 2 * wave + (kq >> 1)};
#pragma unroll
        for (int i = 0; i < 3; ++i)
            yoff[i] = (hs[i] < a.H && w < a.W) ? ((hs[i] * a.W + w) * COUT + co) * 4 : OOB;
    };
    set_yoff();
    const int yplane_bytes = a.H * a.W * COUT * 4;
    auto retire = [&](auto Pc, int o) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int B = (P + 1) % 3;
        const bool plane_ok = (o >= d0) && (o < d1);
        auto emit = [&](f32x4& r, int yo) __attribute__((always_inline)) {
            if (plane_ok) {
                u32x4_t v = {__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, yrsrc, yo + o * yplane_bytes, 0, 0);
                if (yo >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
                }
            }
            r = (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        if (B < 2) {
            if ((kq >> 1) == B) { emit(acc[0], yoff[0]); emit(acc[1], yoff[1]); }
        } else {
            emit(acc[2], yoff[2]);
        }
    };

    // ---- stride-2 part ---------------------------------------------------------------------------
    // Column tile ct, lane column n: output (oh, ow) = (h0/2 + 2ct + (n>>3), w0/2 + (n&7)); input tap
    // (kh, kw) sits at staged (row 4ct + 2(n>>3) + kh + 1, col 2(n&7) + kw + 1).  This wave's k index
    // kq of step j is input channel 8*wave + 2kq + j (one ds_read_b64 per tap and tile).
    float wS[3][9][2];                             // dead (eliminated) when !FUSE
    f32x4 cur[2], prv[2];                          // output planes od_cur / od_cur - 1, per column tile
    int s2b[3];
    float st2_s[4] = {0.f, 0.f, 0.f, 0.f}, st2_q[4] = {0.f, 0.f, 0.f, 0.f};
    int fin_od = -1;                               // output plane whose partial tiles wait in `red`
    const int Ho = a.H / 2, Wo = a.W / 2;
    if (FUSE) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    wS[kd][tap][j] = fa.w2[((size_t)(kd * 9 + tap) * CIN + 8 * wave + 2 * kq + j) * COUT2 + n];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            s2b[kw] = slab_off(2 * (n >> 3), 2 * (n & 7) + kw + 1, 2 * wave + (kq >> 1)) + 2 * (kq & 1);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { cur[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; prv[ct] = cur[ct]; }
    }

    auto s2_sweep = [&](auto Ec, const float* buf) __attribute__((always_inline)) {
        constexpr bool EVEN = decltype(Ec)::value;   // even plane: kd 0 -> cur, kd 2 -> prv; odd: kd 1 -> cur
        f32x2 b2[2][2];
        auto ld = [&](int tap, f32x2 (&b)[2]) __attribute__((always_inline)) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) b[ct] = *(const f32x2*)(buf + s2b[kw] + (4 * ct + kh + 1) * ROWP);
        };
        ld(0, b2[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) ld(tap + 1, b2[(tap + 1) & 1]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float bval = b2[tap & 1][ct][j];
                    if (EVEN) {
                        cur[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wS[0][tap][j], bval, cur[ct], 0, 0, 0);
                        prv[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wS[2][tap][j], bval, prv[ct], 0, 0, 0);
                    } else {
                        cur[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wS[1][tap][j], bval, cur[ct], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // wave 0: sum the four K-quarters of output plane fin_od, store, BatchNorm sums
    auto s2_finish = [&]() __attribute__((always_inline)) {
        if (fin_od >= 0 && wave == 0) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 r = prv[ct];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    f32x4 p = *(const f32x4*)(red + ((w * 2 + ct) * 64 + lane) * 4);
                    r[0] += p[0]; r[1] += p[1]; r[2] += p[2]; r[3] += p[3];
                }
                const int oh = h0 / 2 + 2 * ct + (n >> 3), ow = w0 / 2 + (n & 7);
                if (oh < Ho && ow < Wo) {
                    float* dst = fa.y2 + ((((size_t)fin_od * Ho + oh) * Wo) + ow) * COUT2 + 4 * kq;
                    *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st2_s[k] += r[k]; st2_q[k] += r[k] * r[k]; }
                }
            }
        }
        fin_od = -1;
    };

    // ---- plane march -----------------------------------------------------------------------------
    // Plane q is swept while plane q+1 moves registers -> LDS (steps 6..11) and plane q+2 is requested
    // from global memory (steps 14..19).  The two workgroups of a CU fall into lock-step (the one that
    // is behind gets the whole matrix pipe while the other waits at its barrier), so whatever is NOT
    // hidden under the MFMAs is idle time for both: only the retire stores and the barrier are left.
    auto prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) load_piece(i, d0 - 1);
#pragma unroll
        for (int i = 0; i < NIT; ++i) stage_piece(i, d0 - 1, slab);
#pragma unroll
        for (int i = 0; i < NIT; ++i) load_piece(i, d0);
        __syncthreads();
    };

    auto plane = [&](auto Pc, int t) __attribute__((always_inline)) {
        const int q = d0 - 1 + t;
        float* cur_buf = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
        const bool in_vol = (q >= 0) && (q < a.D);
        // (past the last plane these stage / request planes nobody reads: harmless, and branch-free)
        auto extra = [&](int G) __attribute__((always_inline)) {
            if (G >= 6 && G < 6 + NIT) stage_piece(G - 6, q + 1, nxt);
            if (G >= 14 && G < 14 + NIT) load_piece(G - 14, q + 2);
        };
        if (FUSE) s2_finish();
        sweep(Pc, cur_buf, extra);                 // planes outside the volume are staged as zeros
        if (FUSE && q >= d0) {
            if (q & 1) {
                if (in_vol) s2_sweep(std::false_type{}, cur_buf);
            } else {
                // plane q = 2*od_cur: od_cur starts (kd 0), od_cur - 1 completes (kd 2)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { prv[ct] = cur[ct]; cur[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                if (in_vol) s2_sweep(std::true_type{}, cur_buf);
                const int od_prev = q / 2 - 1;
                if (2 * od_prev >= d0) {
                    if (wave > 0) {
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            *(f32x4*)(red + (((wave - 1) * 2 + ct) * 64 + lane) * 4) = prv[ct];
                    }
                    fin_od = od_prev;
                }
            }
        }
        retire(Pc, q - 1);
        __syncthreads();
    };
    for (int sg = 0; sg < (SPAN ? nseg : 1); ++sg) {
        if (SPAN && sg > 0) {
            // second segment: another tile, its first planes.  Whatever the first march left in the accumulators belongs to
            // planes past its range; the last stride-2 plane of the first range is flushed before its partials are overwritten.
            if (FUSE) s2_finish();
            h0 = (seg_tile[1] / tiles_w) * TH; w0 = (seg_tile[1] % tiles_w) * TW;
            d0 = seg_d0[1]; d1 = seg_d1[1]; T = d1 - d0 + 2;
            set_goff(); set_yoff();
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (FUSE) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) { cur[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; prv[ct] = cur[ct]; }
            }
        }
        prologue();
        for (int t = 0; t < T; t += 3) {
            plane(std::integral_constant<int, 0>{}, t);
            if (t + 1 < T) plane(std::integral_constant<int, 1>{}, t + 1);
            if (t + 2 < T) plane(std::integral_constant<int, 2>{}, t + 2);
        }
    }
    if (FUSE) {
        s2_finish();
        if (fa.stats2 && wave == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float sv = st2_s[k], qv = st2_q[k];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
                if (n == 0) {
                    double* s2p = fa.stats2 + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % fa.slots2) * 2 * COUT2;
                    atomicAdd(&s2p[4 * kq + k], (double)sv);
                    atomicAdd(&s2p[COUT2 + 4 * kq + k], (double)qv);
                }
            }
        }
    }

    // every lane's st_* are the sums of channels 4*(kq&1) .. +3: fold lanes l and l^32
    if (a.stats) stats_commit<COUT>(st_s, st_q, true, slab,
                                    a.stats + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % (FUSE ? fa.slots1 : 1)) * 2 * COUT,
                                    a.cout_total, 0);
}

template <bool FUSE>
int launch_c8(const ConvArgs& a0, const FuseArgs& fa, hipStream_t st) {
    ConvArgs a = a0;
    if ((long long)a.D * a.H * a.W * CIN * 4 >= (1LL << 31)) return MVS_E_SHAPE;   // 32-bit buffer offsets
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    if (FUSE) {
        // chunks start on even planes so that stride-2 output planes never straddle workgroups
        int best = 2; long long best_cost = 1LL << 60;
        for (int dr = 2; dr <= a.D; dr += 2) {
            long long wgs = (long long)tiles * ((a.D + dr - 1) / dr);
            long long cost = ((wgs + 255) / 256) * (dr + 3);
            if (cost < best_cost) { best_cost = cost; best = dr; }
        }
        a.planes_per_wg = best;
    } else {
        a.planes_per_wg = conv_pick_planes(a.D, tiles, 2);
    }
    dim3 grid(tiles, 1, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    const size_t smem = (size_t)(LDS_FLOATS + (FUSE ? RED_FLOATS : 0)) * sizeof(float);
    const bool aff = a.xs != nullptr || a.bn.stats != nullptr;
    static bool attr_done = false;       // per template instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3d_c8_kernel<FUSE, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3d_c8_kernel<FUSE, true>,
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if constexpr (FUSE) {
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3d_c8_kernel<true, false, true>,
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        }
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    if constexpr (FUSE) {
        // SPAN schedule (round 4): groups of G tiles share M workgroups that cut the tiles' concatenated depth into M equal
        // ranges; a range may cross one tile boundary (the workgroup then marches the end of one tile and the start of the
        // next: two more halo planes and a second prologue).  Taken when it costs fewer plane steps than whole chunks: at the
        // metric workload 160 tiles x 192 planes meet 512 workgroup slots -- three chunks of 64 leave 32 slots idle (64 + 3
        // steps), 5 tiles / 16 workgroups fill every slot with ranges of 60 (60 + 3, + 2 across a boundary): 581 -> 565 us,
        // 921 -> 935 depth maps/s on one box.  Ranges 8 planes shorter across a boundary (62 / 54) and a pairing of long with
        // short ranges on a CU measured the same (profiles/r04_pair_span_ab.txt).
        if (!aff && !mvs_hook(MVS_HOOK_CONV_NO_SPAN)) {      // (test hook: whole depth chunks, the schedule SPAN replaces)
            const long long chunk_cost = (((long long)tiles * grid.z + 511) / 512) * (a.planes_per_wg + 3);
            int bg = 0, bm = 0; long long bcost = chunk_cost;
            for (int G = 1; G <= 8; ++G) {
                if (tiles % G) continue;
                for (int M = G; M <= 16; ++M) {
                    const int total = G * a.D;
                    if (total % M || (long long)(tiles / G) * M > 512) continue;
                    const int L = total / M;
                    if ((L & 1) || L < 32 || L > a.D) continue;          // even starts (stride-2 planes), at most one boundary per range
                    const long long cost = L + 3 + ((a.D % L) ? 2 : 0);
                    if (cost < bcost) { bcost = cost; bg = G; bm = M; }
                }
            }
            if (bg) {
                FuseArgs f2 = fa;
                f2.span_g = bg; f2.span_m = bm;
                for (int k = 0; k <= bm; ++k) f2.span_b[k] = k * (bg * a.D / bm);
                conv3d_c8_kernel<true, false, true><<<dim3((tiles / bg) * bm), 256, smem, st>>>(a, f2);
                return (int)hipGetLastError();
            }
        }
    }
    if (aff) conv3d_c8_kernel<FUSE, true><<<grid, 256, smem, st>>>(a, fa);
    else conv3d_c8_kernel<FUSE, false><<<grid, 256, smem, st>>>(a, fa);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_conv3d_c8_launch(const ConvArgs& a, hipStream_t st) {
    if (a.x2 || a.cout_total != COUT) return MVS_E_SHAPE;
    return launch_c8<false>(a, FuseArgs{nullptr, nullptr, nullptr, 1, 1}, st);
}

// Both consumers of a 32-channel volume in one pass: y = conv3d(x, w 32->8, stride 1) as above and
// y2 = conv3d(x, w2 32->16, stride 2).  Even D, H, W only (SAME pads nothing in front).
int mvs_conv3d_c8_s2_launch(const ConvArgs& a, const float* w2, float* y2, double* stats2, hipStream_t st,
                            int slots1, int slots2) {
    if (a.x2 || a.cout_total != COUT || (a.D & 1) || (a.H & 1) || (a.W & 1)) return MVS_E_SHAPE;
    return launch_c8<true>(a, FuseArgs{w2, y2, stats2, slots1 > 1 ? slots1 : 1, slots2 > 1 ? slots2 : 1}, st);
}
