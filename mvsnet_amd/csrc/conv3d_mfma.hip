// fp32-MFMA implicit-GEMM 3x3x3 convolutions for gfx950 (R4/R5, K6a-d).
//
// Reference behaviour: tf.layers.conv3d / conv3d_transpose, SAME padding, no bias, NDHWC
// (mvsnet/cnn_wrapper/network.py:203-215,300-329), with the producer's training-mode BatchNorm +
// ReLU (+ additive skip) applied while the input is staged (network.py:457-459,492-509).
//
// Design ("input-stationary plane march"), written for CDNA4, not a translated CUDA tiling:
//   * A workgroup (4 waves, one per SIMD) owns a TH x 16 (h x w) column of the volume and marches
//     along depth.  Every input plane is staged into LDS exactly once (channel-last rows padded so
//     that the 16-lane ds_read_b128 groups hit 16 distinct 16-B bank slots) and used once.
//   * GEMM roles: rows (A operand) = (kd, cout) weight rows, columns (B operand) = 16 voxels along
//     w, K = (kh, kw, ci).  One staged plane q therefore feeds the three output planes q+1, q, q-1
//     (kd = 0, 1, 2) in the same sweep: 3x fewer LDS reads than an output-stationary loop, one
//     slab (double buffered) instead of a 3-plane ring, and Cout = 8 fills 24 of 32 rows (75 %)
//     instead of 8 of 16.
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD): lane l supplies A[row l&15][k l>>4]
//     and B[k l>>4][col l&15]; D regs r: row (l>>4)*4+r, col l&15.  K is ordered so that one
//     ds_read_b128 per operand feeds 4 consecutive MFMAs (ci = 16s + 4*(l>>4) + j, j = 0..3).
//   * The accumulator block that received kd = 0,1,2 on three successive planes is complete and is
//     stored (coalesced 16-B lanes), with per-channel sum / sum-of-squares for the consumer's
//     BatchNorm accumulated on the fly.  Blocks rotate through kd by re-basing the weight address
//     per plane (loop unrolled by 3), never by moving registers.
//   * Plane q is swept while plane q+1 moves registers -> LDS and plane q+2 is requested from global
//     memory, as branch-free "pieces" issued between the MFMA groups (32-bit buffer offsets: outside
//     the image = out of range = 0); one barrier per plane.
//   * The 32 -> 8 full-resolution layer has its own kernel (conv3d_c8.hip: row packing, fused
//     stride-2 consumer); the low-resolution layers use 2x8 / 4x4 voxel column tiles (S1Geom).
#include "conv_common.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------
// stride-1 convolution, input-stationary.  COUT = output channels handled by this workgroup (8|16).
// ------------------------------------------------------------------------------------------------
// Per-instance geometry.  The big full-resolution layer (32 -> 8) trades the conflict-free slab
// pitch (Cin+8) for Cin+4 (2-way conflicts on the B reads, LDS is ~12 % busy) so that two
// workgroups fit in a CU's 160 KB LDS: two waves per SIMD hide each other's LDS / barrier stalls.
// The workgroup tile is TH x TWG voxels, cut into 16-voxel MFMA column tiles of CR rows x 16/CR
// columns (CR = 1: 16 voxels along w; CR = 2, 4: 2x8 / 4x4 patches for the low-resolution layers,
// whose widths 40 and 20 are not multiples of 16 -- a 16-wide tiling would run 17 % / 38 % of the
// MFMA columns on padding).
template <int CIN, int COUT, int TH, int TWG = CONV_TW> struct S1Geom {
    static constexpr bool TIGHT = (CIN == 32 && COUT == 8 && TH == 8 && TWG == CONV_TW);
    static constexpr int S = TIGHT ? CIN + 4 : SlabGeom<CIN>::S;
    // LDS bytes = weights + two slabs; two workgroups per CU whenever that fits in 160 KB
    static constexpr int LDS_BYTES = (9 * (CIN / 4) * 3 * COUT * 4 + 2 * (TH + 2) * (TWG + 2) * S) * 4;
    static constexpr int WGS_PER_CU = (2 * LDS_BYTES <= 160 * 1024) ? 2 : 1;
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;      // buffer byte offset with bit 31 set: loads give 0, stores are dropped

// FUSE2 (round 4): 3dconv1_1 (16 -> 16, stride 1) and 3dconv2_0 (16 -> 32, stride 2) read the same tensor -- BN + ReLU of
// 3dconv1_0 (mvsnetworks.py:131-132,138-139) -- so the stride-2 layer rides in the stride-1 layer's plane march exactly as
// 3dconv1_0 rides in 3dconv0_1's (conv3d_c8.hip): the workgroup's staged 10 x 18 slab covers a 4 x 8 patch of stride-2
// outputs (two 16-voxel column tiles) x 32 output channels (two row tiles); its 55 KB of weights do not fit in LDS next to
// the slab, so K = (kd, kh, kw, ci) is split four ways by input channel: wave w keeps the A fragments of ci 4w .. 4w+3 for
// all 27 taps x 32 couts in 54 registers, accumulates its quarter of every output, and the four partial tiles are summed
// through 12 KB of LDS when an output plane completes (every second input plane).  One read and one normalisation of the
// input instead of two, one launch instead of two, and the stride-2 layer (0.31 of the MFMA peak as a kernel of its own:
// 40 tiles, 17 % halo planes) costs its 25 % of extra matrix work inside a kernel that already runs at 0.61.
struct Fuse2Args {
    const float* w2;      // stride-2 weights, TensorFlow layout (3,3,3,16,32)
    float* y2;            // (D/2, H/2, W/2, 32) raw output
    double* stats2;       // (slots2, 2, 32) float64 sums or null
    int slots2;
};
constexpr int FUSE2_RED_FLOATS = 3 * 2 * 2 * 64 * 4;      // partial tiles of waves 1..3: [wave][row tile][column tile][lane][4]

template <int CIN, int COUT, int TH, bool HAS_X2, int TWG = CONV_TW, int CR = 1, bool FUSE2 = false>
__global__ void __launch_bounds__(256, (S1Geom<CIN, COUT, TH, TWG>::WGS_PER_CU))
conv3d_s1_kernel(ConvArgs a, Fuse2Args fa) {
    static_assert(!FUSE2 || (CIN == 16 && COUT == 16 && TH == 8 && TWG == CONV_TW && CR == 1 && !HAS_X2), "the fused stride-2 consumer is built for 3dconv1_1 + 3dconv2_0");
    constexpr int S = S1Geom<CIN, COUT, TH, TWG>::S;
    constexpr int PW = TWG + 2;                    // staged row width with halo
    constexpr int CC = 16 / CR;                    // columns of a 16-voxel MFMA column tile
    constexpr int TPR = TWG / CC;                  // column tiles per tile row
    static_assert(TH % CR == 0 && TWG % CC == 0 && (TH * TWG) % 64 == 0, "tile must split into 4 x V column tiles");
    constexpr int NPOS = (TH + 2) * PW;
    constexpr int CQ = CIN / 4;                    // float4 per position
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    constexpr int V = TH * TWG / 64;               // 16-voxel column tiles per wave
    constexpr int NROWS = 3 * COUT;                // (kd, co) weight rows
    constexpr int MT = (NROWS + 15) / 16;          // 16-row MFMA tiles
    constexpr int WROW = NROWS * 4;                // floats per (tap, ci-quad) weight group
    constexpr int W_FLOATS = 9 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");
    static_assert(NF4 >= 256, "spare threads of the last piece redo their previous one");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                              // [9 taps][CQ][3 kd][COUT][4]
    float* slab = smem + W_FLOATS;                 // [2][NPOS][S]
    float* red2 = slab + 2 * SLAB_FLOATS;          // FUSE2: [3 waves][2 row tiles][2 column tiles][64 lanes][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;

    const int tiles_w = (a.W + TWG - 1) / TWG;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TWG;
    const int co_base = blockIdx.y * COUT;
    const int d0 = blockIdx.z * a.planes_per_wg;
    const int d1 = min(d0 + a.planes_per_wg, a.D);
    const int T = d1 - d0 + 2;                     // input planes d0-1 .. d1
    // this lane's voxel (row, column inside the workgroup tile) in each of the wave's column tiles
    int vrow[V], vcol[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int ct = V * wave + v;
        vrow[v] = (ct / TPR) * CR + n / CC;
        vcol[v] = (ct % TPR) * CC + n % CC;
    }

    // ---- weights -> LDS, re-laid out as [tap][ci/4][kd][co][ci%4] ------------------------------
    if (a.wprep) load_prepared_weights(wl, a.wprep, W_FLOATS);
    else for (int i = tid; i < W_FLOATS; i += 256) {
        int j = i & 3;
        int r = (i >> 2) % NROWS;
        int g = (i >> 2) / NROWS;                  // tap * CQ + ciq
        int ciq = g % CQ, tap = g / CQ;
        int kd = r / COUT, co = r - kd * COUT;
        wl[i] = a.w[(((size_t)(kd * 9 + tap)) * CIN + ciq * 4 + j) * a.cout_total + co_base + co];
    }

    // ---- staging -----------------------------------------------------------------------------------
    const int c4 = tid % CQ;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sc2 = sc, sh2 = sh;
    const bool has_aff = a.xs != nullptr || a.bn.stats != nullptr;
    if (a.xs) { sc = *(const float4*)(a.xs + 4 * c4); sh = *(const float4*)(a.xb + 4 * c4); }
    else if (a.bn.stats) bn_affine4(a.bn, 4 * c4, sc, sh);
    const bool has_aff2 = HAS_X2 && (a.x2s != nullptr || a.bn2.stats != nullptr);
    if (HAS_X2 && a.x2s) { sc2 = *(const float4*)(a.x2s + 4 * c4); sh2 = *(const float4*)(a.x2b + 4 * c4); }
    else if (HAS_X2 && a.bn2.stats) bn_affine4(a.bn2, 4 * c4, sc2, sh2);
    // ReLU floor: -inf turns max(v, lo) into the identity when there is no producer BatchNorm
    const float lo = has_aff ? 0.f : -INFINITY, lo2 = has_aff2 ? 0.f : -INFINITY;

    float4 pre[NIT];
    float4 pre2[HAS_X2 ? NIT : 1];

    // Per-thread staging map, identical for every plane: byte offset inside one input plane (bit 31
    // set = outside the image: the buffer load returns 0) and float offset inside the LDS slab.
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int f = tid + 256 * i;
        if (f >= NF4) f -= 256;                    // spare threads of the last piece redo their previous one
        int pos = f / CQ;
        int r = pos / PW, c = pos - r * PW;
        int gh = h0 - 1 + r, gw = w0 - 1 + c;
        bool inb = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? ((gh * a.W + gw) * CIN + 4 * c4) * 4 : OOB;
        loff[i] = pos * S + 4 * c4;
    }
    const int plane_bytes = a.H * a.W * CIN * 4;
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.D * plane_bytes, 0x00020000);
    const auto x2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_X2 ? a.x2 : a.x), 0, a.D * plane_bytes, 0x00020000);
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, a.D * a.H * a.W * a.cout_total * 4, 0x00020000);

    // piece i of a plane's staging: global -> registers (load_piece), registers -> LDS (stage_piece);
    // branch-free, because they are issued between the MFMAs of the sweep
    auto ld4b = [](auto rsrc, int voff, int soff) __attribute__((always_inline)) {
        u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    auto load_piece = [&](int i, int q) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
        const int voff = goff[i] | (plane_ok ? 0 : OOB), soff = plane_ok ? q * plane_bytes : 0;
        pre[i] = ld4b(xrsrc, voff, soff);
        if (HAS_X2) pre2[i] = ld4b(x2rsrc, voff, soff);
    };
    auto stage_piece = [&](int i, int q, float* buf) __attribute__((always_inline)) {
        // SAME padding pads the NORMALISED input with 0: positions outside the volume are forced to 0
        const bool ok = (q >= 0) && (q < a.D) && goff[i] >= 0;
        float4 v = pre[i];
        v.x = fmaxf(v.x * sc.x + sh.x, lo); v.y = fmaxf(v.y * sc.y + sh.y, lo);
        v.z = fmaxf(v.z * sc.z + sh.z, lo); v.w = fmaxf(v.w * sc.w + sh.w, lo);
        if (HAS_X2) {
            float4 u = pre2[i];
            v.x += fmaxf(u.x * sc2.x + sh2.x, lo2); v.y += fmaxf(u.y * sc2.y + sh2.y, lo2);
            v.z += fmaxf(u.z * sc2.z + sh2.z, lo2); v.w += fmaxf(u.w * sc2.w + sh2.w, lo2);
        }
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        *(float4*)(buf + loff[i]) = v;
    };

    // ---- accumulators ------------------------------------------------------------------------------
    f32x4 acc[MT][V];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    // lane constants: B-operand base (floats) per voxel tile, A-operand row decomposition
    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = (vrow[v] * PW + vcol[v]) * S + 4 * kq;
    // row m of tile mt -> block (mt*16+m)/COUT, channel (mt*16+m)%COUT
    int row_blk[MT], row_co[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { int r = m * 16 + n; row_blk[m] = r / COUT; row_co[m] = r % COUT; }

    // One plane sweep; P = plane counter mod 3 (static), block b carries kd = (P - b) mod 3.
    // `extra(g)` is called once per operand group: the plane march hangs the next planes' staging on
    // it, so that those VALU / LDS-write / global-load instructions issue between the MFMAs.
    constexpr int NG = 9 * (CIN / 16);
    auto sweep = [&](auto Pc, const float* buf, auto&& extra) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        int a_off[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int b = row_blk[m];
            int kd = (b < 3) ? ((P - b + 3) % 3) : 0;      // pad blocks read harmless finite weights
            a_off[m] = (kq * NROWS + kd * COUT + row_co[m]) * 4;
        }
        // Operand groups g = (kh, kw, s): one ds_read_b128 per operand tile feeds 4 k-steps.  The
        // reads of group g+1 are issued before the MFMAs of group g (register double buffer), so a
        // lone wave per SIMD never waits on LDS latency with an idle matrix pipe.
        f32x4 bv[2][V], av[2][MT];
        auto load_grp = [&](int g, f32x4 (&b)[V], f32x4 (&aop)[MT]) __attribute__((always_inline)) {
            const int tap = g / (CIN / 16), s = g % (CIN / 16);
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int v = 0; v < V; ++v) b[v] = *(const f32x4*)(buf + b_off[v] + (kh * PW + kw) * S + 16 * s);
#pragma unroll
            for (int m = 0; m < MT; ++m) aop[m] = *(const f32x4*)(wl + a_off[m] + (tap * CQ + 4 * s) * WROW);
        };
        load_grp(0, bv[0], av[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_grp(g + 1, bv[(g + 1) & 1], av[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            extra(g);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][j], bv[g & 1][v][j], acc[m][v], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // store + zero the block that has just received kd = 2 (block (P+1)%3), output plane o
    int yoff[V];                                        // per-lane byte offset inside an output plane
    const int yplane_bytes = a.H * a.W * a.cout_total * 4;
    auto retire = [&](auto Pc, int o) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int B = (P + 1) % 3;
        constexpr int mt = (B * COUT) / 16;                 // tile holding the block
        constexpr int row0 = (B * COUT) % 16;               // first row of the block inside the tile
        const bool plane_ok = (o >= d0) && (o < d1);
        const bool mine = (4 * kq >= row0) && (4 * kq < row0 + COUT);   // lane group holds rows 4kq .. 4kq+3
        const int co = 4 * kq - row0;                       // first of this lane's 4 channels
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (mine) {
                if (plane_ok) {
                    f32x4 r = acc[mt][v];
                    u32x4_t u = {__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(u, yrsrc, yoff[v] + co * 4, o * yplane_bytes, 0);
                    if (yoff[v] >= 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
                    }
                }
                acc[mt][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int h = h0 + vrow[v], w = w0 + vcol[v];
        yoff[v] = (h < a.H && w < a.W) ? ((h * a.W + w) * a.cout_total + co_base) * 4 : OOB;
    }

    // ---- fused stride-2 consumer (FUSE2) -------------------------------------------------------------
    // Column tile ct, lane column n: output (oh, ow) = (h0/2 + 2ct + (n>>3), w0/2 + (n&7)); input tap (kh, kw) sits at staged
    // (row 4ct + 2(n>>3) + kh + 1, column 2(n&7) + kw + 1) -- SAME with even sizes pads nothing in front.  This wave's k index kq
    // is input channel 4*wave + kq (one ds_read_b32 per tap and column tile).
    constexpr int COUT2 = 32;
    float wS2[FUSE2 ? 3 : 1][FUSE2 ? 9 : 1][2];   // [kd][tap][row tile]: A fragments, dead (eliminated) when !FUSE2
    f32x4 cur2[2][2], prv2[2][2];                  // [row tile][column tile]: output planes od_cur / od_cur - 1
    int s2b[3];
    float st2_s[2][4], st2_q[2][4];
    int fin_od = -1;                               // output plane whose partial tiles wait in `red2`
    const int Ho2 = a.H / 2, Wo2 = a.W / 2;
    if (FUSE2) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
                    wS2[kd][tap][rt] = fa.w2[((size_t)(kd * 9 + tap) * CIN + 4 * wave + kq) * COUT2 + 16 * rt + n];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) s2b[kw] = ((2 * (n >> 3) + 1) * PW + 2 * (n & 7) + kw + 1) * S + 4 * wave + kq;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { cur2[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; prv2[rt][ct] = cur2[rt][ct]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { st2_s[rt][k] = 0.f; st2_q[rt][k] = 0.f; }
        }
    }
    // The stride-2 layer's 9 taps of one plane, after the stride-1 sweep of that plane; B values requested two taps ahead (an odd
    // plane's four MFMAs per tap do not cover an LDS read's latency).  (Hanging the taps between the operand groups of the
    // stride-1 sweep instead was measured slower, round 4: 109-112 us for the fused launch against 100 us.)
    auto s2_sweep = [&](auto Ec, const float* buf) __attribute__((always_inline)) {
        constexpr bool EVEN = decltype(Ec)::value;   // even plane: kd 0 -> cur, kd 2 -> prv; odd: kd 1 -> cur
        float b2[3][2];
        auto ld = [&](int tap, float (&b)[2]) __attribute__((always_inline)) {
            const int kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) b[ct] = buf[s2b[kw] + (4 * ct + kh) * PW * S];
        };
        ld(0, b2[0]); ld(1, b2[1]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 2 < 9) ld(tap + 2, b2[(tap + 2) % 3]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const float bval = b2[tap % 3][ct];
                    if (EVEN) {
                        cur2[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wS2[0][tap][rt], bval, cur2[rt][ct], 0, 0, 0);
                        prv2[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wS2[FUSE2 ? 2 : 0][tap][rt], bval, prv2[rt][ct], 0, 0, 0);
                    } else {
                        cur2[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wS2[FUSE2 ? 1 : 0][tap][rt], bval, cur2[rt][ct], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // wave 0: sum the four K-quarters of output plane fin_od, store, BatchNorm sums.  (Spreading this over the four waves --
    // wave t finishing tile t -- was measured slower, round 4: 108 us for the fused launch against 100 us, with every partial
    // going through LDS; picking a wave's own partial out of prv2[][] by the wave number put the accumulators in scratch: 134 us.)
    auto s2_finish = [&]() __attribute__((always_inline)) {
        if (fin_od >= 0 && wave == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    f32x4 r = prv2[rt][ct];
#pragma unroll
                    for (int w = 0; w < 3; ++w) {
                        const f32x4 p = *(const f32x4*)(red2 + (((w * 2 + rt) * 2 + ct) * 64 + lane) * 4);
                        r[0] += p[0]; r[1] += p[1]; r[2] += p[2]; r[3] += p[3];
                    }
                    const int oh = h0 / 2 + 2 * ct + (n >> 3), ow = w0 / 2 + (n & 7);
                    if (oh < Ho2 && ow < Wo2) {
                        float* dst = fa.y2 + ((((size_t)fin_od * Ho2 + oh) * Wo2) + ow) * COUT2 + 16 * rt + 4 * kq;
                        *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { st2_s[rt][k] += r[k]; st2_q[rt][k] += r[k] * r[k]; }
                    }
                }
        }
        fin_od = -1;
    };

    // ---- plane march -------------------------------------------------------------------------------
    // Plane q is swept while plane q+1 moves registers -> LDS (first half of the operand groups) and
    // plane q+2 is requested from global memory (second half): with one or two waves per SIMD,
    // whatever is not issued under the MFMAs is idle matrix-pipe time.
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, d0 - 1);
#pragma unroll
    for (int i = 0; i < NIT; ++i) stage_piece(i, d0 - 1, slab);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, d0);
    __syncthreads();

    auto plane = [&](auto Pc, int t) __attribute__((always_inline)) {
        const int q = d0 - 1 + t;
        float* cur = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
        // (past the last plane these stage / request planes nobody reads: harmless, and branch-free)
        auto extra = [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                if ((i * (NG / 2)) / NIT == g) stage_piece(i, q + 1, nxt);
                if (NG / 2 + (i * (NG - NG / 2)) / NIT == g) load_piece(i, q + 2);
            }
        };
        if (FUSE2) s2_finish();
        sweep(Pc, cur, extra);                      // planes outside the volume are staged as zeros
        if (FUSE2 && q >= d0) {
            const bool in_vol = q < a.D;
            if (q & 1) {
                if (in_vol) s2_sweep(std::false_type{}, cur);
            } else {
                // plane q = 2 * od_cur: od_cur starts (kd 0), od_cur - 1 completes (kd 2)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) { prv2[rt][ct] = cur2[rt][ct]; cur2[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                if (in_vol) s2_sweep(std::true_type{}, cur);
                const int od_prev = q / 2 - 1;
                if (2 * od_prev >= d0) {
                    if (wave > 0) {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct)
                                *(f32x4*)(red2 + ((((wave - 1) * 2 + rt) * 2 + ct) * 64 + lane) * 4) = prv2[rt][ct];
                    }
                    fin_od = od_prev;
                }
            }
        }
        retire(Pc, q - 1);
        __syncthreads();
    };
    for (int t = 0; t < T; t += 3) {
        plane(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) plane(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) plane(std::integral_constant<int, 2>{}, t + 2);
    }

    if (FUSE2) {
        s2_finish();
        if (fa.stats2 && wave == 0) {
            double* s2p = fa.stats2 + (size_t)((blockIdx.x + gridDim.x * blockIdx.z) % (fa.slots2 > 1 ? fa.slots2 : 1)) * 2 * COUT2;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float sv = st2_s[rt][k], qv = st2_q[rt][k];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
                    if (n == 0) {
                        atomicAdd(&s2p[16 * rt + 4 * kq + k], (double)sv);
                        atomicAdd(&s2p[COUT2 + 16 * rt + 4 * kq + k], (double)qv);
                    }
                }
        }
        __syncthreads();      // (stats_commit below re-uses the slab)
    }

    // ---- BatchNorm statistics ----------------------------------------------------------------------
    // channel quad of st_*: COUT=16: blocks start at row 0 of their tile -> quad kq.  COUT=8: blocks
    // 0,2 sit in lane groups 0,1 and block 1 in groups 2,3 -> quad kq&1, fold lanes l and l^32.
    if (a.stats) stats_commit<COUT>(st_s, st_q, COUT == 8, slab, conv_stats_row(a), a.cout_total, co_base);
}

template <int CIN, int COUT, int TH, int TWG = CONV_TW, int CR = 1>
int launch_s1(const ConvArgs& a0, int Cout, hipStream_t st, int slots = 256) {
    ConvArgs a = a0;
    if ((long long)a.D * a.H * a.W * (CIN > Cout ? CIN : Cout) * 4 >= (1LL << 31)) return MVS_E_SHAPE;   // 32-bit buffer offsets
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TWG - 1) / TWG);
    const int groups = Cout / COUT;
    a.planes_per_wg = conv_pick_planes(a.D, (long long)tiles * groups, 2, slots);
    dim3 grid(tiles, groups, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    size_t smem = (size_t)S1Geom<CIN, COUT, TH, TWG>::LDS_BYTES;
    static bool attr_done = false;       // per template instantiation
    if (!attr_done) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s1_kernel<CIN, COUT, TH, true, TWG, CR>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s1_kernel<CIN, COUT, TH, false, TWG, CR>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (a.x2) conv3d_s1_kernel<CIN, COUT, TH, true, TWG, CR><<<grid, 256, smem, st>>>(a, Fuse2Args{});
    else conv3d_s1_kernel<CIN, COUT, TH, false, TWG, CR><<<grid, 256, smem, st>>>(a, Fuse2Args{});
    return (int)hipGetLastError();
}

// 3dconv1_1 + 3dconv2_0 in one pass (FUSE2 above).  Even D, H, W only; MVS_E_SHAPE otherwise (the caller runs the two layers apart).
int launch_s1_fuse2(const ConvArgs& a0, const Fuse2Args& fa, hipStream_t st) {
    ConvArgs a = a0;
    if ((a.D & 1) || (a.H & 1) || (a.W & 1) || a.x2 || a.cout_total != 16) return MVS_E_SHAPE;
    if ((long long)a.D * a.H * a.W * 16 * 4 >= (1LL << 31)) return MVS_E_SHAPE;                      // 32-bit buffer offsets
    const int tiles = ((a.H + 7) / 8) * ((a.W + CONV_TW - 1) / CONV_TW);
    // chunks start on even planes so that stride-2 output planes never straddle workgroups
    int best = 2; long long best_cost = 1LL << 60;
    for (int dr = 2; dr <= a.D; dr += 2) {
        const long long wgs = (long long)tiles * ((a.D + dr - 1) / dr);
        const long long cost = ((wgs + 255) / 256) * (dr + 3);
        if (cost < best_cost) { best_cost = cost; best = dr; }
    }
    a.planes_per_wg = best;
    // measurement hook (round 6): an even number of planes per workgroup -- 8 at the metric size = 480 workgroups,
    // two per CU, two waves per SIMD instead of one: 100.5 against 99.0 us, 968 against 969 depth maps/s
    // (profiles/r06_fuse2_planes.txt).  The float64 BatchNorm atomics arrive in another order: results differ in the last bits.
    if (const int hk = mvs_hook(MVS_HOOK_FUSE2_PLANES)) a.planes_per_wg = hk < a.D ? hk : a.D;
    dim3 grid(tiles, 1, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    const size_t smem = (size_t)S1Geom<16, 16, 8>::LDS_BYTES + FUSE2_RED_FLOATS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3d_s1_kernel<16, 16, 8, false, CONV_TW, 1, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv3d_s1_kernel<16, 16, 8, false, CONV_TW, 1, true><<<grid, 256, smem, st>>>(a, fa);
    return (int)hipGetLastError();
}

}  // namespace

// test hook: MVS_GENERIC_C8=1 keeps the 32 -> 8 layer on the generic kernel (A/B timing, parity)

static int conv_dispatch(ConvArgs& a, int Cin, int Cout, int stride, hipStream_t st) {
    if (stride == 1) {
        if (Cout == 1) return mvs_conv3d_out_launch(a, Cin, st);
        if (Cin == 1) return mvs_conv3d_in1_launch(a, Cout, st);     // input gradient of the one-channel output layer
        if (Cin == 8) return mvs_conv3d_k8_launch(a, Cout, st);      // input gradient of 3dconv0_1 (8 -> 32)
        if (a.wprep_bf && mvs_conv3d_bf16x3_supported(Cin, Cout)) return mvs_conv3d_s1_bf16x3(a, Cin, Cout, st);
        if (mvs_conv3d_os_covers(0, Cin, Cout)) return mvs_conv3d_os_launch(a, 0, Cin, Cout, st);     // low-resolution levels
        if (Cin == 32 && Cout == 8 && !a.x2) {
            int rc = mvs_conv3d_c8_launch(a, st);
            if (rc != MVS_E_SHAPE) return rc;            // >= 2 GB volumes stay on the generic kernel
        }
        if (Cin == 32 && Cout == 8) return launch_s1<32, 8, 8>(a, Cout, st);
        // (round 4: 4-row tiles for this shape -- 480 workgroups, two per CU, instead of 240 with one wave per SIMD -- 70.1 against
        // 70.9 us for 3dconv1_1: the plane march is not short of waves, it is at ~0.73 of the matrix pipe per busy CU)
        if (Cin == 16 && Cout % 16 == 0) return launch_s1<16, 16, 8>(a, Cout, st);
        // widths that are not multiples of 16 (the /4 and /8 levels of a 160-wide volume): 2x8 / 4x4 column tiles
        if (Cin == 32 && Cout % 16 == 0 && a.W % 16 != 0 && a.W % 8 == 0) return launch_s1<32, 16, 8, 8, 2>(a, Cout, st);
        if (Cin == 32 && Cout % 16 == 0) return launch_s1<32, 16, 8>(a, Cout, st);
        if (Cin == 64 && Cout % 8 == 0 && a.W % 16 != 0 && a.W % 4 == 0 && a.H % 16 == 0) return launch_s1<64, 8, 16, 4, 4>(a, Cout, st);
        if (Cin == 64 && Cout % 8 == 0) return launch_s1<64, 8, 4>(a, Cout, st);
        if (Cin == 16 && Cout == 8) return launch_s1<16, 8, 8>(a, Cout, st);
        return MVS_E_SHAPE;
    }
    auto pad_before = [](int n) { int o = (n + 1) / 2; int t = (o - 1) * 2 + 3 - n; return t < 0 ? 0 : t / 2; };
    a.pd = pad_before(a.D); a.ph = pad_before(a.H); a.pw = pad_before(a.W);
    if (mvs_conv3d_os_covers(1, Cin, Cout)) return mvs_conv3d_os_launch(a, 1, Cin, Cout, st);
    return mvs_conv3d_s2_mfma(a, Cin, Cout, st);
}

int mvs_conv3d_mfma(const float* x, const float* xs, const float* xb, const float* x2,
                    const float* x2s, const float* x2b, const float* w, int D, int H, int W,
                    int Cin, int Cout, int stride, float* y, double* stats, hipStream_t st) {
    ConvArgs a{x, xs, xb, x2, x2s, x2b, w, y, stats, D, H, W, Cout, 0, 0, 0, 0, {}, {}, nullptr, nullptr};
    return conv_dispatch(a, Cin, Cout, stride, st);
}

int mvs_deconv3d_mfma(const float* x, const float* xs, const float* xb, const float* x2,
                      const float* x2s, const float* x2b, const float* w, int D, int H, int W,
                      int Cin, int Cout, float* y, double* stats, hipStream_t st) {
    ConvArgs a{x, xs, xb, x2, x2s, x2b, w, y, stats, D, H, W, Cout, 0, 0, 0, 0, {}, {}, nullptr, nullptr};
    return mvs_deconv3d_mfma_launch(a, Cin, Cout, st);
}

// Variants taking the producers' raw BatchNorm sums (regnet.hip): no bn_finalize launch in between.
int mvs_conv3d_mfma_bn(const float* x, const BnSrc& bn, const float* x2, const BnSrc& bn2,
                       const float* w, const float* wprep, const unsigned short* wprep_bf, int D, int H,
                       int W, int Cin, int Cout, int stride, float* y, double* stats, hipStream_t st, int stats_slots) {
    ConvArgs a{x, nullptr, nullptr, x2, nullptr, nullptr, w, y, stats, D, H, W, Cout, 0, 0, 0, 0, bn, bn2, wprep, wprep_bf, stats_slots};
    return conv_dispatch(a, Cin, Cout, stride, st);
}

int mvs_deconv3d_mfma_bn(const float* x, const BnSrc& bn, const float* x2, const BnSrc& bn2,
                         const float* w, const float* wprep, int D, int H, int W, int Cin, int Cout,
                         float* y, double* stats, hipStream_t st, int stats_slots) {
    ConvArgs a{x, nullptr, nullptr, x2, nullptr, nullptr, w, y, stats, D, H, W, Cout, 0, 0, 0, 0, bn, bn2, wprep, nullptr, stats_slots};
    return mvs_deconv3d_mfma_launch(a, Cin, Cout, st);
}

// 3dconv1_1 (x -> y, 16 -> 16, stride 1) and 3dconv2_0 (x -> y2, 16 -> 32, stride 2) over the same BN + ReLU input in one pass
int mvs_conv3d_s1s2_16_bn(const float* x, const BnSrc& bn, const float* w, const float* wprep, int D, int H, int W, float* y,
                          double* stats, const float* w2, float* y2, double* stats2, hipStream_t st, int stats_slots) {
    if (mvs_hook(MVS_HOOK_CONV_NO_FUSE2)) return MVS_E_SHAPE;      // test hook: the caller then runs the two layers apart (tests/test_gpu_parity.py)
    ConvArgs a{x, nullptr, nullptr, nullptr, nullptr, nullptr, w, y, stats, D, H, W, 16, 0, 0, 0, 0, bn, BnSrc{}, wprep, nullptr, stats_slots};
    return launch_s1_fuse2(a, Fuse2Args{w2, y2, stats2, stats_slots}, st);
}

// ---- weight pre-layout (run once per weight set, mvs_regnet_prepare_f32) ---------------------------
namespace {
// conv:   out[g][tap9][ci/4][kd][co][ci%4] <- w[(kd*9+tap)][ci][g*G+co]       (TensorFlow (3,3,3,Cin,Cout))
// deconv: out[g][tap27][ci/4][co][ci%4]    <- w[tap][g*G+co][ci]              (TensorFlow (3,3,3,Cout,Cin))
__global__ void weight_layout_kernel(const float* __restrict__ w, int kind, int Cin, int Cout, int G,
                                     float* __restrict__ out) {
    const int total = 27 * Cin * Cout;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int per_group = 27 * Cin * G;
    const int g = i / per_group;
    int r = i - g * per_group;
    const int j = r & 3; r >>= 2;
    const int CQ = Cin / 4;
    if (kind == 2) {
        const int co = r % G; r /= G;
        const int ciq = r % CQ, tap = r / CQ;
        out[i] = w[((size_t)tap * Cout + g * G + co) * Cin + ciq * 4 + j];
    } else {
        const int rows = 3 * G;
        const int row = r % rows; r /= rows;
        const int ciq = r % CQ, tap = r / CQ;
        const int kd = row / G, co = row - kd * G;
        out[i] = w[((size_t)(kd * 9 + tap) * Cin + ciq * 4 + j) * Cout + g * G + co];
    }
}
}  // namespace

int mvs_conv_weight_layout(const float* w, int kind, int Cin, int Cout, float* out, hipStream_t st) {
    if (mvs_conv3d_os_covers(kind, Cin, Cout)) return mvs_conv3d_os_weight_layout(w, kind, Cin, Cout, out, st);
    const int G = conv_coutg(kind, Cin, Cout);
    if (G == 0 || (Cin % 4)) return MVS_E_SHAPE;
    const int total = 27 * Cin * Cout;
    weight_layout_kernel<<<mvs_cdiv(total, 256), 256, 0, st>>>(w, kind, Cin, Cout, G, out);
    return (int)hipGetLastError();
}
