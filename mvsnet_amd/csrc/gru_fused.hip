// The recurrent sweep as a skewed software pipeline: TWO launches per depth plane on ONE stream carry all three ConvGRU
// cells, prob_conv and the winner-take-all update (mvsnet/convgru.py:82-122, mvsnet/model.py:676-734).
//
// The recurrence couples cell k of plane d only to cell k-1 of plane d and to cell k of plane d-1, and every convolution
// of a cell is followed by a whole-image LayerNorm (convgru.py:30-31) -- a global barrier.  Rounds 2-4 ran the three cells
// as a wavefront over four HIP streams (gru.hip): ~7 launches per plane, cross-stream events per group of planes, and
// 136 KB cell-1 workgroups that time-share the CUs with 26 KB small-cell workgroups (round-4 device trace: the median
// cell-1 workgroup starts 34-53 us late, a plane costs the SUM of its kernels).  Here the cells are skewed by one plane
// each instead, so that everything between two LayerNorm barriers is one launch:
//
//   gates launch  G(t):  cell 1 gate conv of plane t   | cell 2 gate conv of plane t-1 | cell 3 gate conv of plane t-2
//                        | prob_conv + exp + WTA of plane t-3
//   output launch C(t):  cell 1 candidate conv of plane t | cell 2 candidate of plane t-1 | cell 3 candidate of plane t-2
//
// G(t) forms, while it stages its tile, the states the three gate convolutions need -- s1(t-1), s2(t-2), s3(t-3), each the
// blend u*h + (1-u)*tanh(LN c) of the plane before (convgru.py:98,102,114-120) -- and these are exactly the operands of the
// four convolutions: [x(t) | s1(t-1)], [s1(t-1) | s2(t-2)], [s2(t-2) | s3(t-3)], [s3(t-3)].  ONE staged slab (54 channels
// per position, halo 1) feeds all of them; the tile's own pixels of the three states go to memory for C(t).  C(t) stages
// [x(t) | r1*s1 | s1 | r2*s2 | s2 | r3*s3] with the reset gates r = sigmoid(LN g_r) folded in (convgru.py:97,101,107).
// Dependencies: everything G(t) reads was completed by C(t-1) or earlier, everything C(t) reads by G(t) -- stream order is
// the only synchronisation, there are no side streams, events or calibration, and the sweep captures into a hipGraph as is.
//
// Inside a workgroup (8 waves, one 8 x 16 pixel tile at a time, persistent over its tiles, the next tile staged under the
// current tile's matrix instructions as in gru_mfma.hip): cell 1 (90 % of the MACs) on v_mfma_f32_16x16x4_f32 exactly as
// conv2d_cat_mfma_kernel does it (rows = output channels, columns = 16 pixels, x and h halves in separate accumulators:
// the same bits); cells 2 / 3 on v_mfma_f32_4x4x1_16B_f32 from the same slab (a lane = one pixel, four result registers =
// four output channels, the multiply-add chain of conv2d_small_body: the same bits), one small job per wave:
//   G: waves 0-3 cell 2 gates (pixel half x output-channel quad), waves 4-5 cell 3 gates, waves 6-7 prob_conv + WTA
//   C: waves 0-1 cell 2 candidate, waves 2-3 cell 3 candidate
// so that each SIMD's matrix pipe carries one large and one small job beside its two cell-1 rows.
#include "conv_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

constexpr int FMAXV = 8;                 // reference views per launch (mvs_gru_wta_batch_f32)
constexpr int FNT = 512, FTH = 8, FTW = 16, FPW = FTW + 2, FNPOS = (FTH + 2) * FPW;      // 180 staged positions per tile

// Copies of a plane's LayerNorm sums: every workgroup of a launch adds its partial sums with float64 atomics, and atomics on one
// cache line are performed one after the other by the L2 -- 256 workgroups x 3 cells on the same 144 bytes were 6 us of every
// plane (round 6: a build without these atomics ran the c3 sweep in 20.87 instead of 22.41 ms; tools/r6_gru_nostat_diag.patch).
// A workgroup adds to copy blockIdx.x % 8 (256 bytes apart: other lines); the next launch's prologue adds the copies up.
constexpr int GRU_FUSED_SLOTS = 8;
constexpr int GRU_FUSED_SLOT_STRIDE = 32;

struct FusedCell {
    // byte offsets inside the view's workspace block (every activation tensor of the sweep lives in it)
    unsigned h;             // G: the state BEFORE the blend (entered plane p-1), or the state itself when !blend;  C: the state s(p-1)
    unsigned c;             // G: raw candidate convolution of plane p-1 (H,W,F)
    unsigned g;             // G: raw gate convolution of plane p-1 (update half read);  C: of plane p (reset half read)   (H,W,2F)
    unsigned h_out;         // G: receives s(p-1) on the tile's own pixels
    unsigned y;             // the output: G raw gates of plane p (H,W,2F);  C raw candidate of plane p (H,W,F)
    const double* st_in;    // 6 doubles [reset s,q | update s,q | candidate s,q] in each of GRU_FUSED_SLOTS copies, GRU_FUSED_SLOT_STRIDE doubles apart: G of plane p-1, C of plane p (view 0's block)
    double* st_out;         // 6 doubles of plane p (x copies): G adds [0..3], C adds [4..5]
    const float* bias;
    const float *ga, *gb, *oa, *ob;      // G: update gamma / beta, candidate gamma / beta;  C: reset gamma / beta (ga, gb)
    int conv, blend;        // this cell's convolution is live (its plane exists) / the blend of plane p-1 is formed on load
};
struct FusedArgs {
    unsigned x;             // (H,W,32) cost slice of cell 1's plane (byte offset in the block)
    FusedCell cell[3];
    const float* w1;        // cell-1 weights of this phase, [tap9][12][COUT][4] (gru_weight_slice_kernel)
    const float* wsmall;    // small-cell tables of this phase (gru_small_table_kernel)
    char* ws;               // view 0's workspace block; view v's is v * vstride bytes further (one buffer resource per view)
    unsigned max_prob, depth_image, exp_sum; int wta;    // winner-take-all accumulators (G): byte offsets in the block, live flag
    int H, W, tiles_h, tiles_w, wg_per_view;
    size_t vstride;         // bytes between the workspace blocks of consecutive views (< 2^31)
    long long* trace;       // diagnostic (null in the product): [0] = record counter, then 8 x int64 per workgroup (mvs_gru_fused_trace)
    int trace_cap, launch_id;
};
struct FusedDepth { float v[FMAXV]; };   // depth value of the WTA plane, per view (its own kernel argument: indexed on the kernarg)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
// Every activation tensor is addressed through ONE buffer resource over the view's workspace block: `voff` = byte offset in the
// block (tensor offset included), or FBAD = nothing to do for this lane (out of range: a load returns 0, a store is dropped).
// The instruction's SGPR offset stays 0 -- on purpose: with a register in the soffset field the compiler's hazard recognizer
// assumes that the "store of more than 64 bits followed by a VALU write of its data registers" hazard does not exist and
// schedules such a write right behind the store; on gfx950 it does exist (first build of this file: the x component of a float4
// store, overwritten by the next instruction, reached memory corrupted for the last lanes of each row of 16 -- run-to-run
// differences at 16 x 16 pixels; tools/store_hazard_probe.hip reproduces it in isolation).  With soffset = 0 the compiler does
// insert ONE wait state -- still one short of the two the hardware needs (profiles/r05_store_hazard_probe.txt: 1 584 of 8 M stores
// corrupted with one) -- so what GUARANTEES the two wait states in every build is the ISA scan, not the compiler:
// tools/store_hazard_scan.py (both successors of every branch, loop back-edges included) run by tests/test_abi_and_io.py.
constexpr int FBAD = (int)0x80000000;
__device__ __forceinline__ float4 ld_b128(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ float ld_b32(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
}
__device__ __forceinline__ void st_b128(__amdgpu_buffer_rsrc_t rsrc, int voff, float x, float y, float z, float w) {
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)}, rsrc, voff, 0, 0);
}
__device__ __forceinline__ void st_b64(__amdgpu_buffer_rsrc_t rsrc, int voff, float x, float y) {
    __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){__float_as_uint(x), __float_as_uint(y)}, rsrc, voff, 0, 0);
}
__device__ __forceinline__ void st_b32(__amdgpu_buffer_rsrc_t rsrc, int voff, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rsrc, voff, 0, 0);
}

// Sum over the 64 lanes of a wave on the DPP path; LANE 63 receives the total (the other lanes partial sums): inclusive row
// shifts by 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (lanes a shift
// leaves without a source, and rows a broadcast does not address, add 0).  18 vector instructions per float64 against the 12
// ds_bpermute round trips of the shuffle form; eight of these close every launch (tail 3.0 -> 1.6 us, profiles/r05_gru_fused_trace.txt).
__device__ __forceinline__ double wave_sum_dpp63(double v) {
#define MVS_DPP_ADD(ctrl, rmask) { const int lo_ = __double2loint(v), hi_ = __double2hiint(v); \
        const int l2_ = __builtin_amdgcn_update_dpp(0, lo_, ctrl, rmask, 0xf, false), h2_ = __builtin_amdgcn_update_dpp(0, hi_, ctrl, rmask, 0xf, false); \
        v += __hiloint2double(h2_, l2_); }
    MVS_DPP_ADD(0x111, 0xf) MVS_DPP_ADD(0x112, 0xf) MVS_DPP_ADD(0x114, 0xf) MVS_DPP_ADD(0x118, 0xf)      // row_shr:1, 2, 4, 8
    MVS_DPP_ADD(0x142, 0xa) MVS_DPP_ADD(0x143, 0xc)                                                      // row_bcast:15, row_bcast:31
#undef MVS_DPP_ADD
    return v;
}
// PHASE 0 = gates launch G(t), PHASE 1 = output launch C(t).  Slab channel map (floats per staged position):
//   G (S = 56): x 0..31 | s1 32..47 | s2 48..51 | s3 52..53 | 54,55 unused          (56 = 48 + the conflict-free pad of gru_mfma.hip)
//   C (S = 72): x 0..31 | r1*s1 32..47 | s1 48..63 | r2*s2 64..67 | s2 68..71 ;  r3*s3 in a 4-float side slab
// STEADY: an interior plane step -- every cell live and blending, the WTA update on: the flags are compile-time constants (the
// first and last three steps of a sweep take the general instantiation).
//
// Memory operations and the wait counters.  On gfx950 loads AND stores retire in order through one counter (vmcnt), and the
// compiler can only count what it sees on every path: a store inside `if (own pixel)` makes it wait for ALL outstanding memory
// operations before the next staged piece is consumed -- i.e. for the acknowledgement of stores issued a few instructions
// earlier, between the matrix instructions of the sweep (first build of this kernel: 11.3 us per tile against 6.5 us of matrix
// time).  So every memory operation inside the tile loop is issued by every lane of every wave on a straight line, with the
// lane's offset pushed out of range when the lane has nothing to write / read; LDS reads never sit inside a branch either, and
// the wave-specific small jobs (branches) contain matrix / vector instructions and LDS reads only.
//
// Vector instructions are what the tile loop has to save: fp32 vector and fp32 matrix instructions share the SIMD's multipliers
// (DESIGN 4), so every one of them adds to the 6.5 us of matrix time of a tile.  Hence: per-thread byte offsets of every load and
// store computed once (a tile costs one add + one select per memory operation), the blend as ONE division per channel with
// LayerNorm affines pre-scaled for v_exp_f32, per-lane float64 moment sums instead of wave reductions per tile.
template <int PHASE, bool STEADY>
__global__ void __launch_bounds__(FNT, 1)
gru_fused_kernel(FusedArgs a, FusedDepth dv) {
    constexpr int S = PHASE == 0 ? 56 : 72;
    constexpr int COUT = PHASE == 0 ? 32 : 16, MT = COUT / 16;
    constexpr int CQ = 12, WROW = COUT * 4, W1_FLOATS = 9 * CQ * WROW;
    constexpr int T2 = 9 * 5 * 16, T3 = 9 * 2 * 16;                 // small tables: 20-channel / 6-channel input, [tap][quad][m][4]
    constexpr int WS_FLOATS = PHASE == 0 ? 2 * T2 + T3 + 20 : T2 + T3;                // G: + prob_conv (18 weights, bias, pad)
    constexpr int SLAB = FNPOS * S;
    constexpr int XA2 = PHASE == 0 ? 32 : 48;                       // first channel of cell 2's input [s1 | (r2*)s2] in the slab
    constexpr int XA3 = PHASE == 0 ? 48 : 68;                       // first channel of cell 3's xa = s2

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                                // cell-1 weights
    float* wsm = wl + W1_FLOATS;                     // small tables
    float* slab = wsm + WS_FLOATS;                   // [2][FNPOS][S]
    float* mini = slab + 2 * SLAB;                   // C: [2][FNPOS][4] r3*s3 (.xy), then [FNPOS][4] that absorbs the stores a wave has no use for
    // LayerNorm (scale, shift) quads, pre-scaled for v_exp_f32 (see the prologue):
    //   G: 0-3 s1 update, 4-7 s1 candidate, 8 s2 update, 9 s2 candidate, 10 s3 update (2), 11 s3 candidate (2);  C: 0-3 s1 reset, 4 s2 reset, 5 s3 reset (2)
    __shared__ __attribute__((aligned(16))) float lnS[12][4], lnT[12][4];
    __shared__ double red[8][8];
    __builtin_amdgcn_s_setprio(3);

    const long long tr0 = a.trace ? wall_clock64() : 0;
    const long long cy0 = a.trace ? clock64() : 0;         // shader clocks (s_memtime) next to the 100 MHz wall clock: the launch's real frequency
    long long tr1 = 0, tr2 = 0, tr3 = 0, tpa = 0, tpb = 0, tpc = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int view = blockIdx.x / a.wg_per_view, j = blockIdx.x - view * a.wg_per_view;
    const size_t vo = (size_t)view * a.vstride;      // view v's block (no modified copy of `a`: it would live in scratch memory)
    auto vp = [vo](auto* p) { return p ? (decltype(p))((const char*)p + vo) : p; };
    auto cell_sel = [&](int k, auto f) { return k == 0 ? f(a.cell[0]) : k == 1 ? f(a.cell[1]) : f(a.cell[2]); };
    // this workgroup's tiles: an XCD (workgroups b, b+8, ... share an L2) takes a contiguous band of the view's tiles, so that
    // the halo rows / columns neighbouring tiles share are read from the same L2
    const int tiles = a.tiles_h * a.tiles_w;
    int first, stride, end;
    if ((a.wg_per_view & 7) == 0) {
        const int xcd = j & 7, tb = (tiles + 7) >> 3;
        first = xcd * tb + (j >> 3); stride = a.wg_per_view >> 3; end = min(tiles, (xcd + 1) * tb);
    } else { first = j; stride = a.wg_per_view; end = tiles; }

    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)vp(a.ws), 0, (int)a.vstride, 0x00020000);
    const bool lo = wave < 4;                        // wave-uniform: this wave's small piece is s2 (else s3)
    const int FS = lo ? 4 : 2;                       // channels of the small family
    const bool blend1 = STEADY || a.cell[0].blend, blend_s = STEADY || (lo ? a.cell[1].blend : a.cell[2].blend);
    const bool live1 = STEADY || a.cell[0].conv, live2 = STEADY || a.cell[1].conv, live3 = STEADY || a.cell[2].conv;

    // ---- staging pieces (global -> registers -> LDS), hung between the matrix instructions of the sweep ---------------------
    // pieces 0-2: x (1440 float4 per tile), pieces 3-4: s1 (720 channel quads), piece 5: s2 (waves 0-3) / s3 (waves 4-7), 180 each.
    // Per thread and memory operation: the byte offset for the tile whose staged window starts at pixel 0 (tensor offset, position
    // in the window, channel quad); a tile adds (pixel index of its window origin) * (bytes per pixel of the tensor).
    const int q8 = tid & 7, q4 = tid & 3;
    const int spos = min(tid & 255, FNPOS - 1);      // the small piece's position
    // Seven piece slots per thread: 0-3 x, 4-5 s1, 6 the small one.  The split is UNEVEN on purpose: waves 0-3 take four x pieces and two
    // s1 pieces (items 0-1023 of 1440, 0-511 of 720), waves 4-7 two and one (the rest; slots 2, 3 and 5 are empty for them).  The SIMD serves
    // its older wave first: with an even split waves 4-7 took twice as long over their staging as waves 0-3 (7.2 k against 3.6 k clocks of a
    // 23 k-clock G tile), and waves 0-3 waited 3.8 k clocks at the tile's barrier (device stamps, profiles/r05_gru_uneven_staging.txt).
    const bool front = wave < 4;                     // wave-uniform
    const int nxp = front ? 4 : 2, nhp = front ? 2 : 1;
    auto slot_live = [&](int i) { return i < 4 ? i < nxp : i < 6 ? i - 4 < nhp : true; };
    int prc[7], loff[7], lx[4], lh[2], lg[2], lc[PHASE == 0 ? 2 : 1], so1[PHASE == 0 ? 2 : 1];
    int lhs, lgs, lcs = 0, sos = 0;
    unsigned ownbits = 0;                            // bit i: slot i's position is one of the tile's own pixels
    const int t4 = tid & 255;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        int pos;
        if (i < 4) {
            int f = front ? t4 + 256 * i : 1024 + t4 + 256 * (i & 1);
            if (f >= FNPOS * 8) f -= 256;
            pos = f >> 3; loff[i] = pos * S + 4 * q8;
        } else if (i < 6) {
            int f = front ? t4 + 256 * (i - 4) : 512 + t4;
            if (f >= FNPOS * 4) f -= 256;
            pos = f >> 2; loff[i] = pos * S + 32 + 4 * q4;
        } else { pos = spos; loff[i] = pos * S + (PHASE == 0 ? (lo ? 48 : 52) : 64); }
        const int r = pos / FPW, c = pos - r * FPW;
        const int ppix = r * a.W + c;
        prc[i] = ((r - 1) & 0xffff) | ((c - 1) << 16);         // row / column relative to the tile's first pixel (-1 .. 8 / -1 .. 16), two 16-bit fields
        if (r >= 1 && r <= FTH && c >= 1 && c <= FTW) ownbits |= 1u << i;
        if (i < 4) lx[i] = (int)a.x + ppix * 128 + 16 * q8;
        else if (i < 6) {
            lh[i - 4] = (int)a.cell[0].h + ppix * 64 + 16 * q4;
            // G: update gate = channels [16,32) of the previous plane's gates; C: reset gate = channels [0,16) of this plane's
            lg[i - 4] = (int)a.cell[0].g + ppix * 128 + 16 * q4 + (PHASE == 0 ? 64 : 0);
            if (PHASE == 0) { lc[i - 4] = (int)a.cell[0].c + ppix * 64 + 16 * q4; so1[i - 4] = (int)a.cell[0].h_out + ppix * 64 + 16 * q4; }
        } else {
            lhs = (int)(lo ? a.cell[1].h : a.cell[2].h) + ppix * (4 * FS);       // s3: two floats of this pixel, two of the next (unused)
            lgs = (int)(lo ? a.cell[1].g : a.cell[2].g) + ppix * (8 * FS) + (PHASE == 0 ? 4 * FS : 0);
            if (PHASE == 0) { lcs = (int)(lo ? a.cell[1].c : a.cell[2].c) + ppix * (4 * FS); sos = (int)(lo ? a.cell[1].h_out : a.cell[2].h_out) + ppix * (4 * FS); }
        }
    }
    float4 pre[7], preg[3], prec[PHASE == 0 ? 3 : 1];
    unsigned inside = 0;                             // bit i: piece i's position of the tile in flight lies inside the image
    auto load_piece = [&](int i, int tile) __attribute__((always_inline)) {
        const int tg = tile < end ? tile : 0;        // past the end: a harmless reload of tile 0
        const int th = tg / a.tiles_w, h0 = th * FTH, w0 = (tg - th * a.tiles_w) * FTW;
        const int bp = (h0 - 1) * a.W + (w0 - 1);    // pixel index of the staged window's origin (scalar; may be negative)
        const int rr = (prc[i] << 16) >> 16, cc = prc[i] >> 16;
        const bool in = (unsigned)(h0 + rr) < (unsigned)a.H && (unsigned)(w0 + cc) < (unsigned)a.W;      // SAME padding: zeros outside
        if (i < 4) pre[i] = ld_b128(rs, in ? lx[i] + bp * 128 : FBAD);
        else if (i < 6) {
            pre[i] = ld_b128(rs, in ? lh[i - 4] + bp * 64 : FBAD);
            preg[i - 4] = ld_b128(rs, in ? lg[i - 4] + bp * 128 : FBAD);
            if (PHASE == 0) prec[i - 4] = ld_b128(rs, in ? lc[i - 4] + bp * 64 : FBAD);
        } else {
            pre[6] = ld_b128(rs, in ? lhs + bp * (4 * FS) : FBAD);
            preg[2] = ld_b128(rs, in ? lgs + bp * (8 * FS) : FBAD);
            if (PHASE == 0) prec[2] = ld_b128(rs, in ? lcs + bp * (4 * FS) : FBAD);
        }
        if (PHASE == 0 && i >= 4) inside = in ? inside | (1u << i) : inside & ~(1u << i);
    };
    // One channel of the blend u*h + (1-u)*tanh(y), u = sigmoid(gate) (convgru.py:98,102,114-120) as ONE division:
    //   A = e^-gate, B = e^-2|y|:   u = 1/(1+A),  tanh(y) = sgn(y) (1-B)/(1+B)   =>   h' = (h (1+B) + A sgn(y) (1-B)) / ((1+A)(1+B))
    // with the exponents taken straight from the pre-scaled LayerNorm affines (ga = -scale*log2(e), ...): two v_exp_f32 and one
    // v_rcp_f32 per channel (the separate sigmoid and tanh of rounds 3-4: two of each).  A is capped at 2^64 (u < 6e-20 there) so
    // that the product of the denominators stays finite.
    auto blend1ch = [](float h, float g, float c, float ga, float gb, float ca, float cb) __attribute__((always_inline)) -> float {
        const float A = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf(g, ga, gb), 64.0f));
        const float ty = __builtin_fmaf(c, ca, cb);                  // = 2 log2(e) * y
        const float B = __builtin_amdgcn_exp2f(-fabsf(ty));
        const float p2 = 1.0f + B;
        const float rd = __builtin_amdgcn_rcpf((1.0f + A) * p2);
        const float ynum = copysignf(1.0f - B, ty);
        return __builtin_fmaf(A, ynum, h * p2) * rd;
    };
    auto stage_piece = [&](int i, float* buf, float* mbuf, int tile_of) __attribute__((always_inline)) {
        float4 v = pre[i];                           // zeros outside the image (SAME padding)
        if (i < 4) { *(float4*)(buf + loff[i]) = v; return; }
        const int gi = i < 6 ? i - 4 : 2;
        const float4 gq = preg[gi];
        if (PHASE == 0) {
            // the state entering this cell's plane: the blend of the plane before; evaluated on every path (first plane of a
            // cell: the loaded zeros are kept)
            const bool bl = i < 6 ? blend1 : blend_s;
            const float4 cq = prec[gi];
            const int uq = i < 6 ? q4 : (lo ? 8 : 10), oq = i < 6 ? 4 + q4 : (lo ? 9 : 11);
            const float4 ua = *(const float4*)lnS[uq], ub = *(const float4*)lnT[uq], ca = *(const float4*)lnS[oq], cb = *(const float4*)lnT[oq];
            float4 b;
            b.x = blend1ch(v.x, gq.x, cq.x, ua.x, ub.x, ca.x, cb.x); b.y = blend1ch(v.y, gq.y, cq.y, ua.y, ub.y, ca.y, cb.y);
            b.z = blend1ch(v.z, gq.z, cq.z, ua.z, ub.z, ca.z, cb.z); b.w = blend1ch(v.w, gq.w, cq.w, ua.w, ub.w, ca.w, cb.w);
            const bool in = (inside >> i) & 1u;
            if (STEADY || bl) v = in ? b : make_float4(0.f, 0.f, 0.f, 0.f);
            // the tile's own pixels keep the state: the output launch, the next cell and the next plane read it
            const bool own = (STEADY || bl) && tile_of < end && ((ownbits >> i) & 1u) && in;
            const int th = tile_of / a.tiles_w, h0 = th * FTH, w0 = (tile_of - th * a.tiles_w) * FTW;
            const int bp = (h0 - 1) * a.W + (w0 - 1);
            if (i < 6) st_b128(rs, own ? so1[i - 4] + bp * 64 : FBAD, v.x, v.y, v.z, v.w);
            else {
                st_b128(rs, own && lo ? sos + bp * 16 : FBAD, v.x, v.y, v.z, v.w);
                st_b64(rs, own && !lo ? sos + bp * 8 : FBAD, v.x, v.y);
            }
            *(float4*)(buf + loff[i]) = v;           // (s3: floats 54, 55 of the position receive two unused values)
        } else {
            // xb = sigmoid(LN(g_r)) * h (convgru.py:97,101,107) = h / (1 + e^-gate); the next cell's xa is the state itself
            const int rq = i < 6 ? q4 : (lo ? 4 : 5);
            const float4 ra = *(const float4*)lnS[rq], rb = *(const float4*)lnT[rq];
            float4 rv;
            rv.x = v.x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gq.x, ra.x, rb.x)));
            rv.y = v.y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gq.y, ra.y, rb.y)));
            rv.z = v.z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gq.z, ra.z, rb.z)));
            rv.w = v.w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(gq.w, ra.w, rb.w)));
            if (i < 6) { *(float4*)(buf + loff[i]) = rv; *(float4*)(buf + loff[i] + 16) = v; }
            else {       // s2: r2*s2 at 64, s2 at 68;  s3: r3*s3 into the side slab, the state itself is nobody's operand
                float* da = lo ? buf + loff[6] : mbuf + 4 * spos;
                float* db = lo ? buf + loff[6] + 4 : mini + 2 * FNPOS * 4 + 4 * spos;
                *(float4*)da = rv; *(float4*)db = v;
            }
        }
    };

    // ---- prologue: everything that comes from memory is requested before anything is waited for ------------------------------
    constexpr int NAFF = PHASE == 0 ? 44 : 22;
    double ln_s0 = 0.0, ln_s1 = 1.0; float ln_g = 0.f, ln_b = 0.f; int ln_quad = 0, ln_sub = 0, ln_kind = 0; double ln_cnt = 1.0;
    if (tid < NAFF) {   // LayerNorm sums and parameters of (cell, gate, channel)
        int k, idx;
        if (PHASE == 0) { k = tid < 32 ? 0 : tid < 40 ? 1 : 2; idx = tid - (k == 0 ? 0 : k == 1 ? 32 : 40); }
        else { k = tid < 16 ? 0 : tid < 20 ? 1 : 2; idx = tid - (k == 0 ? 0 : k == 1 ? 16 : 20); }
        const int F = k == 0 ? 16 : k == 1 ? 4 : 2;
        const int kind = idx / F, f = idx - kind * F;                // G: 0 update gate, 1 candidate;  C: 0 reset gate
        const double* st = vp(cell_sel(k, [](const FusedCell& c_) { return c_.st_in; })) + (PHASE == 0 ? (kind == 0 ? 2 : 4) : 0);
        const float* gp = kind == 0 ? cell_sel(k, [](const FusedCell& c_) { return c_.ga; }) : cell_sel(k, [](const FusedCell& c_) { return c_.oa; });
        const float* bp = kind == 0 ? cell_sel(k, [](const FusedCell& c_) { return c_.gb; }) : cell_sel(k, [](const FusedCell& c_) { return c_.ob; });
        ln_s0 = 0.0; ln_s1 = 0.0;
#pragma unroll
        for (int sl = 0; sl < GRU_FUSED_SLOTS; ++sl) { ln_s0 += st[GRU_FUSED_SLOT_STRIDE * sl]; ln_s1 += st[GRU_FUSED_SLOT_STRIDE * sl + 1]; }
        ln_g = gp[f]; ln_b = bp[f];
        ln_cnt = (double)a.H * a.W * F;
        if (PHASE == 0) ln_quad = k == 0 ? 4 * kind + (f >> 2) : k == 1 ? 8 + kind : 10 + kind;
        else ln_quad = k == 0 ? (f >> 2) : k == 1 ? 4 : 5;
        ln_sub = f & 3; ln_kind = kind;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) if (slot_live(i)) load_piece(i, first);
    // prepared weights: all loads of a thread in flight at once (one round trip instead of three); they are written to LDS AFTER
    // the first tile has been staged -- its loads were issued before these and return first, so the blend arithmetic of tile 0
    // runs while the weights are still on their way
    constexpr int N4 = W1_FLOATS / 4, NS4 = WS_FLOATS / 4;
    constexpr int K1 = (N4 + FNT - 1) / FNT, K2 = (NS4 + FNT - 1) / FNT;
    f32x4 wt1[K1], wt2[K2];
    {
        const f32x4* s4 = reinterpret_cast<const f32x4*>(a.w1);
        const f32x4* t4 = reinterpret_cast<const f32x4*>(a.wsmall);
#pragma unroll
        for (int k = 0; k < K1; ++k) wt1[k] = s4[min(tid + FNT * k, N4 - 1)];
#pragma unroll
        for (int k = 0; k < K2; ++k) wt2[k] = t4[min(tid + FNT * k, NS4 - 1)];
    }
    if (tid < NAFF) {
        // scale = gamma / sqrt(var + 1e-12), shift = beta - mean * scale in float64 (tf.contrib.layers.layer_norm), then folded
        // with the constant of the exponential that consumes it: gates -> e^-x = 2^(-log2(e) x), candidate -> e^-2|y| = 2^-|2 log2(e) y|
        const double mean = ln_s0 / ln_cnt;
        double var = ln_s1 / ln_cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = (double)ln_g / sqrt(var + 1e-12);
        const double shift = (double)ln_b - mean * inv;
        const double L2E = 1.4426950408889634;
        const double f = (PHASE == 0 && ln_kind == 1) ? 2.0 * L2E : -L2E;
        lnS[ln_quad][ln_sub] = (float)(inv * f); lnT[ln_quad][ln_sub] = (float)(shift * f);
    }
    if (tid >= 64 && tid < 64 + 8) {                 // the unused halves of the 2-channel quads
        const int q = PHASE == 0 ? 10 + ((tid - 64) >> 2) : 5, sub = 2 + ((tid - 64) & 1);
        lnS[q][sub] = 0.f; lnT[q][sub] = 0.f;
    }
    const int b_off = (wave * FPW + n) * S + 4 * kq;
    const int a_off = (kq * COUT + n) * 4;
    float bias4[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 4; ++k) bias4[m][k] = a.cell[0].bias[m * 16 + 4 * kq + k];
    // the small job of this wave (see the header): pixel = row 4 * half + (lane >> 4), column lane & 15; its bias quad
    const int half = wave & 1, srow = 4 * half + (lane >> 4), scol = lane & 15;
    const int soff = (srow * FPW + scol) * S;        // float offset of the pixel's window origin in a slab
    float sbias[4] = {0.f, 0.f, 0.f, 0.f};
    {
        const float* bp = nullptr; int nb = 0, b0 = 0;
        if (PHASE == 0) { if (wave < 4) { bp = a.cell[1].bias; b0 = 4 * (wave >> 1); nb = 4; } else if (wave < 6) { bp = a.cell[2].bias; nb = 4; } }
        else { if (wave < 2) { bp = a.cell[1].bias; nb = 4; } else if (wave < 4) { bp = a.cell[2].bias; nb = 2; } }
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nb) sbias[e] = bp[b0 + e];
    }
    // where this wave's results go: cell 1's row (8 floats per lane of the gates, 4 of the candidate), the small job's quad
    const int sy1 = (int)a.cell[0].y + ((wave * a.W + n) * COUT + 4 * kq) * 4;
    int sy_small = FBAD, small_px = 0; bool small_live = false, small_b64 = false;
    if (PHASE == 0) {
        if (wave < 4) { sy_small = (int)a.cell[1].y + (srow * a.W + scol) * 32 + 16 * (wave >> 1); small_px = 32; small_live = live2; }
        else if (wave < 6) { sy_small = (int)a.cell[2].y + (srow * a.W + scol) * 16; small_px = 16; small_live = live3; }
    } else {
        if (wave < 2) { sy_small = (int)a.cell[1].y + (srow * a.W + scol) * 16; small_px = 16; small_live = live2; }
        else if (wave < 4) { sy_small = (int)a.cell[2].y + (srow * a.W + scol) * 8; small_px = 8; small_live = live3; small_b64 = true; }
    }
    const bool wta_wave = PHASE == 0 && wave >= 6 && (STEADY || a.wta);
    const int swta = (srow * a.W + scol) * 4;
    __syncthreads();
    if (a.trace) tpa = wall_clock64();             // affines in LDS, everybody through the first barrier
#pragma unroll
    for (int i = 0; i < 7; ++i) if (slot_live(i)) stage_piece(i, slab, mini, first);
    if (a.trace) tpb = wall_clock64();             // first tile staged (this wave)
#pragma unroll
    for (int k = 0; k < K1; ++k) reinterpret_cast<f32x4*>(wl)[min(tid + FNT * k, N4 - 1)] = wt1[k];       // (the clamped lanes rewrite the last quad)
#pragma unroll
    for (int k = 0; k < K2; ++k) reinterpret_cast<f32x4*>(wsm)[min(tid + FNT * k, NS4 - 1)] = wt2[k];
#pragma unroll
    for (int i = 0; i < 7; ++i) if (slot_live(i)) load_piece(i, first + stride);
    if (a.trace) tpc = wall_clock64();             // weights in LDS, second tile requested (this wave)
    __syncthreads();

    // LayerNorm moments: float within a tile (fixed lane -> pixel map), float64 per lane across the tiles of a workgroup (the
    // sums do not depend on which workgroup swept which tile: a batch of views gives the single view's bits)
    double st_s[MT], st_q[MT], sm_s[2] = {0.0, 0.0}, sm_q[2] = {0.0, 0.0};
#pragma unroll
    for (int m = 0; m < MT; ++m) { st_s[m] = 0.0; st_q[m] = 0.0; }

    if (a.trace) tr1 = wall_clock64();               // first tile staged, second requested
    int it = 0;
    for (int tile = first; tile < end; tile += stride, ++it) {
        const float* cur = slab + (it & 1) * SLAB;
        float* nxt = slab + ((it + 1) & 1) * SLAB;
        const float* mcur = mini + (it & 1) * FNPOS * 4;
        float* mnxt = mini + ((it + 1) & 1) * FNPOS * 4;
        const int th = tile / a.tiles_w, h0 = th * FTH, w0 = (tile - th * a.tiles_w) * FTW;
        const int tp = h0 * a.W + w0;                // the tile's first pixel
        const bool svalid = h0 + srow < a.H && w0 + scol < a.W;
        // the winner-take-all accumulators of this tile's pixels, requested now (waves 6, 7 of G; out of range elsewhere)
        float wta_mp = 0.f, wta_es = 0.f;
        if (PHASE == 0) {
            const int wo = wta_wave && svalid ? swta + tp * 4 : FBAD;
            wta_mp = ld_b32(rs, wo == FBAD ? FBAD : wo + (int)a.max_prob);
            wta_es = ld_b32(rs, wo == FBAD ? FBAD : wo + (int)a.exp_sum);
        }

        // ---- cell 1: x channels and state channels in separate accumulators, combined as (h part) + ((x part) + bias): what
        // both formulations of gru_mfma.hip compute
        f32x4 acc[MT], accx[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; accx[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        {
            constexpr int NG = 27;
            f32x4 bv[2], av[2][MT];
            auto load_grp = [&](int g, f32x4& b, f32x4 (&aop)[MT]) __attribute__((always_inline)) {
                const int tap = g / 3, s = g % 3;
                const int kh = tap / 3, kw = tap % 3;
                b = *(const f32x4*)(cur + b_off + (kh * FPW + kw) * S + 16 * s);
#pragma unroll
                for (int m = 0; m < MT; ++m) aop[m] = *(const f32x4*)(wl + a_off + m * 64 + (tap * CQ + 4 * s) * WROW);
            };
            load_grp(0, bv[0], av[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_grp(g + 1, bv[(g + 1) & 1], av[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    if ((i * (NG / 2)) / 7 == g && slot_live(i)) stage_piece(i, nxt, mnxt, tile + stride);
                    if (NG / 2 + (i * (NG - NG / 2)) / 7 == g && slot_live(i)) load_piece(i, tile + 2 * stride);
                }
                const bool xgroup = (g % 3) < 2;     // compile-time after unrolling
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (xgroup) accx[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][jj], bv[g & 1][jj], accx[m], 0, 0, 0);
                        else acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m][jj], bv[g & 1][jj], acc[m], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {   // store (+ bias) and LayerNorm moments (gates: tile 0 = reset, tile 1 = update)
            const bool ok1 = live1 && h0 + wave < a.H && w0 + n < a.W;
            const int o1 = ok1 ? sy1 + tp * (COUT * 4) : FBAD;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 r = acc[m], rx = accx[m];
                const float4 o = make_float4(r[0] + (rx[0] + bias4[m][0]), r[1] + (rx[1] + bias4[m][1]), r[2] + (rx[2] + bias4[m][2]), r[3] + (rx[3] + bias4[m][3]));
                st_b128(rs, ok1 ? o1 + m * 64 : FBAD, o.x, o.y, o.z, o.w);
                const float ts = (o.x + o.y) + (o.z + o.w), tq = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                st_s[m] += ok1 ? (double)ts : 0.0;
                st_q[m] += ok1 ? (double)tq : 0.0;
            }
        }

        // ---- this wave's small job: lane = pixel, v_mfma_f32_4x4x1 (lane m & 3 supplies the weights of output channel m & 3 of
        // the job's channel quad, the lane's own staged value is the B operand): acc = bias, then fma per (tap, input channel)
        // in the order [xa | xb] -- conv2d_small_body's chain.  LDS reads and matrix instructions only (see the header).
        auto job20 = [&](const float* tab) __attribute__((always_inline)) -> f32x4 {
            // FOUR independent chains (input channels 4q + e: one chain per e), summed at the end.  One chain of 180 instructions waits
            // on itself -- ~20 clk per 8-cycle instruction with every wave of the workgroup in its small job at the end of a tile:
            // 1.5 us of a 9.5 us G tile (profiles/r05_gru_fused_wave_roles.txt).  The sum order differs from conv2d_small_body's
            // single chain in the last bits (batch = single view still holds: the same kernel either way).
            f32x4 r4[4] = {(f32x4){sbias[0], sbias[1], sbias[2], sbias[3]}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            const float* ap = tab + (lane & 3) * 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* bt = cur + soff + ((tap / 3) * FPW + (tap % 3)) * S + XA2;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const f32x4 bq = *(const f32x4*)(bt + 4 * q);
                    const f32x4 aq = *(const f32x4*)(ap + (tap * 5 + q) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) r4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[e], bq[e], r4[e], 0, 0, 0);
                }
            }
            return (r4[0] + r4[1]) + (r4[2] + r4[3]);
        };
        auto job6 = [&](const float* tab) __attribute__((always_inline)) -> f32x4 {
            f32x4 r4[4] = {(f32x4){sbias[0], sbias[1], sbias[2], sbias[3]}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            const float* ap = tab + (lane & 3) * 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int po = ((tap / 3) * FPW + (tap % 3));
                const f32x4 b0 = *(const f32x4*)(cur + soff + po * S + XA3);
                float2 b1;
                if (PHASE == 0) b1 = *(const float2*)(cur + soff + po * S + 52);
                else b1 = *(const float2*)(mcur + 4 * (srow * FPW + scol + po));
                const f32x4 a0 = *(const f32x4*)(ap + (tap * 2) * 16), a1 = *(const f32x4*)(ap + (tap * 2 + 1) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) r4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[e], b0[e], r4[e], 0, 0, 0);
                r4[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[0], b1.x, r4[0], 0, 0, 0);
                r4[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[1], b1.y, r4[1], 0, 0, 0);
            }
            return (r4[0] + r4[1]) + (r4[2] + r4[3]);
        };
        f32x4 sr = {0.f, 0.f, 0.f, 0.f};             // the small job's four output channels of this lane's pixel
        float pr = 0.f;                              // waves 6, 7 of G: exp(prob_conv)
        if (PHASE == 0) {
            if (wave < 4) sr = job20(wsm + (wave >> 1) * T2);        // cell 2 gates: output channels 4 * grp .. + 3 (grp 0 = reset, 1 = update group)
            else if (wave < 6) sr = job6(wsm + 2 * T2);             // cell 3 gates: channels 0,1 reset | 2,3 update
            else {                                                   // prob_conv + exp (model.py:701-703)
                const float* pw = wsm + 2 * T2 + T3;
                float pacc = pw[18];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float2 b = *(const float2*)(cur + soff + ((tap / 3) * FPW + (tap % 3)) * S + 52);
                    pacc += b.x * pw[tap * 2]; pacc += b.y * pw[tap * 2 + 1];
                }
                pr = expf(pacc);
            }
        } else {
            if (wave < 2) sr = job20(wsm);           // cell 2 candidate (4 channels)
            else if (wave < 4) sr = job6(wsm + T2);  // cell 3 candidate (2 channels; rows 2, 3 of the table are zero)
        }
        {   // the small job's store and LayerNorm moments, the winner-take-all update: straight-line for every wave
            const bool sok = small_live && svalid;
            const int os = sok ? sy_small + tp * small_px : FBAD;
            st_b128(rs, small_b64 ? FBAD : os, sr[0], sr[1], sr[2], sr[3]);
            if (PHASE == 1) st_b64(rs, small_b64 ? os : FBAD, sr[0], sr[1]);
            // moments: group 0 = all four channels (cell 2, one group per wave; C: the candidate) or channels 0,1 (cell 3);
            // group 1 = channels 2,3 (cell 3 gates: the update group)
            const bool pair = PHASE == 0 ? (wave >= 4) : (wave >= 2);      // cell 3: channel pairs
            const float s01 = sr[0] + sr[1], q01 = __builtin_fmaf(sr[1], sr[1], sr[0] * sr[0]);
            const float s23 = sr[2] + sr[3], q23 = __builtin_fmaf(sr[3], sr[3], sr[2] * sr[2]);
            // (the four-channel group in the order of conv2d_small_body: s = ((r0 + r1) + r2) + r3, q = fma chain)
            const float s4 = (s01 + sr[2]) + sr[3], q4s = __builtin_fmaf(sr[3], sr[3], __builtin_fmaf(sr[2], sr[2], q01));
            const float s0 = pair ? s01 : s4, q0 = pair ? q01 : q4s;
            sm_s[0] += sok ? (double)s0 : 0.0; sm_q[0] += sok ? (double)q0 : 0.0;
            sm_s[1] += sok && pair ? (double)s23 : 0.0; sm_q[1] += sok && pair ? (double)q23 : 0.0;
            if (PHASE == 0) {   // winner-take-all update (model.py:721-731; strict '<' keeps the first maximum)
                const int wo = wta_wave && svalid ? swta + tp * 4 : FBAD;
                const bool better = wo != FBAD && wta_mp < pr;
                st_b32(rs, better ? wo + (int)a.max_prob : FBAD, pr);
                st_b32(rs, better ? wo + (int)a.depth_image : FBAD, dv.v[view]);
                st_b32(rs, wo == FBAD ? FBAD : wo + (int)a.exp_sum, wta_es + pr);
            }
        }
        __syncthreads();
        if (a.trace && it == 0) tr2 = wall_clock64();   // first tile done
    }
    if (a.trace) tr3 = wall_clock64();

    // ---- LayerNorm sums of this workgroup -> float64 atomics.  red[wave][0..3] : cell 1 (MT groups x 2), [4..7] the small job (2 x 2)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const double s = wave_sum_dpp63(st_s[m]), q = wave_sum_dpp63(st_q[m]);
        if (lane == 63) { red[wave][2 * m] = s; red[wave][2 * m + 1] = q; }
    }
    {
        const double s0 = wave_sum_dpp63(sm_s[0]), q0 = wave_sum_dpp63(sm_q[0]), s1 = wave_sum_dpp63(sm_s[1]), q1 = wave_sum_dpp63(sm_q[1]);
        if (lane == 63) { red[wave][4] = s0; red[wave][5] = q0; red[wave][6] = s1; red[wave][7] = q1; }
    }
    __syncthreads();
    const int so = GRU_FUSED_SLOT_STRIDE * (blockIdx.x % GRU_FUSED_SLOTS);      // this workgroup's copy of the row
    if (PHASE == 0) {
        // cell 1: [0..3] = reset s,q | update s,q.  cell 2: reset = waves 0,1 ; update = waves 2,3.  cell 3: waves 4,5 (both groups)
        if (tid < 4 && live1) { double t = 0.0; for (int w = 0; w < 8; ++w) t += red[w][tid]; atomicAdd(&vp(a.cell[0].st_out)[so + tid], t); }
        else if (tid >= 4 && tid < 8 && live2) { const int e = tid - 4, w0 = (e >> 1) * 2; atomicAdd(&vp(a.cell[1].st_out)[so + e], red[w0][4 + (e & 1)] + red[w0 + 1][4 + (e & 1)]); }
        else if (tid >= 8 && tid < 12 && live3) { const int e = tid - 8; atomicAdd(&vp(a.cell[2].st_out)[so + e], red[4][4 + e] + red[5][4 + e]); }
    } else {
        if (tid < 2 && live1) { double t = 0.0; for (int w = 0; w < 8; ++w) t += red[w][tid]; atomicAdd(&vp(a.cell[0].st_out)[so + 4 + tid], t); }
        else if (tid >= 2 && tid < 4 && live2) { const int e = tid - 2; atomicAdd(&vp(a.cell[1].st_out)[so + 4 + e], red[0][4 + e] + red[1][4 + e]); }
        else if (tid >= 4 && tid < 6 && live3) { const int e = tid - 4; atomicAdd(&vp(a.cell[2].st_out)[so + 4 + e], red[2][4 + e] + red[3][4 + e]); }
    }
    if (a.trace && tid == 0) {
        const long long slot = (long long)atomicAdd((unsigned long long*)a.trace, 1ULL);
        if (slot < a.trace_cap) {
            long long* r = a.trace + 1 + slot * 8;
            r[0] = ((long long)a.launch_id << 32) | (PHASE << 16) | blockIdx.x; r[1] = tr0; r[2] = tr1; r[3] = tr2; r[4] = tr3; r[5] = wall_clock64();
            r[6] = it | ((tpa - tr0) << 16) | ((tpb - tr0) << 32) | ((tpc - tr0) << 48); r[7] = clock64() - cy0;      // tiles swept + three prologue stamps (ticks since entry, 16 bits each); shader clocks between entry and exit
        }
    }
}


// small-cell weight tables: TensorFlow kernel (3,3,CT,CO) -> out[tap][quad][m][4]: weight of input channel 4*quad + e, output
// channel co0 + m (zero outside either range)
__global__ void gru_small_table_kernel(const float* __restrict__ w, int CT, int CO, int co0, int nquad, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * nquad * 16) return;
    const int e = i & 3, m = (i >> 2) & 3, q = (i >> 4) % nquad, tap = (i >> 4) / nquad;
    const int ci = 4 * q + e, co = co0 + m;
    out[i] = (ci < CT && co < CO) ? w[((size_t)tap * CT + ci) * CO + co] : 0.f;
}

template <int PHASE, bool STEADY>
int launch_fused2(const FusedArgs& a, const FusedDepth& dv, int grid, size_t smem, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gru_fused_kernel<PHASE, STEADY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    // (the two-group form of this launch -- exact, 3 % slower -- is profiles/r05_gru_fused_two_groups.patch, not product code)
    gru_fused_kernel<PHASE, STEADY><<<grid, FNT, smem, st>>>(a, dv);
    return (int)hipGetLastError();
}
template <int PHASE>
int launch_fused(const FusedArgs& a0, const FusedDepth& dv, int views, bool steady, hipStream_t st) {
    FusedArgs a = a0;
    a.tiles_h = (a.H + FTH - 1) / FTH;
    a.tiles_w = (a.W + FTW - 1) / FTW;
    const int tiles = a.tiles_h * a.tiles_w;
    int per = 256 / views;
    if (per >= 8) per &= ~7;                          // a multiple of 8: every XCD gets its own band of tiles
    if (per < 1) per = 1;
    if (per > tiles) per = tiles;
    a.wg_per_view = per;
    constexpr int S = PHASE == 0 ? 56 : 72;
    constexpr int COUT = PHASE == 0 ? 32 : 16;
    constexpr int WS_FLOATS = PHASE == 0 ? 2 * 720 + 288 + 20 : 720 + 288;
    const size_t smem = (size_t)(9 * 48 * COUT + WS_FLOATS + 2 * FNPOS * S + (PHASE == 1 ? 3 * FNPOS * 4 : 0)) * sizeof(float);
    return steady ? launch_fused2<PHASE, true>(a, dv, per * views, smem, st) : launch_fused2<PHASE, false>(a, dv, per * views, smem, st);
}

}  // namespace

// ---- host side of the fused sweep (called from mvs_gru_wta_batch_f32, gru.hip) -------------------------------------------
// Workspace of one view as the fused sweep sees it (carved by gru.hip): S[k][2] state ping-pong, G[k][2] gate ping-pong, Cb[k]
// candidate, stats (a ring of planes x GRU_FUSED_SLOTS copies x 3 cells x 6 doubles), x (a batch of XB cost slices).
struct GruFusedWs {
    char* base;                              // view 0's workspace block (everything below lies inside it)
    float* x; float* S[3][2]; float* G[3][2]; float* Cb[3]; double* stats;
    float *max_prob, *depth, *exp_sum;
    float *w1g, *w1c, *wsg, *wsc;            // prepared weights (shared by the views; view 0's block)
};
constexpr int GRU_FUSED_RING = 64;           // LayerNorm-sum rows: plane p uses row p % 64 (gru.hip zeroes them a batch ahead)
constexpr int GRU_FUSED_ROW = GRU_FUSED_SLOT_STRIDE * GRU_FUSED_SLOTS;      // doubles per row: [slot][cell][6] (the same constants as in gru.hip)

namespace {
__global__ void gru_prob_table_kernel(const float* __restrict__ pw, const float* __restrict__ pb, float* __restrict__ out) {
    const int i = threadIdx.x;                   // prob_conv (3,3,2,1): 18 weights, the bias, a pad float
    if (i < 20) out[i] = i < 18 ? pw[i] : (i == 18 && pb ? pb[0] : 0.f);
}
}  // namespace

int mvs_gru_fused_prepare_weights(const float* const* params, const GruFusedWs& ws, hipStream_t st) {
    // small tables.  G: cell 2 gates (20 -> 8) as two output-channel quads, cell 3 gates (6 -> 4), prob_conv.  C: cell 2 candidate
    // (20 -> 4), cell 3 candidate (6 -> 2).
    auto tab = [&](const float* w, int CT, int CO, int co0, int nquad, float* out) {
        gru_small_table_kernel<<<mvs_cdiv(9 * nquad * 16, 256), 256, 0, st>>>(w, CT, CO, co0, nquad, out);
    };
    tab(params[10], 20, 8, 0, 5, ws.wsg); tab(params[10], 20, 8, 4, 5, ws.wsg + 720); tab(params[20], 6, 4, 0, 2, ws.wsg + 1440);
    gru_prob_table_kernel<<<1, 64, 0, st>>>(params[30], params[31], ws.wsg + 1728);
    tab(params[16], 20, 4, 0, 5, ws.wsc); tab(params[26], 6, 2, 0, 2, ws.wsc + 720);
    return (int)hipGetLastError();
}

// Diagnostic: per-workgroup time stamps of the fused launches (100 MHz wall clock: entry, first tile staged, first tile done, loop
// done, exit) appended to a caller-owned device buffer of 1 + 8 * capacity int64 (tools/gru_fused_trace.py).  Off (null) by default.
static long long* g_fused_trace = nullptr; static int g_fused_trace_cap = 0; static int g_fused_launch = 0;
extern "C" int mvs_gru_fused_trace(void* buffer, int capacity) {
    g_fused_trace = (long long*)buffer; g_fused_trace_cap = buffer ? capacity : 0; g_fused_launch = 0;
    return 0;
}

// one plane step of the pipeline: G(t) then C(t).  t runs 0 .. depth_num + 2.
int mvs_gru_fused_step(const GruFusedWs& ws, const float* const* params, int t, int depth_num, const float* x_t, int H, int W,
                       int views, size_t vstride, const float* depth_values /* host, (views, depth_num) */, hipStream_t st) {
    if (vstride >= ((size_t)1 << 31)) return MVS_E_SHAPE;            // the kernels address a view's block with 32-bit byte offsets
    auto off = [&](const void* p) { return (unsigned)((const char*)p - ws.base); };
    FusedArgs g = {}, c = {};
    g.x = c.x = off(x_t);
    g.ws = c.ws = ws.base;
    g.H = c.H = H; g.W = c.W = W; g.vstride = c.vstride = vstride;
    bool steady = true;                              // every cell live and blending, the WTA update on
    for (int k = 0; k < 3; ++k) {
        const int p = t - k;                         // this cell's plane
        const float* const* pp = params + 10 * k;
        const bool conv = p >= 0 && p < depth_num, blend = p >= 1 && p <= depth_num;
        steady = steady && conv && blend;
        const int pc = p < 0 ? 0 : p, pm = p < 1 ? 0 : p - 1;        // clamped plane indices for the stats rows of dead cells
        FusedCell& gc = g.cell[k];
        // s(q) lives in S[k][q & 1] (s(-1) = 0 in S[k][1]); G forms s(p-1) from s(p-2)
        gc.h = off(blend ? ws.S[k][p & 1] : ws.S[k][(p - 1) & 1]);
        gc.c = off(ws.Cb[k]); gc.g = off(ws.G[k][(p - 1) & 1]);
        gc.st_in = ws.stats + (size_t)(pm % GRU_FUSED_RING) * GRU_FUSED_ROW + k * 6;
        gc.h_out = off(ws.S[k][(p - 1) & 1]);
        gc.y = off(ws.G[k][p & 1]);
        gc.st_out = ws.stats + (size_t)(pc % GRU_FUSED_RING) * GRU_FUSED_ROW + k * 6;
        gc.bias = pp[1]; gc.ga = pp[4]; gc.gb = pp[5]; gc.oa = pp[8]; gc.ob = pp[9];
        gc.conv = conv; gc.blend = blend;
        FusedCell& cc = c.cell[k];
        cc.h = off(ws.S[k][(p - 1) & 1]); cc.c = 0; cc.g = off(ws.G[k][p & 1]);
        cc.st_in = ws.stats + (size_t)(pc % GRU_FUSED_RING) * GRU_FUSED_ROW + k * 6;
        cc.h_out = 0; cc.y = off(ws.Cb[k]); cc.st_out = ws.stats + (size_t)(pc % GRU_FUSED_RING) * GRU_FUSED_ROW + k * 6;
        cc.bias = pp[7]; cc.ga = pp[2]; cc.gb = pp[3]; cc.oa = pp[2]; cc.ob = pp[3];
        cc.conv = conv; cc.blend = 0;
    }
    g.w1 = ws.w1g; g.wsmall = ws.wsg; c.w1 = ws.w1c; c.wsmall = ws.wsc;
    g.max_prob = off(ws.max_prob); g.depth_image = off(ws.depth); g.exp_sum = off(ws.exp_sum);
    const int q = t - 3;
    g.wta = q >= 0 && q < depth_num;
    steady = steady && g.wta;
    FusedDepth dv = {};
    for (int v = 0; v < views && v < FMAXV; ++v) dv.v[v] = g.wta ? depth_values[(size_t)v * depth_num + q] : 0.f;
    g.trace = c.trace = g_fused_trace; g.trace_cap = c.trace_cap = g_fused_trace_cap;
    g.launch_id = g_fused_launch++; c.launch_id = g_fused_launch++;
    int rc = launch_fused<0>(g, dv, views, steady, st);
    if (rc) return rc;
    if (t > depth_num + 1) return 0;                 // the last output launch is C(depth_num + 1): cell 3's plane depth_num - 1
    return launch_fused<1>(c, dv, views, steady, st);
}
