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

namespace {

constexpr int FMAXV = 8;                 // reference views per launch (mvs_gru_wta_batch_f32)
constexpr int FNT = 512, FTH = 8, FTW = 16, FPW = FTW + 2, FNPOS = (FTH + 2) * FPW;      // 180 staged positions per tile

struct FusedCell {
    const float* h;         // G: the state BEFORE the blend (entered plane p-1), or the state itself when !blend;  C: the state s(p-1)
    const float* c;         // G: raw candidate convolution of plane p-1 (H,W,F)
    const float* g;         // G: raw gate convolution of plane p-1 (update half read);  C: of plane p (reset half read)   (H,W,2F)
    const double* st_in;    // 6 doubles [reset s,q | update s,q | candidate s,q]: G of plane p-1, C of plane p
    double* st_out;         // 6 doubles of plane p: G adds [0..3], C adds [4..5]
    const float* bias;
    const float *ga, *gb, *oa, *ob;      // G: update gamma / beta, candidate gamma / beta;  C: reset gamma / beta (ga, gb)
    unsigned h_out;         // G: byte offset (in the view's workspace block) of the tensor that receives s(p-1) on the tile's own pixels
    unsigned y;             // byte offset of the output: G raw gates of plane p (H,W,2F);  C raw candidate of plane p (H,W,F)
    int conv, blend;        // this cell's convolution is live (its plane exists) / the blend of plane p-1 is formed on load
};
struct FusedArgs {
    const float* x;         // (H,W,32) cost slice of cell 1's plane
    FusedCell cell[3];
    const float* w1;        // cell-1 weights of this phase, [tap9][12][COUT][4] (gru_weight_slice_kernel)
    const float* wsmall;    // small-cell tables of this phase (gru_small_table_kernel)
    char* ws;               // view 0's workspace block: every tensor this kernel WRITES lives in it (one buffer resource per view)
    unsigned max_prob, depth_image, exp_sum; int wta;    // winner-take-all accumulators (G): byte offsets in the block, live flag
    int H, W, tiles_h, tiles_w, wg_per_view;
    size_t vstride;         // bytes between the workspace blocks of consecutive views (< 2^31)
};
struct FusedDepth { float v[FMAXV]; };   // depth value of the WTA plane, per view (its own kernel argument: indexed on the kernarg)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_b128(__amdgpu_buffer_rsrc_t rsrc, int voff) {
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
// Stores (and the WTA loads) address a tensor inside the view's workspace block: `toff` = the tensor's byte offset in the block,
// `voff` = the lane's byte offset in the tensor, or negative = nothing to do for this lane (the offset is pushed out of range and
// the hardware drops the lane).  The tensor offset is ADDED INTO THE VGPR OFFSET, the instruction's SGPR offset stays 0 -- on
// purpose: with a register in the soffset field the compiler's hazard recognizer assumes that the "store of more than 64 bits
// followed by a VALU write of its data registers" hazard does not exist and schedules such a write right behind the store; on
// gfx950 it does exist (first build of this file: the x component of a float4 store, overwritten by the next instruction,
// reached memory corrupted for the last lanes of each row -- run-to-run differences at 16 x 16 pixels;
// tools/store_hazard_probe.hip reproduces it in isolation).
constexpr int FBAD = (int)0x80000000;
__device__ __forceinline__ int fold(int voff, int toff) { return voff < 0 ? FBAD : voff + toff; }
__device__ __forceinline__ float ld_b32(__amdgpu_buffer_rsrc_t rsrc, int voff, int toff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, fold(voff, toff), 0, 0));
}
__device__ __forceinline__ void st_b128(__amdgpu_buffer_rsrc_t rsrc, int voff, int toff, float x, float y, float z, float w) {
    __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)}, rsrc, fold(voff, toff), 0, 0);
}
__device__ __forceinline__ void st_b64(__amdgpu_buffer_rsrc_t rsrc, int voff, int toff, float x, float y) {
    __builtin_amdgcn_raw_buffer_store_b64((u32x2_t){__float_as_uint(x), __float_as_uint(y)}, rsrc, fold(voff, toff), 0, 0);
}
__device__ __forceinline__ void st_b32(__amdgpu_buffer_rsrc_t rsrc, int voff, int toff, float x) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rsrc, fold(voff, toff), 0, 0);
}

// PHASE 0 = gates launch G(t), PHASE 1 = output launch C(t).  Slab channel map (floats per staged position):
//   G (S = 56): x 0..31 | s1 32..47 | s2 48..51 | s3 52..53 | 54,55 unused          (56 = 48 + the conflict-free pad of gru_mfma.hip)
//   C (S = 72): x 0..31 | r1*s1 32..47 | s1 48..63 | r2*s2 64..67 | s2 68..71 ;  r3*s3 in a 4-float side slab
//
// Memory operations and the wait counters.  On gfx950 loads AND stores retire in order through one counter (vmcnt), and the
// compiler can only count what it sees on every path: a store inside `if (own pixel)` makes it wait for ALL outstanding memory
// operations before the next staged piece is consumed -- i.e. for the acknowledgement of stores issued a few instructions
// earlier, between the matrix instructions of the sweep (first build of this kernel: 11.3 us per tile against 6.5 us of matrix
// time).  So every memory operation inside the tile loop is issued by every lane of every wave on a straight line: stores and
// the WTA loads go through ONE buffer resource over the view's workspace block with the lane's offset pushed out of range
// when the lane has nothing to write / read (the hardware drops such lanes; st_b128 and friends below), LDS reads never sit inside a branch either, and
// the wave-specific small jobs (branches) contain matrix / vector instructions and LDS reads only.
template <int PHASE>
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
    constexpr int BAD = (int)0x80000000;                            // a byte offset outside every buffer

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                                // cell-1 weights
    float* wsm = wl + W1_FLOATS;                     // small tables
    float* slab = wsm + WS_FLOATS;                   // [2][FNPOS][S]
    float* mini = slab + 2 * SLAB;                   // C: [2][FNPOS][4] r3*s3 (.xy), then [FNPOS][4] that absorbs the stores a wave has no use for
    // LayerNorm (scale, shift) quads: G: 0-3 s1 update, 4-7 s1 candidate, 8 s2 update, 9 s2 candidate, 10 s3 update (2), 11 s3 candidate (2)
    //                                 C: 0-3 s1 reset, 4 s2 reset, 5 s3 reset (2)
    __shared__ __attribute__((aligned(16))) float lnS[12][4], lnT[12][4];
    __shared__ double red[8][8];
    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int view = blockIdx.x / a.wg_per_view, j = blockIdx.x - view * a.wg_per_view;
    // view v's tensors live v * vstride bytes after view 0's.  (No modified copy of `a`: a struct that is written to and then
    // indexed with a runtime cell number would live in scratch memory; the kernel arguments stay in SGPRs.)
    const size_t vo = (size_t)view * a.vstride;
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

    // ---- staging pieces (global -> registers -> LDS), hung between the matrix instructions of the sweep ---------------------
    // pieces 0-2: x (1440 float4 per tile), pieces 3-4: s1 (720 channel quads), piece 5: s2 (waves 0-3) / s3 (waves 4-7), 180 each
    const int HW4 = a.H * a.W * 4;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)vp(a.x), 0, HW4 * 32, 0x00020000);
    const auto rs_h1 = __builtin_amdgcn_make_buffer_rsrc((void*)vp(a.cell[0].h), 0, HW4 * 16, 0x00020000);
    const auto rs_g1 = __builtin_amdgcn_make_buffer_rsrc((void*)vp(a.cell[0].g), 0, HW4 * 32, 0x00020000);
    const auto rs_c1 = __builtin_amdgcn_make_buffer_rsrc((void*)vp(PHASE == 0 ? a.cell[0].c : a.cell[0].h), 0, HW4 * 16, 0x00020000);
    const bool lo = wave < 4;                        // wave-uniform: this wave's small piece is s2 (else s3)
    const int FS = lo ? 4 : 2;                       // channels of the small family
    const auto rs_hs = __builtin_amdgcn_make_buffer_rsrc((void*)vp(lo ? a.cell[1].h : a.cell[2].h), 0, HW4 * FS, 0x00020000);
    const auto rs_gs = __builtin_amdgcn_make_buffer_rsrc((void*)vp(lo ? a.cell[1].g : a.cell[2].g), 0, HW4 * 2 * FS, 0x00020000);
    const auto rs_cs = __builtin_amdgcn_make_buffer_rsrc((void*)vp(PHASE == 0 ? (lo ? a.cell[1].c : a.cell[2].c) : (lo ? a.cell[1].h : a.cell[2].h)), 0, HW4 * FS, 0x00020000);
    const auto rs_ws = __builtin_amdgcn_make_buffer_rsrc((void*)vp(a.ws), 0, (int)a.vstride, 0x00020000);      // everything this kernel writes
    const int blend1 = a.cell[0].blend, blend_s = lo ? a.cell[1].blend : a.cell[2].blend;
    const int so_hout1 = (int)a.cell[0].h_out, so_hout_s = (int)(lo ? a.cell[1].h_out : a.cell[2].h_out);
    const int so_y1 = (int)a.cell[0].y;
    const int live1 = a.cell[0].conv, live2 = a.cell[1].conv, live3 = a.cell[2].conv;

    const int q8 = tid & 7, q4 = tid & 3;
    const int spos = min(tid & 255, FNPOS - 1);      // the small piece's position
    int ppix[6], prc[6], loff[6];                    // pixel offset inside the staged window, (row | col << 8), LDS float offset
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int pos;
        if (i < 3) { int f = tid + FNT * i; if (f >= FNPOS * 8) f -= FNT; pos = f >> 3; loff[i] = pos * S + 4 * q8; }
        else if (i < 5) { int f = tid + FNT * (i - 3); if (f >= FNPOS * 4) f -= FNT; pos = f >> 2; loff[i] = pos * S + 32 + 4 * q4; }
        else { pos = spos; loff[i] = pos * S + (PHASE == 0 ? (lo ? 48 : 52) : 64); }
        const int r = pos / FPW, c = pos - r * FPW;
        ppix[i] = r * a.W + c; prc[i] = r | (c << 8);
    }
    float4 pre[6], preg[3], prec[PHASE == 0 ? 3 : 1];
    unsigned inside = 0;                             // bit i: piece i's position lies inside the image (G: the blend of a padded position is 0)
    auto load_piece = [&](int i, int tile) __attribute__((always_inline)) {
        const int tg = tile < end ? tile : 0;        // past the end: a harmless reload of tile 0
        const int th = tg / a.tiles_w, h0 = th * FTH, w0 = (tg - th * a.tiles_w) * FTW;
        const int r = prc[i] & 255, c = prc[i] >> 8;
        const bool ok = (unsigned)(w0 - 1 + c) < (unsigned)a.W;      // rows above / below the image fall outside the buffers and read 0
        const int pix = (h0 - 1) * a.W + (w0 - 1) + ppix[i];          // may be negative: out of range as an unsigned byte offset
        if (i < 3) pre[i] = ld_b128(rs_x, ok ? pix * 128 + 16 * q8 : BAD);
        else if (i < 5) {
            pre[i] = ld_b128(rs_h1, ok ? pix * 64 + 16 * q4 : BAD);
            // G: update gate = channels [16,32) of the previous plane's gates; C: reset gate = channels [0,16) of this plane's
            preg[i - 3] = ld_b128(rs_g1, ok ? pix * 128 + 16 * q4 + (PHASE == 0 ? 64 : 0) : BAD);
            if (PHASE == 0) prec[i - 3] = ld_b128(rs_c1, ok ? pix * 64 + 16 * q4 : BAD);
        } else {
            pre[5] = ld_b128(rs_hs, ok ? pix * (4 * FS) : BAD);      // s3: two floats of this pixel, two of the next (unused)
            preg[2] = ld_b128(rs_gs, ok ? pix * (8 * FS) + (PHASE == 0 ? 4 * FS : 0) : BAD);
            if (PHASE == 0) prec[2] = ld_b128(rs_cs, ok ? pix * (4 * FS) : BAD);
        }
        if (PHASE == 0 && i >= 3) {
            const bool in = ok && (unsigned)(h0 - 1 + r) < (unsigned)a.H;
            inside = in ? inside | (1u << i) : inside & ~(1u << i);
        }
    };
    auto sig = [](float x) { return mvs_sigmoid_fast(x); };
    auto tanh_ = [](float x) { return mvs_tanh_fast(x); };
    auto stage_piece = [&](int i, float* buf, float* mbuf, int tile_of) __attribute__((always_inline)) {
        float4 v = pre[i];                           // zeros outside the image (SAME padding)
        if (i < 3) { *(float4*)(buf + loff[i]) = v; return; }
        const int gi = i < 5 ? i - 3 : 2;
        const float4 gq = preg[gi];
        if (PHASE == 0) {
            // the state entering this cell's plane: u*h + (1-u)*tanh(LN c) of the plane before (convgru.py:98,102,114-120);
            // evaluated on every path (first plane of a cell: the loaded zeros are kept)
            const bool bl = i < 5 ? blend1 : blend_s;
            const float4 cq = prec[gi];
            const int uq = i < 5 ? q4 : (lo ? 8 : 10), oq = i < 5 ? 4 + q4 : (lo ? 9 : 11);
            const float4 ua = *(const float4*)lnS[uq], ub = *(const float4*)lnT[uq], ca = *(const float4*)lnS[oq], cb = *(const float4*)lnT[oq];
            const float u0 = sig(gq.x * ua.x + ub.x), u1 = sig(gq.y * ua.y + ub.y), u2 = sig(gq.z * ua.z + ub.z), u3 = sig(gq.w * ua.w + ub.w);
            float4 b;
            b.x = u0 * v.x + (1.0f - u0) * tanh_(cq.x * ca.x + cb.x); b.y = u1 * v.y + (1.0f - u1) * tanh_(cq.y * ca.y + cb.y);
            b.z = u2 * v.z + (1.0f - u2) * tanh_(cq.z * ca.z + cb.z); b.w = u3 * v.w + (1.0f - u3) * tanh_(cq.w * ca.w + cb.w);
            const bool in = (inside >> i) & 1u;
            if (bl) v = in ? b : make_float4(0.f, 0.f, 0.f, 0.f);
            // the tile's own pixels keep the state: the output launch, the next cell and the next plane read it
            const int r = prc[i] & 255, c = prc[i] >> 8;
            const bool own = bl && tile_of < end && r >= 1 && r <= FTH && c >= 1 && c <= FTW && in;
            const int th = tile_of / a.tiles_w, h0 = th * FTH, w0 = (tile_of - th * a.tiles_w) * FTW;
            const int p = (h0 - 1 + r) * a.W + (w0 - 1 + c);
            if (i < 5) st_b128(rs_ws, own ? p * 64 + 16 * q4 : BAD, so_hout1, v.x, v.y, v.z, v.w);
            else {
                st_b128(rs_ws, own && lo ? p * 16 : BAD, so_hout_s, v.x, v.y, v.z, v.w);
                st_b64(rs_ws, own && !lo ? p * 8 : BAD, so_hout_s, v.x, v.y);
            }
            *(float4*)(buf + loff[i]) = v;           // (s3: floats 54, 55 of the position receive two unused values)
        } else {
            // xb = sigmoid(LN(g_r)) * h (convgru.py:97,101,107); the next cell's xa is the state itself
            const int rq = i < 5 ? q4 : (lo ? 4 : 5);
            const float4 ra = *(const float4*)lnS[rq], rb = *(const float4*)lnT[rq];
            float4 rv;
            rv.x = v.x * sig(gq.x * ra.x + rb.x); rv.y = v.y * sig(gq.y * ra.y + rb.y);
            rv.z = v.z * sig(gq.z * ra.z + rb.z); rv.w = v.w * sig(gq.w * ra.w + rb.w);
            if (i < 5) { *(float4*)(buf + loff[i]) = rv; *(float4*)(buf + loff[i] + 16) = v; }
            else {       // s2: r2*s2 at 64, s2 at 68;  s3: r3*s3 into the side slab, the state itself is nobody's operand
                float* da = lo ? buf + loff[5] : mbuf + 4 * spos;
                float* db = lo ? buf + loff[5] + 4 : mini + 2 * FNPOS * 4 + 4 * spos;
                *(float4*)da = rv; *(float4*)db = v;
            }
        }
    };

    // ---- prologue: everything that comes from memory is requested before anything is waited for ------------------------------
    constexpr int NAFF = PHASE == 0 ? 44 : 22;
    double ln_s0 = 0.0, ln_s1 = 1.0; float ln_g = 0.f, ln_b = 0.f; int ln_quad = 0, ln_sub = 0; double ln_cnt = 1.0;
    if (tid < NAFF) {   // LayerNorm sums and parameters of (cell, gate, channel)
        int k, idx;
        if (PHASE == 0) { k = tid < 32 ? 0 : tid < 40 ? 1 : 2; idx = tid - (k == 0 ? 0 : k == 1 ? 32 : 40); }
        else { k = tid < 16 ? 0 : tid < 20 ? 1 : 2; idx = tid - (k == 0 ? 0 : k == 1 ? 16 : 20); }
        const int F = k == 0 ? 16 : k == 1 ? 4 : 2;
        const int kind = idx / F, f = idx - kind * F;                // G: 0 update gate, 1 candidate;  C: 0 reset gate
        const double* st = vp(cell_sel(k, [](const FusedCell& c_) { return c_.st_in; })) + (PHASE == 0 ? (kind == 0 ? 2 : 4) : 0);
        const float* gp = kind == 0 ? cell_sel(k, [](const FusedCell& c_) { return c_.ga; }) : cell_sel(k, [](const FusedCell& c_) { return c_.oa; });
        const float* bp = kind == 0 ? cell_sel(k, [](const FusedCell& c_) { return c_.gb; }) : cell_sel(k, [](const FusedCell& c_) { return c_.ob; });
        ln_s0 = st[0]; ln_s1 = st[1];
        ln_g = gp[f]; ln_b = bp[f];
        ln_cnt = (double)a.H * a.W * F;
        if (PHASE == 0) ln_quad = k == 0 ? 4 * kind + (f >> 2) : k == 1 ? 8 + kind : 10 + kind;
        else ln_quad = k == 0 ? (f >> 2) : k == 1 ? 4 : 5;
        ln_sub = f & 3;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) load_piece(i, first);
    {   // prepared weights -> LDS: all loads of a thread in flight at once (one round trip instead of three)
        constexpr int N4 = W1_FLOATS / 4, NS4 = WS_FLOATS / 4;
        constexpr int K1 = (N4 + FNT - 1) / FNT, K2 = (NS4 + FNT - 1) / FNT;
        const f32x4* s4 = reinterpret_cast<const f32x4*>(a.w1);
        const f32x4* t4 = reinterpret_cast<const f32x4*>(a.wsmall);
        f32x4 t1[K1], t2[K2];
#pragma unroll
        for (int k = 0; k < K1; ++k) t1[k] = s4[min(tid + FNT * k, N4 - 1)];
#pragma unroll
        for (int k = 0; k < K2; ++k) t2[k] = t4[min(tid + FNT * k, NS4 - 1)];
#pragma unroll
        for (int k = 0; k < K1; ++k) reinterpret_cast<f32x4*>(wl)[min(tid + FNT * k, N4 - 1)] = t1[k];       // (the clamped lanes rewrite the last quad)
#pragma unroll
        for (int k = 0; k < K2; ++k) reinterpret_cast<f32x4*>(wsm)[min(tid + FNT * k, NS4 - 1)] = t2[k];
    }
    if (tid < NAFF) {   // scale = gamma / sqrt(var + 1e-12), shift = beta - mean * scale in float64 (tf.contrib.layers.layer_norm)
        const double mean = ln_s0 / ln_cnt;
        double var = ln_s1 / ln_cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double inv = (double)ln_g / sqrt(var + 1e-12);
        lnS[ln_quad][ln_sub] = (float)inv; lnT[ln_quad][ln_sub] = (float)((double)ln_b - mean * inv);
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
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) stage_piece(i, slab, mini, first);
#pragma unroll
    for (int i = 0; i < 6; ++i) load_piece(i, first + stride);
    __syncthreads();

    // LayerNorm moments: float within a tile (fixed lane -> pixel map), double across the tiles of a workgroup
    double st_s[MT], st_q[MT], sm_s[2] = {0.0, 0.0}, sm_q[2] = {0.0, 0.0};
#pragma unroll
    for (int m = 0; m < MT; ++m) { st_s[m] = 0.0; st_q[m] = 0.0; }
    // where this wave's small job stores (offset of the tensor in the block, bytes per pixel, first byte inside the pixel)
    int so_small = 0, small_px = 0, small_b0 = 0; bool small_live = false, small_b64 = false;
    if (PHASE == 0) {
        if (wave < 4) { so_small = (int)a.cell[1].y; small_px = 32; small_b0 = 16 * (wave >> 1); small_live = live2; }
        else if (wave < 6) { so_small = (int)a.cell[2].y; small_px = 16; small_live = live3; }
    } else {
        if (wave < 2) { so_small = (int)a.cell[1].y; small_px = 16; small_live = live2; }
        else if (wave < 4) { so_small = (int)a.cell[2].y; small_px = 8; small_live = live3; small_b64 = true; }
    }
    const bool wta_wave = PHASE == 0 && wave >= 6 && a.wta;

    int it = 0;
    for (int tile = first; tile < end; tile += stride, ++it) {
        const float* cur = slab + (it & 1) * SLAB;
        float* nxt = slab + ((it + 1) & 1) * SLAB;
        const float* mcur = mini + (it & 1) * FNPOS * 4;
        float* mnxt = mini + ((it + 1) & 1) * FNPOS * 4;
        const int th = tile / a.tiles_w, h0 = th * FTH, w0 = (tile - th * a.tiles_w) * FTW;
        const int sh = h0 + srow, sw = w0 + scol;
        const bool svalid = sh < a.H && sw < a.W;
        const int spix = sh * a.W + sw;
        // the winner-take-all accumulators of this tile's pixels, requested now (waves 6, 7 of G; out of range elsewhere)
        float wta_mp = 0.f, wta_es = 0.f;
        if (PHASE == 0) {
            wta_mp = ld_b32(rs_ws, wta_wave && svalid ? spix * 4 : BAD, (int)a.max_prob);
            wta_es = ld_b32(rs_ws, wta_wave && svalid ? spix * 4 : BAD, (int)a.exp_sum);
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
                for (int i = 0; i < 6; ++i) {
                    if ((i * (NG / 2)) / 6 == g) stage_piece(i, nxt, mnxt, tile + stride);
                    if (NG / 2 + (i * (NG - NG / 2)) / 6 == g) load_piece(i, tile + 2 * stride);
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
            const int h = h0 + wave, w = w0 + n;
            const bool ok1 = live1 && h < a.H && w < a.W;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 r = acc[m], rx = accx[m];
                const float4 o = make_float4(r[0] + (rx[0] + bias4[m][0]), r[1] + (rx[1] + bias4[m][1]), r[2] + (rx[2] + bias4[m][2]), r[3] + (rx[3] + bias4[m][3]));
                st_b128(rs_ws, ok1 ? (h * a.W + w) * (COUT * 4) + (m * 16 + 4 * kq) * 4 : BAD, so_y1, o.x, o.y, o.z, o.w);
                const float ts = (o.x + o.y) + (o.z + o.w), tq = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                st_s[m] += ok1 ? (double)ts : 0.0;
                st_q[m] += ok1 ? (double)tq : 0.0;
            }
        }

        // ---- this wave's small job: lane = pixel, v_mfma_f32_4x4x1 (lane m & 3 supplies the weights of output channel m & 3 of
        // the job's channel quad, the lane's own staged value is the B operand): acc = bias, then fma per (tap, input channel)
        // in the order [xa | xb] -- conv2d_small_body's chain.  LDS reads and matrix instructions only (see the header).
        auto job20 = [&](const float* tab) __attribute__((always_inline)) -> f32x4 {
            f32x4 r = {sbias[0], sbias[1], sbias[2], sbias[3]};
            const float* ap = tab + (lane & 3) * 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* bt = cur + soff + ((tap / 3) * FPW + (tap % 3)) * S + XA2;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const f32x4 bq = *(const f32x4*)(bt + 4 * q);
                    const f32x4 aq = *(const f32x4*)(ap + (tap * 5 + q) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) r = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[e], bq[e], r, 0, 0, 0);
                }
            }
            return r;
        };
        auto job6 = [&](const float* tab) __attribute__((always_inline)) -> f32x4 {
            f32x4 r = {sbias[0], sbias[1], sbias[2], sbias[3]};
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
                for (int e = 0; e < 4; ++e) r = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[e], b0[e], r, 0, 0, 0);
                r = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[0], b1.x, r, 0, 0, 0);
                r = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[1], b1.y, r, 0, 0, 0);
            }
            return r;
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
            st_b128(rs_ws, sok && !small_b64 ? spix * small_px + small_b0 : BAD, so_small, sr[0], sr[1], sr[2], sr[3]);
            if (PHASE == 1) st_b64(rs_ws, sok && small_b64 ? spix * small_px : BAD, so_small, sr[0], sr[1]);
            // moments: group 0 = all four channels (cell 2, one group per wave; C: the candidate) or channels 0,1 (cell 3);
            // group 1 = channels 2,3 (cell 3 gates: the update group)
            const bool pair = PHASE == 0 ? (wave >= 4) : (wave >= 2);      // cell 3: channel pairs
            float s0, q0, s1, q1;
            if (pair) { s0 = sr[0] + sr[1]; q0 = __builtin_fmaf(sr[1], sr[1], sr[0] * sr[0]); s1 = sr[2] + sr[3]; q1 = __builtin_fmaf(sr[3], sr[3], sr[2] * sr[2]); }
            else {
                s0 = 0.f; q0 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0 += sr[e]; q0 = __builtin_fmaf(sr[e], sr[e], q0); }
                s1 = 0.f; q1 = 0.f;
            }
            if (!sok) { s0 = 0.f; q0 = 0.f; s1 = 0.f; q1 = 0.f; }
            s0 = wave_sum(s0); q0 = wave_sum(q0); s1 = wave_sum(s1); q1 = wave_sum(q1);
            sm_s[0] += (double)s0; sm_q[0] += (double)q0; sm_s[1] += (double)s1; sm_q[1] += (double)q1;
            if (PHASE == 0) {   // winner-take-all update (model.py:721-731; strict '<' keeps the first maximum)
                const bool wok = wta_wave && svalid;
                const bool better = wok && wta_mp < pr;
                st_b32(rs_ws, better ? spix * 4 : BAD, (int)a.max_prob, pr);
                st_b32(rs_ws, better ? spix * 4 : BAD, (int)a.depth_image, dv.v[view]);
                st_b32(rs_ws, wok ? spix * 4 : BAD, (int)a.exp_sum, wta_es + pr);
            }
        }
        __syncthreads();
    }

    // ---- LayerNorm sums of this workgroup -> float64 atomics.  red[wave][0..3] : cell 1 (MT groups x 2), [4..7] the small job (2 x 2)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const double s = wave_sum(st_s[m]), q = wave_sum(st_q[m]);
        if (lane == 0) { red[wave][2 * m] = s; red[wave][2 * m + 1] = q; }
    }
    if (lane == 0) { red[wave][4] = sm_s[0]; red[wave][5] = sm_q[0]; red[wave][6] = sm_s[1]; red[wave][7] = sm_q[1]; }
    __syncthreads();
    if (PHASE == 0) {
        // cell 1: [0..3] = reset s,q | update s,q.  cell 2: reset = waves 0,1 ; update = waves 2,3.  cell 3: waves 4,5 (both groups)
        if (tid < 4 && live1) { double t = 0.0; for (int w = 0; w < 8; ++w) t += red[w][tid]; atomicAdd(&vp(a.cell[0].st_out)[tid], t); }
        else if (tid >= 4 && tid < 8 && live2) { const int e = tid - 4, w0 = (e >> 1) * 2; atomicAdd(&vp(a.cell[1].st_out)[e], red[w0][4 + (e & 1)] + red[w0 + 1][4 + (e & 1)]); }
        else if (tid >= 8 && tid < 12 && live3) { const int e = tid - 8; atomicAdd(&vp(a.cell[2].st_out)[e], red[4][4 + e] + red[5][4 + e]); }
    } else {
        if (tid < 2 && live1) { double t = 0.0; for (int w = 0; w < 8; ++w) t += red[w][tid]; atomicAdd(&vp(a.cell[0].st_out)[4 + tid], t); }
        else if (tid >= 2 && tid < 4 && live2) { const int e = tid - 2; atomicAdd(&vp(a.cell[1].st_out)[4 + e], red[0][4 + e] + red[1][4 + e]); }
        else if (tid >= 4 && tid < 6 && live3) { const int e = tid - 4; atomicAdd(&vp(a.cell[2].st_out)[4 + e], red[2][4 + e] + red[3][4 + e]); }
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

template <int PHASE>
int launch_fused(const FusedArgs& a0, const FusedDepth& dv, int views, hipStream_t st) {
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
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gru_fused_kernel<PHASE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    gru_fused_kernel<PHASE><<<per * views, FNT, smem, st>>>(a, dv);
    return (int)hipGetLastError();
}

}  // namespace

// ---- host side of the fused sweep (called from mvs_gru_wta_batch_f32, gru.hip) -------------------------------------------
// Workspace of one view as the fused sweep sees it (carved by gru.hip): S[k][2] state ping-pong, G[k][2] gate ping-pong, Cb[k]
// candidate, stats (a ring of planes x 3 cells x 6 doubles), x (a batch of XB cost slices).
struct GruFusedWs {
    char* base;                              // view 0's workspace block (everything below lies inside it)
    float* x; float* S[3][2]; float* G[3][2]; float* Cb[3]; double* stats;
    float *max_prob, *depth, *exp_sum;
    float *w1g, *w1c, *wsg, *wsc;            // prepared weights (shared by the views; view 0's block)
};
constexpr int GRU_FUSED_RING = 64;           // LayerNorm-sum rows: plane p uses row p % 64 (gru.hip zeroes them a batch ahead)

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

// one plane step of the pipeline: G(t) then C(t).  t runs 0 .. depth_num + 2.
int mvs_gru_fused_step(const GruFusedWs& ws, const float* const* params, int t, int depth_num, const float* x_t, int H, int W,
                       int views, size_t vstride, const float* depth_values /* host, (views, depth_num) */, hipStream_t st) {
    if (vstride >= ((size_t)1 << 31)) return MVS_E_SHAPE;            // the kernels address a view's block with 32-bit byte offsets
    auto off = [&](const void* p) { return (unsigned)((const char*)p - ws.base); };
    FusedArgs g = {}, c = {};
    g.x = c.x = x_t;
    g.ws = c.ws = ws.base;
    g.H = c.H = H; g.W = c.W = W; g.vstride = c.vstride = vstride;
    for (int k = 0; k < 3; ++k) {
        const int p = t - k;                         // this cell's plane
        const float* const* pp = params + 10 * k;
        const bool conv = p >= 0 && p < depth_num, blend = p >= 1 && p <= depth_num;
        const int pc = p < 0 ? 0 : p, pm = p < 1 ? 0 : p - 1;        // clamped plane indices for the stats rows of dead cells
        FusedCell& gc = g.cell[k];
        // s(q) lives in S[k][q & 1] (s(-1) = 0 in S[k][1]); G forms s(p-1) from s(p-2)
        gc.h = blend ? ws.S[k][p & 1] : ws.S[k][(p - 1) & 1];
        gc.c = ws.Cb[k]; gc.g = ws.G[k][(p - 1) & 1];
        gc.st_in = ws.stats + ((size_t)(pm % GRU_FUSED_RING) * 3 + k) * 6;
        gc.h_out = off(ws.S[k][(p - 1) & 1]);
        gc.y = off(ws.G[k][p & 1]);
        gc.st_out = ws.stats + ((size_t)(pc % GRU_FUSED_RING) * 3 + k) * 6;
        gc.bias = pp[1]; gc.ga = pp[4]; gc.gb = pp[5]; gc.oa = pp[8]; gc.ob = pp[9];
        gc.conv = conv; gc.blend = blend;
        FusedCell& cc = c.cell[k];
        cc.h = ws.S[k][(p - 1) & 1]; cc.c = nullptr; cc.g = ws.G[k][p & 1];
        cc.st_in = ws.stats + ((size_t)(pc % GRU_FUSED_RING) * 3 + k) * 6;
        cc.h_out = 0; cc.y = off(ws.Cb[k]); cc.st_out = ws.stats + ((size_t)(pc % GRU_FUSED_RING) * 3 + k) * 6;
        cc.bias = pp[7]; cc.ga = pp[2]; cc.gb = pp[3]; cc.oa = pp[2]; cc.ob = pp[3];
        cc.conv = conv; cc.blend = 0;
    }
    g.w1 = ws.w1g; g.wsmall = ws.wsg; c.w1 = ws.w1c; c.wsmall = ws.wsc;
    g.max_prob = off(ws.max_prob); g.depth_image = off(ws.depth); g.exp_sum = off(ws.exp_sum);
    const int q = t - 3;
    g.wta = q >= 0 && q < depth_num;
    FusedDepth dv = {};
    for (int v = 0; v < views && v < FMAXV; ++v) dv.v[v] = g.wta ? depth_values[(size_t)v * depth_num + q] : 0.f;
    int rc = launch_fused<0>(g, dv, views, st);
    if (rc) return rc;
    if (t > depth_num + 1) return 0;                 // the last output launch is C(depth_num + 1): cell 3's plane depth_num - 1
    return launch_fused<1>(c, dv, views, st);
}
