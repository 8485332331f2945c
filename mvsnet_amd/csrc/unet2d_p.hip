// Persistent form of the UNetDS2GN convolution kernel (round 6).  Same arithmetic per output as conv2d_gn_kernel
// (unet2d.hip: fp32 MFMA implicit GEMM, producer GroupNorm (+ReLU) applied while staging, raw output + float64 group sums),
// another schedule.  What the round-6 counters and timing-only builds showed about the one-tile-per-workgroup kernel
// (profiles/r06_unet_counters_before.txt, r06_unet_diag_before.txt): the LDS staging + barriers + MFMAs ALONE take 821 of
// the 1 155 us of a pass and everything else (GroupNorm table per workgroup, loads, stores, sum atomics) ALONE 633 us -- the
// two barely overlap, because a workgroup lives for one 8 x 16 tile (72-288 MFMAs per wave) behind a prologue that reads 64
// float64 partial sums per group, and the 800-13 000 workgroups of a launch quantise badly over 256 CUs (3.1 -> 4 rounds).
//
// Here a launch is a few workgroups per CU that live for the whole layer:
//   * the layer's weights are staged in LDS ONCE per workgroup (<= 37 KB for the layers routed here), not once per tile;
//   * a workgroup owns a contiguous, balanced range of 8 x 32 (or 8 x 16) tiles, XCD-banded so that neighbouring tiles share an
//     L2.  Two slabs: while tile t is multiplied out of one, tile t+1's patch (requested a tile earlier, in registers) is written
//     to the other and tile t+2's is requested -- those instructions are placed BETWEEN the MFMAs and issue in their shadow;
//     one barrier per tile (round-6 trace of the first form, which staged between two barriers after the MFMAs: ~3 us of
//     matrix phase and ~3 us of everything else per tile, and two workgroups per CU do not hide that);
//   * the GroupNorm (scale, shift) table is built once per (workgroup, view) instead of once per tile;
//   * the output sums are float within a tile (fixed order), float64 across the tiles of a workgroup and float64 atomics
//     across workgroups -- once per (workgroup, view).
// K is the whole Cin in one pass (no chunk loop); slab pitch Cin + 4 floats (measured LDS conflict share 0.39 of the LDS-active
// cycles at + 4, 0.13 at + 8, with the same launch time: profiles/r06_unet_lds_pitch.txt).  Weight layout = the one
// mvs_conv2d_prepare_f32 writes for conv2d_gn_kernel (chunks of CG channels), the pixel-pair layout behind it for PAIR instances.
// What the final form's trace and timing-only builds say (profiles/r06_unet_persistent_device_trace.txt, r06_unet_persistent_diag.txt):
// the full-resolution 8-channel layers move ~115 MB in ~30 us (memory system), a 16 -> 16 layer is 26 us with no memory traffic
// at all (4 us prologue + 3-4 tiles of 2.4 us shared by two workgroups) and 22-25 us without MFMAs, overlapped to ~30.
#include "unet2d_common.h"

namespace {

constexpr int PTH = 8;
constexpr int NSLOT = GN_NSLOT;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// PAIR (3 x 3, stride 1, at most 8 output channels): the 16 MFMA rows are (pixel parity dx, cout) and a column is a PAIR of
// neighbouring output pixels, so that no row multiplies zeros only: against the staged column 2n + j (j = 0..3) row (dx, co)
// carries the weight of tap kw = j - dx (zero for the two corner combinations), 12 steps per input-channel quad and 32 pixels
// instead of 18 -- two thirds of the matrix time of the plain form, whose rows 8..15 are padding.
template <int KS, int STRIDE, int CIN, int CG, int MT, int TWT, bool PAIR = false>
struct PGeom {
    static constexpr int NT = PAIR ? 12 : KS * KS;       // steps per channel chunk
    static constexpr int TW = (PAIR ? 32 : 16) * TWT;
    static constexpr int IH = (PTH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
    static constexpr int NPOS = IH * IW;
    static constexpr int CQ = CIN / 4;
    static constexpr int S = CIN + 4;                   // (+ 8 / + 12 / + 20 measured: profiles/r06_unet_lds_pitch.txt -- the conflict share moves, the time does not)
    static constexpr int COUT_T = 16 * MT;
    static constexpr int W_FLOATS = NT * CIN * COUT_T;
    static constexpr int NIN = (NPOS * CQ + 255) / 256;
    static constexpr int SLAB = (NPOS * S + 3) & ~3;
    static constexpr size_t SMEM = (size_t)(2 * SLAB + W_FLOATS + 4) * sizeof(float);  // two slabs, weights, one spare float4 (writes of idle lanes)
};

template <int KS, int STRIDE, int CIN, int CG, int MT, int TWT, int WPE, bool PAIR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))      // WPE workgroups per CU: 512 / WPE registers per lane
conv2d_p_kernel(Conv2dArgs p, int tiles_h, int tiles_w, int ntiles, int per_cu) {
    using G = PGeom<KS, STRIDE, CIN, CG, MT, TWT, PAIR>;
    static_assert(!PAIR || (KS == 3 && STRIDE == 1 && MT == 1), "pixel-pair rows");
    constexpr int NT = G::NT;
    constexpr int CQ = G::CQ, S = G::S, IW = G::IW, NPOS = G::NPOS, NIN = G::NIN, COUT_T = G::COUT_T, TW = G::TW;
    constexpr int V = 2 * TWT;                      // column tiles per wave: rows 2w, 2w+1 x TWT tiles of 16 pixels
    constexpr int KL = CG / 4;                      // channels a lane reads per operand read (4: b128, 2: b64)
    constexpr int NH = CIN / CG;                    // operand reads per tap (= weight chunks)
    constexpr int CGQ = CG / 4;
    static_assert(CG == 16 || CG == 8 || CG == 4, "operand read width");
    static_assert(CIN % CG == 0 && CIN <= 64 && 256 % CQ == 0, "channel tiling");
    constexpr int NWR = (G::W_FLOATS / 4 + 255) / 256;     // weight float4 per thread
    extern __shared__ __attribute__((aligned(16))) float smem_p[];
    float* wl = smem_p + 2 * G::SLAB;               // [chunk][tap][CG/4][COUT_T][4]; slabs [2][NPOS][S] in front
    __shared__ __attribute__((aligned(16))) float aff_s[64], aff_b[64];
    __shared__ double g_mean[8], g_inv[8];
    __shared__ double red[4][MT][2][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const int cog = blockIdx.y;

    // This workgroup's tiles: a contiguous range.  The grid is `per_cu` rounds of `slots` workgroups; workgroups b, b + slots, ...
    // are dispatched to the same CU (round-robin over XCDs, then over the CUs with room), so the tiles are first dealt evenly to
    // the SLOTS (one balanced range each; slots of one XCD = blockIdx % 8 take neighbouring ranges) and a slot's range is then cut
    // into its workgroups' parts: a CU's total is what is balanced (round-6 trace: with the ranges dealt to workgroups, 1 600 tiles
    // over 512 workgroups put the 4-tile ranges of both rounds on the same CUs -- 8 tiles there, 6 everywhere else).
    const int slots = gridDim.x / per_cu, slot = blockIdx.x % slots, part = blockIdx.x / slots;
    const int ls = xcd_swizzle(slot, slots);
    const int s0 = (int)((long long)ls * ntiles / slots), s1 = (int)((long long)(ls + 1) * ntiles / slots);
    const int t0 = s0 + (int)((long long)part * (s1 - s0) / per_cu), t1 = s0 + (int)((long long)(part + 1) * (s1 - s0) / per_cu);
    const int tpv = tiles_h * tiles_w;

    // staging bookkeeping of this thread (the same for every tile): its channel quad, source, patch positions
    const int q = tid % CQ;
    const bool from_b = 4 * q >= p.a.C;
    const float* sx = from_b ? p.b.x : p.a.x;
    const int sC = from_b ? p.b.C : p.a.C;
    const int cs = from_b ? 4 * q - p.a.C : 4 * q;
    const bool relu_on = (from_b ? p.b.relu : p.a.relu) != 0;
    int rc[NIN], loff[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int f = tid + 256 * i;
        const int pos = f / CQ;
        const int r = pos / IW, c = pos - r * IW;
        const bool valid = f < NPOS * CQ;
        rc[i] = valid ? ((r << 16) | c) : (0x3fff << 16);          // a row no image has: never loaded
        loff[i] = valid ? pos * S + 4 * q : -1;
    }
    // TWO register sets for patches in flight: tile t+2's is requested at the START of tile t's MFMAs into the set tile t's own
    // patch came from, while tile t+1's is staged from the other one -- a request has a whole tile (~2 us) to come back
    // (with one set it could only go out after the last staging piece, half a microsecond before it was needed).
    float4 pinA[NIN], pinB[NIN];
    unsigned okA = 0, okB = 0;
    // Branch-free request of a tile's patch: every lane loads -- from its patch position when that lies inside the image, from
    // the tensor's first element otherwise (zeroed when staged) -- so the NIN loads go out back to back.
    auto fetch = [&](int tile, float4 (&pin)[NIN], unsigned& okmask) __attribute__((always_inline)) {
        const int view = tile / tpv, rem = tile - view * tpv;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const int ih0 = th * PTH * STRIDE - p.pad_h, iw0 = tw * TW * STRIDE - p.pad_w;
        const float* base = sx + (size_t)view * p.H * p.W * sC + cs;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int gh = ih0 + (rc[i] >> 16), gw = iw0 + (rc[i] & 0xffff);
            const bool ok = ((unsigned)gh < (unsigned)p.H) & ((unsigned)gw < (unsigned)p.W);      // (&: no branch around the load)
            const int off = ok ? (gh * p.W + gw) * sC : 0;
            pin[i] = *reinterpret_cast<const float4*>(base + off);
            okmask |= (ok ? 1u : 0u) << i;
        }
    };
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    // registers -> LDS, one patch piece: GroupNorm affine (+ReLU) on the way, zeros outside the image (SAME padding of the
    // normalised input); idle lanes of the last round write the spare float4 behind the weights
    auto stage_piece = [&](int i, float* slab_to, const float4 (&pin)[NIN], unsigned okmask) __attribute__((always_inline)) {
        float4 v = pin[i];
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (relu_on) { v.x = relu(v.x); v.y = relu(v.y); v.z = relu(v.z); v.w = relu(v.w); }
        if (!((okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float* dst = loff[i] >= 0 ? slab_to + loff[i] : wl + G::W_FLOATS;
        *reinterpret_cast<float4*>(dst) = v;
    };
    // GroupNorm (scale, shift) of every input channel of `view` -> LDS (network.py:253-267: biased variance, eps 1e-5), in two
    // halves so that the prologue can have the partial-sum loads in flight beside the weight and patch loads
    double af_sum = 0.0, af_sq = 0.0;
    auto affine_begin = [&](int view) __attribute__((always_inline)) {
        const int g = tid >> 3, part = tid & 7;             // up to 8 groups of 8 channels in [a | b]
        const int ga = p.a.C / 8;
        const bool in_a = g < ga;
        const GnSrc& src = in_a ? p.a : p.b;
        const int gl = in_a ? g : g - ga;
        af_sum = 0.0; af_sq = 0.0;
        if (g < CIN / 8 && src.stats != nullptr) {
            const double* st = src.stats + (((size_t)view * (src.C / 8) + gl) * NSLOT + part * (NSLOT / 8)) * 2;
#pragma unroll
            for (int i = 0; i < NSLOT / 8; ++i) { af_sum += st[2 * i]; af_sq += st[2 * i + 1]; }
        }
    };
    auto affine_end = [&]() __attribute__((always_inline)) {
        const int g = tid >> 3, part = tid & 7;
        const GnSrc& src = g < p.a.C / 8 ? p.a : p.b;
        const bool live = g < CIN / 8 && src.stats != nullptr;
        double sum = af_sum, sq = af_sq;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { sum += __shfl_xor(sum, o, 64); sq += __shfl_xor(sq, o, 64); }
        if (part == 0 && g < 8) {
            double mean = 0.0, inv = 1.0;                    // identity when the source has no GroupNorm
            if (live) {
                mean = sum / src.count;
                double var = sq / src.count - mean * mean;
                if (var < 0.0) var = 0.0;
                inv = 1.0 / sqrt(var + 1e-5);
            }
            g_mean[g] = mean; g_inv[g] = inv;
        }
        __syncthreads();
        if (tid < CQ) {
            const int c = 4 * tid;
            const bool ca = c < p.a.C;
            const GnSrc& s2 = ca ? p.a : p.b;
            const int cl = ca ? c : c - p.a.C;
            float a4[4] = {1.f, 1.f, 1.f, 1.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (s2.stats) {
                const double mean = g_mean[c / 8], inv = g_inv[c / 8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double a = (double)s2.gamma[cl + k] * inv;
                    a4[k] = (float)a; b4[k] = (float)((double)s2.beta[cl + k] - mean * a);
                }
            }
            *reinterpret_cast<float4*>(aff_s + c) = make_float4(a4[0], a4[1], a4[2], a4[3]);
            *reinterpret_cast<float4*>(aff_b + c) = make_float4(b4[0], b4[1], b4[2], b4[3]);
        }
        __syncthreads();
        sc = *reinterpret_cast<const float4*>(aff_s + 4 * q); sh = *reinterpret_cast<const float4*>(aff_b + 4 * q);
    };
    // output sums of this workgroup for one view: float64 per lane -> wave -> workgroup -> one atomic per (group, sum)
    double gsd[MT], gqd[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { gsd[m] = 0.0; gqd[m] = 0.0; }
    auto flush = [&](int view) __attribute__((always_inline)) {
        // lane (kq, n) holds channels 4kq..4kq+3 of row tile m: group 2m + (kq >> 1); fold n and the kq pair
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            double s = gsd[m], qq = gqd[m];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); qq += __shfl_xor(qq, o, 64); }
            s += __shfl_xor(s, 16, 64); qq += __shfl_xor(qq, 16, 64);
            if (PAIR) { s += __shfl_xor(s, 32, 64); qq += __shfl_xor(qq, 32, 64); }      // both pixel parities: the one group of 8 couts
            if (n == 0 && (kq & 1) == 0) { red[wave][m][kq >> 1][0] = s; red[wave][m][kq >> 1][1] = qq; }
            gsd[m] = 0.0; gqd[m] = 0.0;
        }
        __syncthreads();
        if (tid < MT * 4) {
            const int m = tid >> 2, h = (tid >> 1) & 1, k = tid & 1;
            const int row8 = cog * COUT_T + m * 16 + 8 * h;
            if (row8 < p.Cout) {
                const double t = (red[0][m][h][k] + red[1][m][h][k]) + (red[2][m][h][k] + red[3][m][h][k]);
                atomicAdd(&p.stats[(((size_t)view * (p.Cout / 8) + row8 / 8) * NSLOT + (blockIdx.x & (NSLOT - 1))) * 2 + k], t);
            }
        }
        __syncthreads();
    };

    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int row = 2 * wave + v / TWT, ct = v % TWT;
        b_off[v] = PAIR ? (row * IW + ct * 32 + 2 * n) * S + KL * kq : ((row * STRIDE) * IW + (ct * 16 + n) * STRIDE) * S + KL * kq;
    }
    // A rows: [chunk][tap][CG/4][co][4]; this lane supplies ci = KL * kq + j of the chunk
    const int a_off = (CG == 16) ? (kq * COUT_T + n) * 4 : (CG == 8) ? (((kq >> 1) * COUT_T + n) * 4 + 2 * (kq & 1)) : n * 4 + kq;
    // outputs leave through one buffer resource over y: an out-of-range offset drops the lane's store (no branches around them)
    const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((size_t)p.V * p.Ho * p.Wo * p.Cout * sizeof(float)), 0x00020000);
    constexpr int OOB = (int)0x80000000;

    // ---- prologue: first patch, the first view's partial sums and the weights requested together
    if (t0 >= t1) return;
    fetch(t0, pinA, okA);
    int view = t0 / tpv;
    affine_begin(view);
    {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(PAIR ? p.wpair : p.wprep + (size_t)cog * G::W_FLOATS);
        constexpr int NW4 = G::W_FLOATS / 4;
        f32x4 w_[NWR];                                 // no predicates: spare threads of the last round re-read element 0
#pragma unroll                                         // and write it to the spare float4 behind the weights
        for (int i = 0; i < NWR; ++i) { const int k = tid + 256 * i; w_[i] = w4[k < NW4 ? k : 0]; }
#pragma unroll
        for (int i = 0; i < NWR; ++i) { const int k = tid + 256 * i; reinterpret_cast<f32x4*>(wl)[k < NW4 ? k : NW4] = w_[i]; }
    }
    affine_end();                                    // (its barriers also publish the weights)
    float* slab_cur = smem_p;                        // holds the tile being multiplied
    float* slab_nxt = smem_p + G::SLAB;              // receives the next tile while that happens
#pragma unroll
    for (int i = 0; i < NIN; ++i) stage_piece(i, slab_cur, pinA, okA);
    __syncthreads();
    if (t0 + 1 < t1) fetch(t0 + 1, pinB, okB);

    // One barrier per tile.  Everything that is not an MFMA -- writing tile t+1's patch (already in registers) to the other slab,
    // requesting tile t+2's patch -- sits BETWEEN the MFMAs of tile t in program order (sched_barrier keeps it there): an MFMA
    // occupies the matrix pipe for 32 clocks and the wave's issue port for 8, so those instructions issue in its shadow.
    constexpr int NS = NT * NH;                      // (tap, chunk) steps of a tile
    constexpr int PPS = (NIN + NS - 1) / NS;         // staging pieces per step
    // A finished tile's results stay in registers (accp) and leave during the NEXT tile's MFMAs: raw outputs through buffer stores
    // (an out-of-range offset drops the lane), sums in float within the tile (fixed order), float64 across tiles; the sums of a
    // view are flushed when the first tile of another view has been emitted, or at the end.
    f32x4 accp[MT][V];
    int p_view = -1, p_oh0 = 0, p_ow0 = 0, sum_view = -1;
    auto emit_prev = [&]() __attribute__((always_inline)) {
        if (p_view < 0) return;
        if (sum_view >= 0 && sum_view != p_view && p.stats) flush(sum_view);
        sum_view = p_view;
        float gs[MT], gq[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) { gs[m] = 0.f; gq[m] = 0.f; }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            // lane (kq, n) holds GEMM rows 4kq .. 4kq+3 of column n: plain = couts 4kq.. of pixel n; PAIR = couts 4(kq&1).. of pixel 2n + (kq>>1)
            const int oh = p_oh0 + 2 * wave + v / TWT, ow = PAIR ? p_ow0 + (v % TWT) * 32 + 2 * n + (kq >> 1) : p_ow0 + (v % TWT) * 16 + n;
            const bool in = oh < p.Ho && ow < p.Wo;
            const int pix = ((p_view * p.Ho + oh) * p.Wo + ow) * p.Cout;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int co = PAIR ? 4 * (kq & 1) : cog * COUT_T + m * 16 + 4 * kq;
                const bool ok = in && co < p.Cout;
                const f32x4 r = accp[m][v];
                __builtin_amdgcn_raw_buffer_store_b128((u32x4_t){__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])},
                                                       ry, ok ? (pix + co) * 4 : OOB, 0, 0);
                const float s1 = (r[0] + r[1]) + (r[2] + r[3]), s2 = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
                gs[m] += ok ? s1 : 0.f;
                gq[m] += ok ? s2 : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) { gsd[m] += (double)gs[m]; gqd[m] += (double)gq[m]; }
        p_view = -1;
    };
    constexpr int SE = NS - 1;                       // the step behind whose MFMAs the previous tile's results leave
    // one tile: MFMAs out of slab_cur; tile + 1 staged from (pinS, okS) into slab_nxt; tile + 2 requested into (pinF, okF)
    auto tile_body = [&](int tile, const float4 (&pinS)[NIN], unsigned okS, float4 (&pinF)[NIN], unsigned& okF) __attribute__((always_inline)) {
        const int rem = tile - view * tpv;
        const int th = rem / tiles_w, tw = rem - th * tiles_w;
        const bool have_next = tile + 1 < t1;
        const int nview = have_next ? (tile + 1) / tpv : view;
        if (have_next && nview != view) { affine_begin(nview); affine_end(); }      // rare: the range crosses into another view
        f32x4 acc[MT][V];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float bq[2][V][4], aq[2][MT][4];
        auto operands = [&](int s_, int buf) __attribute__((always_inline)) {
            const int tap = s_ / NH, h = s_ - tap * NH;
            const int kh = PAIR ? tap / 4 : tap / KS, kw = PAIR ? tap - kh * 4 : tap - kh * KS;      // PAIR: kw = staged column offset j
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float* bp = slab_cur + b_off[v] + (kh * IW + kw) * S + CG * h;
                if (CG == 16) { f32x4 t = *(const f32x4*)bp; bq[buf][v][0] = t[0]; bq[buf][v][1] = t[1]; bq[buf][v][2] = t[2]; bq[buf][v][3] = t[3]; }
                else if (CG == 8) { f32x2 t = *(const f32x2*)bp; bq[buf][v][0] = t[0]; bq[buf][v][1] = t[1]; }
                else bq[buf][v][0] = *bp;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float* ap = wl + a_off + ((h * NT + tap) * CGQ * COUT_T + m * 16) * 4;
                if (CG == 16) { f32x4 t = *(const f32x4*)ap; aq[buf][m][0] = t[0]; aq[buf][m][1] = t[1]; aq[buf][m][2] = t[2]; aq[buf][m][3] = t[3]; }
                else if (CG == 8) { f32x2 t = *(const f32x2*)ap; aq[buf][m][0] = t[0]; aq[buf][m][1] = t[1]; }
                else aq[buf][m][0] = *ap;
            }
        };
        operands(0, 0);
        if (tile + 2 < t1) fetch(tile + 2, pinF, okF);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            if (s_ + 1 < NS) operands(s_ + 1, (s_ + 1) & 1);
#pragma unroll
            for (int j = 0; j < KL; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[s_ & 1][m][j], bq[s_ & 1][v][j], acc[m][v], 0, 0, 0);
            // in the shadow of this step's MFMAs: one piece of the next tile's patch -> the other slab; last step: next request
            if (have_next) {
#pragma unroll
                for (int i = s_ * PPS; i < (s_ + 1) * PPS && i < NIN; ++i) stage_piece(i, slab_nxt, pinS, okS);
            }
            if (s_ == SE) emit_prev();
            __builtin_amdgcn_sched_barrier(0);
        }
        // this tile's results wait in registers: they are stored in the shadow of the NEXT tile's MFMAs (emit_prev above)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int v = 0; v < V; ++v) accp[m][v] = acc[m][v];
        p_view = view; p_oh0 = th * PTH; p_ow0 = tw * TW;
        __syncthreads();                             // the other slab is complete, this one is free
        float* t_ = slab_cur; slab_cur = slab_nxt; slab_nxt = t_;
        view = nview;
    };
    for (int tile = t0; tile < t1; tile += 2) {
        tile_body(tile, pinB, okB, pinA, okA);
        if (tile + 1 < t1) tile_body(tile + 1, pinA, okA, pinB, okB);
    }
    emit_prev();
    if (sum_view >= 0 && p.stats) flush(sum_view);
}

template <int KS, int STRIDE, int CIN, int CG, int MT, int TWT, int WPE = 2, bool PAIR = false>
int launch_p(const Conv2dArgs& p, hipStream_t st) {
    using G = PGeom<KS, STRIDE, CIN, CG, MT, TWT, PAIR>;
    if (PAIR && (p.Cout > 8 || !p.wpair)) return MVS_E_SHAPE;
    const int tiles_h = (p.Ho + PTH - 1) / PTH, tiles_w = (p.Wo + G::TW - 1) / G::TW;
    const long long nt = (long long)p.V * tiles_h * tiles_w;
    // 32-bit element offsets inside a view of a source, 31-bit byte offsets into y (buffer stores; 0x80000000 = dropped)
    if (nt >= (1ll << 30) || (long long)p.H * p.W * (p.a.C > p.b.C ? p.a.C : p.b.C) >= (1ll << 31) ||
        (long long)p.V * p.Ho * p.Wo * p.Cout * 4 >= (1ll << 31)) return MVS_E_SHAPE;
    const int ncog = (p.Cout + G::COUT_T - 1) / G::COUT_T;
    // workgroups per CU: what fits the 160 KB of LDS beside the ~1.3 KB of tables and the register file (2 per CU at 129-256
    // registers per lane); the test hook MVS_HOOK_UNET_GRID overrides the count per launch
    int per_cu = (int)((160 * 1024) / (G::SMEM + 1536));
    if (per_cu > WPE) per_cu = WPE;
    if (per_cu < 1) per_cu = 1;
    long long slots = 256ll * per_cu / ncog;
    if (const int hk = mvs_hook(MVS_HOOK_UNET_GRID)) { slots = hk; per_cu = hk >= 512 && hk % 256 == 0 && hk / 256 <= WPE ? hk / 256 : 1; }
    int grid = (int)(nt < slots ? nt : slots);
    if (grid >= 8) grid &= ~7;                          // every XCD the same number of ranges
    if (grid < 1) grid = 1;
    if (grid % per_cu || grid / per_cu < 8) per_cu = 1; // small launches: one range per workgroup
    static bool attr_done = false;                      // per template instantiation
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_p_kernel<KS, STRIDE, CIN, CG, MT, TWT, WPE, PAIR>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv2d_p_kernel<KS, STRIDE, CIN, CG, MT, TWT, WPE, PAIR><<<dim3(grid, ncog), 256, G::SMEM, st>>>(p, tiles_h, tiles_w, (int)nt, per_cu);
    return (int)hipGetLastError();
}

// instance table: (ks, stride, cin, cg, mt) -> launcher.  CG and MT follow conv2d_tiling (unet2d.hip): the weight layout is shared.
struct PInst { int ks, stride, cin, cg, mt, cout_max; int (*fn)(const Conv2dArgs&, hipStream_t); };
const PInst P_TABLE[] = {
    {3, 1, 4, 4, 1, 8, launch_p<3, 1, 4, 4, 1, 1, 3, true>},       // 2dconv0_1 (the image, 3 + 1 channels)
    {3, 1, 8, 8, 1, 8, launch_p<3, 1, 8, 8, 1, 1, 3, true>},       // 2dconv0_2, 8_2 (8 couts: pixel-pair rows)
    {3, 1, 16, 8, 1, 8, launch_p<3, 1, 16, 8, 1, 1, 2, true>},     // 2dconv8_1 (8 | 8 -> 8)
    {3, 1, 8, 8, 1, 16, launch_p<3, 1, 8, 8, 1, 2, 3>},
    {3, 1, 16, 8, 1, 16, launch_p<3, 1, 16, 8, 1, 2>},
    {3, 1, 16, 16, 1, 16, launch_p<3, 1, 16, 16, 1, 1>},           // 2dconv1_1, 1_2, 7_2, conv9_1, 9_2 (8 x 16 tiles: 12.5 -> 13 per CU instead of 6.25 -> 7; 28.6-29.5 against 30.4-31.9 us)
    {3, 1, 32, 16, 1, 16, launch_p<3, 1, 32, 16, 1, 1>},           // 2dconv7_1 (16 | 16)
    // (measured and NOT routed here, same box: 32 -> 32 at 128 x 160 as <3, 1, 32, 16, 2, 1> 30-32 us against 28 us of the tile kernel
    //  -- 800 tiles of 4.5 us over 256 workgroups quantise 3.1 -> 4; 2dconv2_0 as <3, 2, 16, 16, 2, 1> 25.6 against 24.6 us)
    {5, 2, 8, 8, 1, 16, launch_p<5, 2, 8, 8, 1, 1>},               // conv9_0
    {5, 2, 16, 16, 2, 32, launch_p<5, 2, 16, 16, 2, 1>},           // conv10_0
};

}  // namespace

int mvs_conv2d_p_find(int ks, int stride, int cin, int cg, int mt, int cout) {      // first match: the narrowest instance that holds `cout`
    if (!mvs_hook(MVS_HOOK_UNET_PERSISTENT)) return -1;
    for (unsigned i = 0; i < sizeof(P_TABLE) / sizeof(P_TABLE[0]); ++i) {
        const PInst& t = P_TABLE[i];
        if (t.ks == ks && t.stride == stride && t.cin == cin && t.cg == cg && t.mt == mt && cout <= t.cout_max) return (int)i;
    }
    return -1;
}

// pixel-pair weight layout (PAIR instances): [chunk][step = kh * 4 + j][CG/4][16 rows = (dx, co)][4], zero where no tap applies
// (`CinSrc`: channels `w` really has: the image layer's (3,3,3,Cout) kernel is laid out for 4 channels, the fourth zero)
__global__ void conv2d_pair_weight_layout_kernel(const float* __restrict__ w, int Cin, int Cout, int CK, float* __restrict__ out, int flipT, int CinSrc) {
    const int nch = Cin / CK, CQ = CK / 4;
    const int total = nch * 12 * CQ * 16 * 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int r = i;
    const int e = r & 3; r >>= 2;
    const int row = r & 15; r >>= 4;
    const int ciq = r % CQ; r /= CQ;
    const int step = r % 12; const int ch = r / 12;
    const int kh = step / 4, j = step - 4 * kh, dx = row >> 3, co = row & 7, kw = j - dx;
    const int ci = ch * CK + ciq * 4 + e;
    if (!(kw >= 0 && kw < 3 && co < Cout && ci < Cin && ci < CinSrc)) { out[i] = 0.f; return; }
    // flipT: w is the forward kernel (3,3,Cout,Cin) of the layer whose input gradient this convolution computes (mirrored tap, swapped roles)
    out[i] = flipT ? w[((size_t)(8 - (kh * 3 + kw)) * Cout + co) * CinSrc + ci] : w[((size_t)(kh * 3 + kw) * CinSrc + ci) * Cout + co];
}

size_t mvs_conv2d_pair_floats(int ks, int stride_unused, int cin, int cout) {
    (void)stride_unused;
    return (ks == 3 && cout <= 8 && (cin == 4 || cin == 8 || cin == 16)) ? (size_t)12 * cin * 16 : 0;
}
int mvs_conv2d_pair_prepare(const float* w, int cin, int cout, int ck, float* out, hipStream_t st, int flipT, int cin_src) {
    const int total = 12 * cin * 16;
    conv2d_pair_weight_layout_kernel<<<(total + 255) / 256, 256, 0, st>>>(w, cin, cout, ck, out, flipT, cin_src > 0 ? cin_src : cin);
    return (int)hipGetLastError();
}

int mvs_conv2d_p_run(int inst, const Conv2dArgs& p, hipStream_t st) { return P_TABLE[inst].fn(p, st); }
