// fp32-MFMA stride-2 3x3x3 transposed convolution for 8 output channels per workgroup: the decoder's
// full-resolution layer 3dconv6_0 (16 -> 8) and, in four channel groups, 3dconv4_0 (64 -> 32)
// (mvsnet/cnn_wrapper/mvsnetworks.py:146-154; tf.layers.conv3d_transpose SAME, network.py:327).
//
// Arithmetic as in deconv3d_mfma.hip: out[2i + k] += in[i] * W[k][co][ci] per axis, cropped to
// [0, 2n); coarse input planes are marched once through LDS (one halo row / column in front); plane q
// completes the even output plane 2q (its kd = 0 taps + the kd = 2 taps of plane q-1, carried in
// registers), produces the odd plane 2q+1 (kd = 1) and starts the next even plane (kd = 2).
//
// With Cout = 8 a 16-row MFMA tile is half empty, so two taps share each tile: rows 0-7 carry the
// weights of one tap, rows 8-15 those of another tap that reads the SAME staged position (same
// -1 shifts in h and w) but lands in a different output plane or parity class:
//   F[c]  c = 2*(kh&1) + (kw&1):  kd = 0 | kd = 1 of class c  (halves swapped for c = 1, 3)
//   CA = (kd = 2, class 00) | (kd = 2, class 01),   CB = (kd = 2, class 10) | (kd = 2, class 11)
// so that the carried kd = 2 halves line up lane-for-lane with the kd = 0 halves they are added to.
// 15 tile-taps per plane instead of 27, and every lane stores useful rows.
//
// Staging of the next plane and the request for the one after are issued between the MFMAs
// (branch-free pieces, buffer addressing), as in conv3d_c8.hip.
#include "conv_common.h"

namespace {

constexpr int TW = CONV_TW;
constexpr int PW = TW + 1;              // staged row width (one halo column on the left)
constexpr int COUT = 8;
constexpr int TH = 8;
constexpr int V = TH / 4;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;

// tile-tap table: accumulator (0-3 = F[c], 4 = CA, 5 = CB), tap of rows 0-7, tap of rows 8-15
// (-1 = none), both as (kd*3 + kh)*3 + kw; grouped by the staged-position shift they read.
struct TapPair { int acc, t0, t1; };
constexpr int tp(int kd, int kh, int kw) { return (kd * 3 + kh) * 3 + kw; }
constexpr int NSH = 4;                                   // shifts: none, -1 (w), -PW (h), -PW-1
constexpr int SH_COUNT[NSH] = {6, 4, 3, 2};
constexpr int SH_FIRST[NSH] = {0, 6, 10, 13};
constexpr TapPair TAPS[15] = {
    // shift none: taps with kh, kw in {0, 1}
    {0, tp(0, 0, 0), tp(1, 0, 0)}, {1, tp(1, 0, 1), tp(0, 0, 1)}, {2, tp(0, 1, 0), tp(1, 1, 0)},
    {3, tp(1, 1, 1), tp(0, 1, 1)}, {4, tp(2, 0, 0), tp(2, 0, 1)}, {5, tp(2, 1, 0), tp(2, 1, 1)},
    // shift -1: kw = 2, kh in {0, 1}
    {0, tp(0, 0, 2), tp(1, 0, 2)}, {2, tp(0, 1, 2), tp(1, 1, 2)}, {4, tp(2, 0, 2), -1}, {5, tp(2, 1, 2), -1},
    // shift -PW: kh = 2, kw in {0, 1}
    {0, tp(0, 2, 0), tp(1, 2, 0)}, {1, tp(1, 2, 1), tp(0, 2, 1)}, {4, tp(2, 2, 0), tp(2, 2, 1)},
    // shift -PW-1: kh = kw = 2
    {0, tp(0, 2, 2), tp(1, 2, 2)}, {4, tp(2, 2, 2), -1},
};

template <int CIN, bool HAS_X2>
__global__ void __launch_bounds__(256, (CIN <= 32 ? 2 : 1))     // CIN = 64: LDS allows one workgroup per CU anyway
deconv3d_c8_kernel(ConvArgs a) {
    constexpr int S = SlabGeom<CIN>::S;
    constexpr int NPOS = (TH + 1) * PW;
    constexpr int CQ = CIN / 4;
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    constexpr int WROW = COUT * 4;                 // floats per (tap, ci-quad) group
    constexpr int W_FLOATS = 27 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    constexpr int NS = CIN / 16;
    constexpr int NG = NSH * NS;                   // operand groups (s, shift)
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");
    static_assert(NF4 >= 256, "spare threads of the last piece redo their previous one");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                              // [27 taps][CQ][8 co][4], then a zero block of CQ*WROW floats
    float* slab = smem + W_FLOATS + CQ * WROW;     // [2][NPOS][S]
    constexpr int ZTAP = 27;                       // index of the all-zero "tap"

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;

    const int tiles_w = (a.W + TW - 1) / TW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TW;
    const int co_base = blockIdx.y * COUT;
    const int q0 = blockIdx.z * a.planes_per_wg;
    const int q1 = min(q0 + a.planes_per_wg, a.D);
    const int T = q1 - q0 + 1;                     // coarse planes q0-1 .. q1-1
    const int Ho = 2 * a.H, Wo = 2 * a.W;

    // weights (kd,kh,kw,Cout,Cin) -> LDS [tap][ci/4][co][ci%4]
    if (a.wprep) load_prepared_weights(wl, a.wprep, W_FLOATS);
    else for (int i = tid; i < W_FLOATS; i += 256) {
        int j = i & 3;
        int co = (i >> 2) % COUT;
        int g = (i >> 2) / COUT;
        int ciq = g % CQ, tap = g / CQ;
        wl[i] = a.w[((size_t)tap * a.cout_total + co_base + co) * CIN + ciq * 4 + j];
    }
    for (int i = tid; i < CQ * WROW; i += 256) wl[W_FLOATS + i] = 0.f;

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
    const float lo = has_aff ? 0.f : -INFINITY, lo2 = has_aff2 ? 0.f : -INFINITY;   // ReLU floor (or identity)

    float4 pre[NIT];
    float4 pre2[HAS_X2 ? NIT : 1];
    int goff[NIT], loff[NIT];                      // byte offset in an input plane (bit 31 = outside), LDS float offset
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
    const int yplane_bytes = Ho * Wo * a.cout_total * 4;
    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, 2 * a.D * yplane_bytes, 0x00020000);

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

    // ---- accumulators: [0..3] = F[c], [4] = CA, [5] = CB; per voxel tile ------------------------------
    f32x4 acc[6][V];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[t][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = ((V * wave + v + 1) * PW + n + 1) * S + 4 * kq;
    const bool hi = (n >> 3) != 0;                           // tile rows 8-15
    const int a_lane = (kq * COUT + (n & 7)) * 4;

    // One coarse plane.  FIRST: only the kd = 2 tiles (plane q0-1 feeds nothing else of this chunk).
    auto sweep = [&](auto Fc, const float* buf, auto&& extra) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(Fc)::value;
        constexpr int MAXT = 6;
        f32x4 bv[2][V], av[2][MAXT];
        auto load_grp = [&](int g, f32x4 (&b)[V], f32x4 (&aop)[MAXT]) __attribute__((always_inline)) {
            const int s = g / NSH, shi = g % NSH;
            const int shift = -((shi >> 1) ? PW : 0) - ((shi & 1) ? 1 : 0);
#pragma unroll
            for (int v = 0; v < V; ++v) b[v] = *(const f32x4*)(buf + b_off[v] + shift * S + 16 * s);
#pragma unroll
            for (int i = 0; i < SH_COUNT[shi]; ++i) {
                const TapPair tpair = TAPS[SH_FIRST[shi] + i];
                if (FIRST && tpair.acc < 4) continue;
                const int t1 = tpair.t1 < 0 ? ZTAP : tpair.t1;
                const int o0 = (tpair.t0 * CQ + 4 * s) * WROW, o1 = (t1 * CQ + 4 * s) * WROW;
                aop[i] = *(const f32x4*)(wl + a_lane + (hi ? o1 : o0));
            }
        };
        load_grp(0, bv[0], av[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) load_grp(g + 1, bv[(g + 1) & 1], av[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            extra(g);
            const int shi = g % NSH;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < SH_COUNT[shi]; ++i) {
                    const int t = TAPS[SH_FIRST[shi] + i].acc;
                    if (FIRST && t < 4) continue;
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        acc[t][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][i][j], bv[g & 1][v][j], acc[t][v], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Lane (kq, n) holds rows 4kq .. 4kq+3 of every tile: rows 0-7 (kq < 2) and 8-15 (kq >= 2) are the
    // two halves.  Class c even-plane half: rows 0-7 for c = 0, 2 and rows 8-15 for c = 1, 3.
    const bool upper = kq >= 2;
    int yoff[4][V];                                  // byte offset of this lane's float4 inside its output plane
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int h = h0 + V * wave + v, w = w0 + n;
            const int oh = 2 * h + (c >> 1), ow = 2 * w + (c & 1);
            yoff[c][v] = (h < a.H && w < a.W) ? ((oh * Wo + ow) * a.cout_total + co_base + 4 * (kq & 1)) * 4 : OOB;
        }
    auto store_plane = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool even_half = ((c & 1) != 0) == upper;      // this lane's half belongs to plane 2q (else 2q+1)
            const int od = 2 * q + (even_half ? 0 : 1);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x4 r = acc[c][v];
                u32x4_t u = {__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])};
                __builtin_amdgcn_raw_buffer_store_b128(u, yrsrc, yoff[c][v] + od * yplane_bytes, 0, 0);
                if (yoff[c][v] >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
                }
            }
        }
    };
    // next plane: F[c] starts from the carried kd = 2 half on the even-plane lanes, 0 elsewhere
    auto rotate = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool even_half = ((c & 1) != 0) == upper;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                f32x4 cv = acc[4 + (c >> 1)][v];
                acc[c][v] = even_half ? cv : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) { acc[4][v] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[5][v] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    };

    // ---- plane march -------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, q0 - 1);
#pragma unroll
    for (int i = 0; i < NIT; ++i) stage_piece(i, q0 - 1, slab);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, q0);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int q = q0 - 1 + t;
        float* cur = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
        auto extra = [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                if ((i * (NG / 2)) / NIT == g) stage_piece(i, q + 1, nxt);
                if (NG / 2 + (i * (NG - NG / 2)) / NIT == g) load_piece(i, q + 2);
            }
        };
        if (t == 0) sweep(std::true_type{}, cur, extra);       // planes outside the volume are staged as zeros
        else {
            sweep(std::false_type{}, cur, extra);
            store_plane(q);
        }
        rotate();
        __syncthreads();
    }

    // every lane's st_* are the sums of channels 4*(kq&1) .. +3: fold lanes l and l^32
    if (a.stats) stats_commit<COUT>(st_s, st_q, true, slab, conv_stats_row(a), a.cout_total, co_base);
}

template <int CIN>
int launch_deconv_c8(const ConvArgs& a0, int Cout, hipStream_t st) {
    ConvArgs a = a0;
    if ((long long)a.D * a.H * a.W * 8 * (CIN > Cout ? CIN / 8 : Cout) * 4 >= (1LL << 31)) return MVS_E_SHAPE;   // 32-bit buffer offsets
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    const int groups = Cout / COUT;
    // CIN = 16 (3dconv6_0): two workgroups share a CU and the kernel waits on memory more than half of its time, so shorter
    // plane runs for twice the workgroups pay (56.8 -> 53.3 us at the metric size); the other users of this rule measured slower
    a.planes_per_wg = conv_pick_planes(a.D, (long long)tiles * groups, 1, CIN <= 16 ? 512 : 256);
    dim3 grid(tiles, groups, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    size_t smem = (size_t)(28 * CIN * COUT + 2 * (TH + 1) * PW * SlabGeom<CIN>::S) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)deconv3d_c8_kernel<CIN, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)deconv3d_c8_kernel<CIN, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (a.x2) deconv3d_c8_kernel<CIN, true><<<grid, 256, smem, st>>>(a);
    else deconv3d_c8_kernel<CIN, false><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_deconv3d_c8_launch(const ConvArgs& a, int Cin, int Cout, hipStream_t st) {
    if (Cout % COUT) return MVS_E_SHAPE;
    if (Cin == 16) return launch_deconv_c8<16>(a, Cout, st);
    if (Cin == 64) return launch_deconv_c8<64>(a, Cout, st);
    return MVS_E_SHAPE;
}
