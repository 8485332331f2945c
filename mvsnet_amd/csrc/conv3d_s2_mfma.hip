// fp32-MFMA stride-2 3x3x3 convolution (the encoder's down-sampling layers 3dconv1_0/2_0/3_0,
// mvsnet/cnn_wrapper/mvsnetworks.py:130-133; tf.layers.conv3d SAME stride 2, network.py:210).
//
// Same input-stationary plane march as conv3d_mfma.hip.  Output voxel (od,oh,ow) reads input
// (2od+kd-pd, 2oh+kh-ph, 2ow+kw-pw) (pad_before = 0 for even sizes).  An input plane q = 2od+kd-pd
// feeds output plane od = (q+pd)/2 with kd = 0 and od-1 with kd = 2 when q+pd is even, and od with
// kd = 1 when it is odd; two accumulator blocks (od even / odd) are live at a time.  The staged slab
// keeps even and odd input columns in separate runs so that the 16 voxels of an MFMA column tile
// (consecutive ow, i.e. input columns 2ow+kw) are unit-stride in LDS for every kw.
#include "conv_common.h"
#include <cstdlib>

#define S2_STAGE_G(i) (((i) * (NG / 2)) / NIT)
#define S2_LOAD_G(i) (NG / 2 + ((i) * (NG - NG / 2)) / NIT)

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000u;   // buffer byte offset with bit 31 set: loads give 0, stores are dropped
constexpr int TW = CONV_TW;
constexpr int IW = 2 * TW + 1;          // staged input columns: 2*ow0-pw .. +32
constexpr int NEVEN = TW + 1;           // even-indexed staged columns (c = 0,2,..,32)

// COUT = 16 output channels per workgroup (blockIdx.y), TOH x 16 output pixels, 4 waves.
template <int CIN, int TOH, bool HAS_X2>
__global__ void __launch_bounds__(256, 1)
conv3d_s2_kernel(ConvArgs a) {
    constexpr int COUT = 16;
    constexpr int S = SlabGeom<CIN>::S;
    constexpr int IH = 2 * TOH + 1;
    constexpr int NPOS = IH * IW;
    constexpr int CQ = CIN / 4;
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    constexpr int V = TOH / 4;                     // output rows (voxel tiles) per wave
    constexpr int NROWS = 3 * COUT;
    constexpr int WROW = NROWS * 4;
    constexpr int W_FLOATS = 9 * CQ * WROW;
    constexpr int SLAB_FLOATS = NPOS * S;
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");
    static_assert(NF4 >= 256, "spare threads of the last piece redo their previous one");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                              // [9 taps][CQ][3 kd][16 co][4]
    float* slab = smem + W_FLOATS;                 // [2][NPOS][S]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;

    const int Do = (a.D + 1) / 2, Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;
    const int tiles_w = (Wo + TW - 1) / TW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int oh0 = tile_h * TOH, ow0 = tile_w * TW;
    const int co_base = blockIdx.y * COUT;
    const int od0 = blockIdx.z * a.planes_per_wg;
    const int od1 = min(od0 + a.planes_per_wg, Do);
    const int T = 2 * (od1 - od0) + 1;             // input planes 2*od0-pd .. 2*(od1-1)+2-pd
    const int q0 = 2 * od0 - a.pd;
    const int ih0 = 2 * oh0 - a.ph, iw0 = 2 * ow0 - a.pw;

    if (a.wprep) load_prepared_weights(wl, a.wprep, W_FLOATS);
    else for (int i = tid; i < W_FLOATS; i += 256) {
        int j = i & 3;
        int r = (i >> 2) % NROWS;
        int g = (i >> 2) / NROWS;
        int ciq = g % CQ, tap = g / CQ;
        int kd = r / COUT, co = r - kd * COUT;
        wl[i] = a.w[(((size_t)(kd * 9 + tap)) * CIN + ciq * 4 + j) * a.cout_total + co_base + co];
    }

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

    // Per-thread staging map, identical for every plane: byte offset inside one input plane (bit 31
    // set = outside the image: the buffer load returns 0) and float offset inside the LDS slab.
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int f = tid + 256 * i;
        if (f >= NF4) f -= 256;                    // spare threads of the last piece redo their previous one
        int pos = f / CQ;
        int r = pos / IW, c = pos - r * IW;
        int gh = ih0 + r, gw = iw0 + c;
        bool inb = gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? ((gh * a.W + gw) * CIN + 4 * c4) * 4 : OOB;
        loff[i] = (r * IW + ((c & 1) ? NEVEN + (c >> 1) : (c >> 1))) * S + 4 * c4;
    }
    const int plane_bytes = a.H * a.W * CIN * 4;
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.D * plane_bytes, 0x00020000);
    const auto x2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_X2 ? a.x2 : a.x), 0, a.D * plane_bytes, 0x00020000);

    // piece i of a plane's staging: global -> registers (load_piece), registers -> LDS (stage_piece);
    // branch-free, issued between the MFMAs of the sweep (see conv3d_mfma.hip)
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
        const bool ok = (q >= 0) && (q < a.D) && goff[i] >= 0;       // SAME pads the NORMALISED input with 0
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

    f32x4 acc[2][V];                               // block = output plane parity (relative to od0)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[b][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    int b_off[V];
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = (2 * (V * wave + v) * IW + n) * S + 4 * kq;
    const int a_lane = (kq * NROWS + n) * 4;
    constexpr int NG = 9 * (CIN / 16);             // operand groups (kh, kw, s) of one plane

    // One plane: BOTH = even input plane (kd 0 -> block NEWB, kd 2 -> the other block), else a single
    // (block, kd) pair.  `extra(g)` is called once per operand group: the march hangs the staging of
    // the next planes on it.  Operands of group g+1 are read before the MFMAs of group g.
    auto sweep = [&](auto Bc, auto Kc, auto Both, const float* buf, auto&& extra) __attribute__((always_inline)) {
        constexpr int BLK = decltype(Bc)::value, KD = decltype(Kc)::value;
        constexpr bool BOTH = decltype(Both)::value;             // BLK = NEWB, KD ignored
        f32x4 bv[2][V], av[2][2];
        auto load_grp = [&](int g, f32x4 (&b)[V], f32x4 (&aop)[2]) __attribute__((always_inline)) {
            const int tap = g / (CIN / 16), s = g % (CIN / 16);
            const int kh = tap / 3, kw = tap % 3;
            const int kwoff = (kw == 0) ? 0 : (kw == 1 ? NEVEN : 1);
#pragma unroll
            for (int v = 0; v < V; ++v) b[v] = *(const f32x4*)(buf + b_off[v] + (kh * IW + kwoff) * S + 16 * s);
            const float* wp = wl + a_lane + (tap * CQ + 4 * s) * WROW;
            if (BOTH) { aop[0] = *(const f32x4*)(wp); aop[1] = *(const f32x4*)(wp + 2 * COUT * 4); }
            else aop[0] = *(const f32x4*)(wp + KD * COUT * 4);
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
                for (int v = 0; v < V; ++v) {
                    acc[BLK][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][0][j], bv[g & 1][v][j], acc[BLK][v], 0, 0, 0);
                    if (BOTH) acc[1 - BLK][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][1][j], bv[g & 1][v][j], acc[1 - BLK][v], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const auto yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, Do * Ho * Wo * a.cout_total * 4, 0x00020000);
    int yoff[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int oh = oh0 + V * wave + v, ow = ow0 + n;
        yoff[v] = (oh < Ho && ow < Wo) ? ((oh * Wo + ow) * a.cout_total + co_base + 4 * kq) * 4 : OOB;
    }
    const int yplane_bytes = Ho * Wo * a.cout_total * 4;
    auto retire = [&](auto Bc, int od) __attribute__((always_inline)) {
        constexpr int BLK = decltype(Bc)::value;
        const bool plane_ok = (od >= od0) && (od < od1);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (plane_ok) {
                f32x4 r = acc[BLK][v];
                u32x4_t u = {__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3])};
                __builtin_amdgcn_raw_buffer_store_b128(u, yrsrc, yoff[v], od * yplane_bytes, 0);
                if (yoff[v] >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
                }
            }
            acc[BLK][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- plane march: plane q swept, q+1 staged, q+2 requested (all under the MFMAs) ------------------
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, q0);
#pragma unroll
    for (int i = 0; i < NIT; ++i) stage_piece(i, q0, slab);
#pragma unroll
    for (int i = 0; i < NIT; ++i) load_piece(i, q0 + 1);
    __syncthreads();

    // t even: od = od0 + t/2 starts (kd=0), od-1 finishes (kd=2); t odd: od = od0 + (t-1)/2 gets kd=1
    auto plane = [&](auto Pc, int t) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;      // t mod 4
        const int q = q0 + t;
        float* cur = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
        auto extra = [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                if (S2_STAGE_G(i) == g) stage_piece(i, q + 1, nxt);
                if (S2_LOAD_G(i) == g) load_piece(i, q + 2);
            }
        };
        using F = std::false_type; using Tt = std::true_type;
        if (P == 0 || P == 2) {
            // even plane: kd 0 opens output plane od_new in block NEWB, kd 2 closes od_new-1 in the other.
            // The last plane of the range (od_new == od1) opens a plane nobody retires: swept anyway --
            // a second code path here costs more (and lets the compiler hoist the staging arithmetic
            // above both paths, which gfx950 codegen then got wrong) than the 1/(3n+1) extra MFMAs.
            constexpr int NEWB = (P == 0) ? 0 : 1;
            const int od_new = od0 + t / 2;
            sweep(std::integral_constant<int, NEWB>{}, std::integral_constant<int, 0>{}, Tt{}, cur, extra);
            retire(std::integral_constant<int, 1 - NEWB>{}, od_new - 1);
        } else {
            constexpr int BLK = (P == 1) ? 0 : 1;
            sweep(std::integral_constant<int, BLK>{}, std::integral_constant<int, 1>{}, F{}, cur, extra);
        }
        __syncthreads();
    };
    for (int t = 0; t < T; t += 4) {
        plane(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) plane(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) plane(std::integral_constant<int, 2>{}, t + 2);
        if (t + 3 < T) plane(std::integral_constant<int, 3>{}, t + 3);
    }

    if (a.stats) stats_commit<COUT>(st_s, st_q, false, slab, conv_stats_row(a), a.cout_total, co_base);
}

template <int CIN, int TOH>
int launch_s2(const ConvArgs& a0, int Cout, hipStream_t st) {
    ConvArgs a = a0;
    if ((long long)a.D * a.H * a.W * (CIN > Cout ? CIN : Cout) * 4 >= (1LL << 31)) return MVS_E_SHAPE;   // 32-bit buffer offsets
    const int Do = (a.D + 1) / 2, Ho = (a.H + 1) / 2, Wo = (a.W + 1) / 2;
    const int tiles = ((Ho + TOH - 1) / TOH) * ((Wo + TW - 1) / TW);
    const int groups = Cout / 16;
    a.planes_per_wg = conv_pick_planes(Do, (long long)tiles * groups, 1);
    // test hook (tests/test_gpu_parity.py::test_stride2_plane_ranges): output planes per workgroup, so that every
    // residue of the 4-way unrolled plane march and every end-of-range case is reachable at small sizes
    if (const int v = mvs_hook(MVS_HOOK_S2_PLANES)) a.planes_per_wg = v < Do ? v : Do;
    dim3 grid(tiles, groups, (Do + a.planes_per_wg - 1) / a.planes_per_wg);
    size_t smem = (size_t)(9 * (CIN / 4) * 48 * 4 + 2 * (2 * TOH + 1) * IW * SlabGeom<CIN>::S) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s2_kernel<CIN, TOH, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s2_kernel<CIN, TOH, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (a.x2) conv3d_s2_kernel<CIN, TOH, true><<<grid, 256, smem, st>>>(a);
    else conv3d_s2_kernel<CIN, TOH, false><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

int mvs_conv3d_s2_mfma(const ConvArgs& a, int Cin, int Cout, hipStream_t st) {
    if (Cout % 16 != 0) return MVS_E_SHAPE;
    if (Cin == 32) return launch_s2<32, 4>(a, Cout, st);
    if (Cin == 16) return launch_s2<16, 4>(a, Cout, st);
    return MVS_E_SHAPE;
}
