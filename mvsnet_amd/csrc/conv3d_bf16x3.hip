// Split-precision ("bf16x3") variant of the stride-1 input-stationary convolution of conv3d_mfma.hip.
//
// OPT-IN (mvs_set_conv_impl(MVS_CONV_IMPL_BF16X3)); the default regulariser is exact fp32 MFMA.
// gfx950's bf16 matrix rate is 16x its fp32 matrix rate, so every fp32 operand is split into two
// bf16 numbers x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits) and a product is
// evaluated as  a*b ~ ah*bh + ah*bl + al*bh  with three v_mfma_f32_16x16x32_bf16 accumulating in
// fp32 (the dropped al*bl term is 2^-16 of 2^-16).  Relative error per product ~1.5e-5 instead of
// 6e-8; 3 MFMAs at 16x the rate = 5.3x the fp32-MFMA throughput.  Inputs, outputs, BatchNorm
// statistics and every other kernel stay fp32; tests hold the end-to-end depth to the same 1e-3
// relative-L1 bar of the north star (tests/test_gpu_bf16x3.py).
//
// Structure identical to conv3d_s1_kernel: a workgroup marches a TH x 16 column along depth, rows =
// (kd, cout), columns = 16 voxels, K = (kh, kw, ci); one MFMA now covers 32 input channels of one
// tap.  LDS holds, per staged position, CIN hi halves then CIN lo halves (the split is done once,
// while staging, after the producer's BatchNorm+ReLU); weights are pre-split on the host side
// (mvs_regnet_prepare_f32).
#include "conv_common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TW = CONV_TW;
constexpr int PW = TW + 2;

__device__ __forceinline__ unsigned short to_bf16(float x) {
    __bf16 h = (__bf16)x;                                   // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float from_bf16(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }

// fp32 x4 -> (hi x4, lo x4) packed as two 8-byte words
__device__ __forceinline__ void split4(float4 v, uint2& hi, uint2& lo) {
    unsigned short h0 = to_bf16(v.x), h1 = to_bf16(v.y), h2 = to_bf16(v.z), h3 = to_bf16(v.w);
    unsigned short l0 = to_bf16(v.x - from_bf16(h0)), l1 = to_bf16(v.y - from_bf16(h1));
    unsigned short l2 = to_bf16(v.z - from_bf16(h2)), l3 = to_bf16(v.w - from_bf16(h3));
    hi = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2 | ((unsigned)h3 << 16));
    lo = make_uint2((unsigned)l0 | ((unsigned)l1 << 16), (unsigned)l2 | ((unsigned)l3 << 16));
}

template <int CIN, int COUT, int TH> struct BfGeom {
    static constexpr int PB = CIN * 4 + 16;                 // bytes per staged position (hi | lo | pad)
    static constexpr int RB = CIN * 4;                      // bytes per weight row (hi | lo)
    static constexpr int NROWS = 3 * COUT;
    static constexpr int W_BYTES = 9 * NROWS * RB;
    static constexpr int SLAB_BYTES = (TH + 2) * PW * PB;
    static constexpr int LDS_BYTES = W_BYTES + 2 * SLAB_BYTES;
    static constexpr int WGS_PER_CU = (2 * LDS_BYTES <= 160 * 1024) ? 2 : 1;
};

template <int CIN, int COUT, int TH, bool HAS_X2>
__global__ void __launch_bounds__(256, (BfGeom<CIN, COUT, TH>::WGS_PER_CU))
conv3d_s1_bf16_kernel(ConvArgs a) {
    using G = BfGeom<CIN, COUT, TH>;
    constexpr int PB = G::PB, RB = G::RB, NROWS = G::NROWS;
    constexpr int NPOS = (TH + 2) * PW;
    constexpr int CQ = CIN / 4;
    constexpr int NF4 = NPOS * CQ;
    constexpr int NIT = (NF4 + 255) / 256;
    constexpr int V = TH / 4;
    constexpr int MT = (NROWS + 15) / 16;
    constexpr int KC = CIN / 32;                   // 32-channel K chunks per tap
    static_assert(256 % CQ == 0 && CIN % 32 == 0, "tiling");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* wl = smem_b;                    // [9 taps][NROWS][CIN hi | CIN lo] bf16
    unsigned char* slab = smem_b + G::W_BYTES;     // [2][NPOS][PB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;

    const int tiles_w = (a.W + TW - 1) / TW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TW;
    const int co_base = blockIdx.y * COUT;
    const int d0 = blockIdx.z * a.planes_per_wg;
    const int d1 = min(d0 + a.planes_per_wg, a.D);
    const int T = d1 - d0 + 2;

    {   // pre-split weights of this cout group: coalesced copy
        const uint4* s4 = reinterpret_cast<const uint4*>(a.wprep_bf + (size_t)blockIdx.y * (G::W_BYTES / 2));
        for (int i = tid; i < G::W_BYTES / 16; i += 256) reinterpret_cast<uint4*>(wl)[i] = s4[i];
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

    float4 pre[NIT];
    float4 pre2[HAS_X2 ? NIT : 1];
    int goff[NIT], loff[NIT];                      // plane element offset / LDS byte offset of the hi word
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int f = tid + 256 * i;
        int pos = f / CQ;
        int r = pos / PW, c = pos - r * PW;
        int gh = h0 - 1 + r, gw = w0 - 1 + c;
        bool inb = (f < NF4) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? (gh * a.W + gw) * CIN + 4 * c4 : -1;
        loff[i] = (f < NF4) ? pos * PB + 8 * c4 : -1;
    }
    const size_t plane_elems = (size_t)a.H * a.W * CIN;

    auto issue_loads = [&](int q) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
        const float* px = a.x + (size_t)(plane_ok ? q : 0) * plane_elems;
        const float* px2 = HAS_X2 ? a.x2 + (size_t)(plane_ok ? q : 0) * plane_elems : nullptr;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = plane_ok && goff[i] >= 0;
            pre[i] = ok ? *(const float4*)(px + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (HAS_X2) pre2[i] = ok ? *(const float4*)(px2 + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto write_slab = [&](int q, unsigned char* buf) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (loff[i] < 0) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plane_ok && goff[i] >= 0) {
                v = bn_relu4(pre[i], sc, sh, has_aff);
                if (HAS_X2) {
                    float4 v2 = bn_relu4(pre2[i], sc2, sh2, has_aff2);
                    v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
                }
            }
            uint2 hi, lo;
            split4(v, hi, lo);
            *(uint2*)(buf + loff[i]) = hi;
            *(uint2*)(buf + loff[i] + CIN * 2) = lo;
        }
    };

    f32x4 acc[MT][V];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[m][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    int b_off[V];                                  // byte offsets
#pragma unroll
    for (int v = 0; v < V; ++v) b_off[v] = ((V * wave + v) * PW + n) * PB + 16 * kq;
    int row_blk[MT], row_co[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { int r = m * 16 + n; row_blk[m] = r / COUT; row_co[m] = r % COUT; }

    auto sweep = [&](auto Pc, const unsigned char* buf) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        int a_off[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int b = row_blk[m];
            int kd = (b < 3) ? ((P - b + 3) % 3) : 0;
            a_off[m] = (kd * COUT + row_co[m]) * RB + 16 * kq;
        }
        constexpr int NG = 9 * KC;
        constexpr int NR = 2 * (V + MT);           // hi and lo read per operand tile
        constexpr int NM = 3 * MT * V;             // hh, hl, lh per (row tile, voxel tile)
        bf16x8 bh[2][V], bl[2][V], ah[2][MT], al[2][MT];
        auto load_one = [&](int g, int r, int buf_i) __attribute__((always_inline)) {
            const int tap = g / KC, s = g % KC;
            const int kh = tap / 3, kw = tap % 3;
            const int t = r >> 1, lo = r & 1;
            if (t < V) {
                const bf16x8 x = *(const bf16x8*)(buf + b_off[t] + (kh * PW + kw) * PB + s * 64 + lo * (CIN * 2));
                if (lo) bl[buf_i][t] = x; else bh[buf_i][t] = x;
            } else {
                const bf16x8 x = *(const bf16x8*)(wl + a_off[t - V] + tap * NROWS * RB + s * 64 + lo * (CIN * 2));
                if (lo) al[buf_i][t - V] = x; else ah[buf_i][t - V] = x;
            }
        };
#pragma unroll
        for (int r = 0; r < NR; ++r) load_one(0, r, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (g + 1 < NG) load_one(g + 1, r, (g + 1) & 1);
#pragma unroll
                for (int i = (r * NM) / NR; i < ((r + 1) * NM) / NR; ++i) {
                    const int term = i / (MT * V), m = (i / V) % MT, v = i % V;
                    const bf16x8 av = (term == 2) ? al[g & 1][m] : ah[g & 1][m];
                    const bf16x8 bv = (term == 1) ? bl[g & 1][v] : bh[g & 1][v];
                    acc[m][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[m][v], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto retire = [&](auto Pc, int o) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int B = (P + 1) % 3;
        constexpr int mt = (B * COUT) / 16;
        constexpr int row0 = (B * COUT) % 16;
        const bool plane_ok = (o >= d0) && (o < d1);
        const bool mine = (4 * kq >= row0) && (4 * kq < row0 + COUT);
        const int co = 4 * kq - row0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (mine) {
                int h = h0 + V * wave + v, w = w0 + n;
                if (plane_ok && h < a.H && w < a.W) {
                    f32x4 r = acc[mt][v];
                    float* dst = a.y + ((((size_t)o * a.H + h) * a.W) + w) * a.cout_total + co_base + co;
                    *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
                }
                acc[mt][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    issue_loads(d0 - 1);
    write_slab(d0 - 1, slab);
    __syncthreads();

    auto plane = [&](auto Pc, int t) __attribute__((always_inline)) {
        const int q = d0 - 1 + t;
        unsigned char* cur = slab + (t & 1) * G::SLAB_BYTES;
        unsigned char* nxt = slab + ((t + 1) & 1) * G::SLAB_BYTES;
        const bool more = (t + 1 < T);
        if (more) issue_loads(q + 1);
        if (q >= 0 && q < a.D) sweep(Pc, cur);
        retire(Pc, q - 1);
        if (more) write_slab(q + 1, nxt);
        __syncthreads();
    };
    for (int t = 0; t < T; t += 3) {
        plane(std::integral_constant<int, 0>{}, t);
        if (t + 1 < T) plane(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < T) plane(std::integral_constant<int, 2>{}, t + 2);
    }

    if (a.stats) stats_commit<COUT>(st_s, st_q, COUT == 8, reinterpret_cast<float*>(slab), conv_stats_row(a), a.cout_total, co_base);
}

// TensorFlow (3,3,3,Cin,Cout) fp32 -> per cout group [tap9][(kd,co) rows][Cin hi | Cin lo] bf16
__global__ void weight_split_kernel(const float* __restrict__ w, int Cin, int Cout, int G,
                                    unsigned short* __restrict__ out) {
    const int total = 27 * Cin * Cout;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int per_group = 27 * Cin * G;             // weights per group (each becomes hi and lo)
    const int g = i / per_group;
    int r = i - g * per_group;
    const int ci = r % Cin; r /= Cin;
    const int rows = 3 * G;
    const int row = r % rows, tap = r / rows;
    const int kd = row / G, co = row - kd * G;
    float x = w[((size_t)(kd * 9 + tap) * Cin + ci) * Cout + g * G + co];
    unsigned short hi = to_bf16(x);
    unsigned short lo = to_bf16(x - from_bf16(hi));
    size_t base = (size_t)g * per_group * 2 + ((size_t)(tap * rows + row) * 2) * Cin;
    out[base + ci] = hi;
    out[base + Cin + ci] = lo;
}

template <int CIN, int COUT, int TH>
int launch_s1_bf16(const ConvArgs& a0, int Cout, hipStream_t st) {
    using G = BfGeom<CIN, COUT, TH>;
    ConvArgs a = a0;
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    const int groups = Cout / COUT;
    a.planes_per_wg = conv_pick_planes(a.D, (long long)tiles * groups, 2);
    dim3 grid(tiles, groups, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    size_t smem = G::LDS_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s1_bf16_kernel<CIN, COUT, TH, true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        if ((e = hipFuncSetAttribute((const void*)conv3d_s1_bf16_kernel<CIN, COUT, TH, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return (int)e;
        attr_done = true;
    }
    if (a.x2) conv3d_s1_bf16_kernel<CIN, COUT, TH, true><<<grid, 256, smem, st>>>(a);
    else conv3d_s1_bf16_kernel<CIN, COUT, TH, false><<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

// stride-1 shapes covered by the split-precision kernel (same cout grouping as conv_coutg kind 0)
bool mvs_conv3d_bf16x3_supported(int Cin, int Cout) {
    return (Cin == 32 && (Cout == 8 || Cout % 16 == 0)) || (Cin == 64 && Cout % 8 == 0);
}

int mvs_conv3d_s1_bf16x3(const ConvArgs& a, int Cin, int Cout, hipStream_t st) {
    if (!a.wprep_bf) return MVS_E_SHAPE;
    if (Cin == 32 && Cout == 8) return launch_s1_bf16<32, 8, 8>(a, Cout, st);
    if (Cin == 32 && Cout % 16 == 0) return launch_s1_bf16<32, 16, 8>(a, Cout, st);
    if (Cin == 64 && Cout % 8 == 0) return launch_s1_bf16<64, 8, 4>(a, Cout, st);
    return MVS_E_SHAPE;
}

int mvs_conv_weight_split(const float* w, int Cin, int Cout, unsigned short* out, hipStream_t st) {
    const int G = conv_coutg(0, Cin, Cout);
    if (G == 0 || !mvs_conv3d_bf16x3_supported(Cin, Cout)) return MVS_E_SHAPE;
    const int total = 27 * Cin * Cout;
    weight_split_kernel<<<mvs_cdiv(total, 256), 256, 0, st>>>(w, Cin, Cout, G, out);
    return (int)hipGetLastError();
}
