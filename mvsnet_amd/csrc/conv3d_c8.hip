// fp32-MFMA 3x3x3 stride-1 convolution specialised for the full-resolution 32 -> 8 layer (3dconv0_1,
// mvsnet/cnn_wrapper/mvsnetworks.py:134; tf.layers.conv3d SAME, network.py:203-215): 60 % of the
// regulariser's FLOPs run through this one kernel.
//
// Same input-stationary plane march as conv3d_mfma.hip (a workgroup owns an 8 x 16 (h x w) column,
// stages every input plane into LDS once, and each staged plane feeds the output planes q+1, q, q-1
// through the (kd, cout) weight rows), with two changes that only pay at Cout = 8:
//
//   * Row packing.  (kd, cout) is 24 rows = 1.5 MFMA row tiles; the generic kernel pads the third
//     8-row block to a full tile (75 % useful).  Here a wave owns two adjacent output rows r, r+1
//     and the third block of BOTH rows shares one tile: against the staged row r-1+i (i = 0..3)
//     tile rows 0-7 carry the weights of kh = i (output row r) and rows 8-15 those of kh = i-1
//     (output row r+1) -- the same B operand is tap kh = i of row r and tap kh = i-1 of row r+1.
//     Per (kw, ci-group) that is 6 + 4 = 10 MFMA tiles instead of 12 (90 % useful rows).
//   * XOR-swizzled slab.  Positions are stored at a 128-B pitch (no padding) with the 16-B channel
//     slot XORed by (pos >> 1) & 7: the (col = lane&15, k-quad = lane>>4) ds_read_b128 groups stay
//     bank-conflict free and the slab shrinks by 11 %, which keeps two workgroups per CU.
//
// Operand reads of (kw, ci-group) g+1 are issued between the MFMAs of g (register double buffer).
#include "conv_common.h"

namespace {

constexpr int TW = CONV_TW;
constexpr int TH = 8;
constexpr int PW = TW + 2;
constexpr int CIN = 32, COUT = 8, CQ = CIN / 4;
constexpr int NPOS = (TH + 2) * PW;            // 180 staged positions
constexpr int NF4 = NPOS * CQ;                 // 1440 float4
constexpr int NIT = (NF4 + 255) / 256;         // 6
constexpr int NROWS = 3 * COUT;                // 24
constexpr int WROW = NROWS * 4;                // floats per (tap, ci-quad)
constexpr int W_FLOATS = 9 * CQ * WROW;        // 6912
constexpr int SP = CIN;                        // slab pitch (floats), swizzled
constexpr int SLAB_FLOATS = NPOS * SP;         // 5760
constexpr int KH_FLOATS = 3 * CQ * WROW;       // weight floats per kh
constexpr int LDS_FLOATS = 2 * SLAB_FLOATS + W_FLOATS + 4;

__device__ __forceinline__ int slab_off(int pos, int slot) { return pos * SP + ((slot ^ ((pos >> 1) & 7)) << 2); }

__global__ void __launch_bounds__(256, 2)
conv3d_c8_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* slab = smem;                            // [2][NPOS][SP]
    float* wl = smem + 2 * SLAB_FLOATS;            // [9 taps][CQ][3 kd][8 co][4], then 4 zeros
    constexpr int ZOFF = W_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;

    const int tiles_w = (a.W + TW - 1) / TW;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_h = bid / tiles_w, tile_w = bid - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TW;
    const int d0 = blockIdx.z * a.planes_per_wg;
    const int d1 = min(d0 + a.planes_per_wg, a.D);
    const int T = d1 - d0 + 2;                     // input planes d0-1 .. d1

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
    const bool has_aff = a.xs != nullptr || a.bn.stats != nullptr;
    if (a.xs) { sc = *(const float4*)(a.xs + 4 * c4); sh = *(const float4*)(a.xb + 4 * c4); }
    else if (a.bn.stats) bn_affine4(a.bn, 4 * c4, sc, sh);

    float4 pre[NIT];
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int f = tid + 256 * i;
        int pos = f / CQ;
        int r = pos / PW, c = pos - r * PW;
        int gh = h0 - 1 + r, gw = w0 - 1 + c;
        bool inb = (f < NF4) && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        goff[i] = inb ? (gh * a.W + gw) * CIN + 4 * c4 : -1;
        loff[i] = (f < NF4) ? slab_off(pos, c4) : -1;
    }
    const size_t plane_elems = (size_t)a.H * a.W * CIN;

    auto issue_loads = [&](int q) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
        const float* px = a.x + (size_t)(plane_ok ? q : 0) * plane_elems;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const bool ok = plane_ok && goff[i] >= 0;
            pre[i] = ok ? *(const float4*)(px + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto write_slab = [&](int q, float* buf) __attribute__((always_inline)) {
        const bool plane_ok = (q >= 0) && (q < a.D);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (loff[i] < 0) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plane_ok && goff[i] >= 0) v = bn_relu4(pre[i], sc, sh, has_aff);   // SAME pads the normalised input
            *(float4*)(buf + loff[i]) = v;
        }
    };

    // ---- accumulators: [0],[1] = blocks 0|1 (rows 0-7 | 8-15) of output rows r, r+1; [2] = block 2
    // of row r (rows 0-7) and of row r+1 (rows 8-15).  Block b carries kd = (P - b) mod 3.
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};

    // B operand: staged row 2*wave + i (i = 0..3), column n + kw, channel slot kq (s = 0; s = 1 is ^16)
    int boff[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) boff[i][kw] = slab_off((2 * wave + i) * PW + n + kw, kq);
    const int hi = n >> 3;                         // 0: tile rows 0-7, 1: rows 8-15
    const int a_lane = (kq * NROWS + (n & 7)) * 4;

    auto sweep = [&](auto Pc, const float* buf) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int KD_B0 = P % 3, KD_B1 = (P + 2) % 3, KD_B2 = (P + 1) % 3;
        const int a0 = a_lane + (hi ? KD_B1 : KD_B0) * COUT * 4;          // blocks 0|1
        const int ap = a_lane + KD_B2 * COUT * 4 - hi * KH_FLOATS;        // block 2: kh = i - hi
        f32x4 bv[2][4], av[2][3], pv[2][4];
        // read r of group g = (kw, s): b0 a0 p0 b1 a1 p1 b2 a2 p2 b3 p3
        auto load_one = [&](int g, int r, f32x4 (&b)[4], f32x4 (&a0v)[3], f32x4 (&p)[4]) __attribute__((always_inline)) {
            const int kw = g >> 1, s = g & 1;
            const int i = (r < 9) ? r / 3 : 3, kind = (r < 9) ? r % 3 : (r == 9 ? 0 : 2);
            const int wconst = (kw * CQ + 4 * s) * WROW;
            if (kind == 0) b[i] = *(const f32x4*)(buf + (boff[i][kw] ^ (16 * s)));
            else if (kind == 1) a0v[i] = *(const f32x4*)(wl + a0 + i * KH_FLOATS + wconst);
            else {
                int off = ap + i * KH_FLOATS + wconst;
                if (i == 0) off = hi ? ZOFF : off;       // rows 8-15 would need kh = -1
                if (i == 3) off = hi ? off : ZOFF;       // rows 0-7 would need kh = 3
                p[i] = *(const f32x4*)(wl + off);
            }
        };
        constexpr int NG = 6, NR = 11, NM = 40;
#pragma unroll
        for (int r = 0; r < NR; ++r) load_one(0, r, bv[0], av[0], pv[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (g + 1 < NG) load_one(g + 1, r, bv[(g + 1) & 1], av[(g + 1) & 1], pv[(g + 1) & 1]);
#pragma unroll
                for (int m = (r * NM) / NR; m < ((r + 1) * NM) / NR; ++m) {
                    const int j = m / 10, c = m % 10;
                    // c: 0 v0.b0a0  1 P.b0p0  2 v1.b1a0  3 v0.b1a1  4 P.b1p1  5 v1.b2a1  6 v0.b2a2  7 P.b2p2  8 v1.b3a2  9 P.b3p3
                    const int t = (c == 9) ? 2 : (c % 3 == 0 ? 0 : (c % 3 == 1 ? 2 : 1));
                    const int bi = (c == 9) ? 3 : (c + 1) / 3;
                    const int ai = (c == 9) ? 3 : c / 3;
                    const float av_ = (t == 2) ? pv[g & 1][ai][j] : av[g & 1][ai][j];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, bv[g & 1][bi][j], acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // store + zero the block that has just received kd = 2 (block (P+1)%3), output plane o
    auto retire = [&](auto Pc, int o) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        constexpr int B = (P + 1) % 3;
        const bool plane_ok = (o >= d0) && (o < d1);
        const int co = 4 * (kq & 1);
        const int w = w0 + n;
        auto emit = [&](f32x4& r, int h) __attribute__((always_inline)) {
            if (plane_ok && h < a.H && w < a.W) {
                float* dst = a.y + ((((size_t)o * a.H + h) * a.W) + w) * a.cout_total + co;
                *(float4*)dst = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
            }
            r = (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        if (B < 2) {
            if ((kq >> 1) == B) { emit(acc[0], h0 + 2 * wave); emit(acc[1], h0 + 2 * wave + 1); }
        } else {
            emit(acc[2], h0 + 2 * wave + (kq >> 1));
        }
    };

    // ---- plane march -----------------------------------------------------------------------------
    issue_loads(d0 - 1);
    write_slab(d0 - 1, slab);
    __syncthreads();

    auto plane = [&](auto Pc, int t) __attribute__((always_inline)) {
        const int q = d0 - 1 + t;
        float* cur = slab + (t & 1) * SLAB_FLOATS;
        float* nxt = slab + ((t + 1) & 1) * SLAB_FLOATS;
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

    // every lane's st_* are the sums of channels 4*(kq&1) .. +3: fold lanes l and l^32
    if (a.stats) stats_commit<COUT>(st_s, st_q, true, slab, a.stats, a.cout_total, 0);
}

}  // namespace

int mvs_conv3d_c8_launch(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    if (a.x2 || a.cout_total != COUT) return MVS_E_SHAPE;
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    a.planes_per_wg = conv_pick_planes(a.D, tiles, 2);
    dim3 grid(tiles, 1, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    const size_t smem = (size_t)LDS_FLOATS * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3d_c8_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    conv3d_c8_kernel<<<grid, 256, smem, st>>>(a);
    return (int)hipGetLastError();
}
