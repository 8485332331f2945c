// Output-stationary fp32-MFMA 3x3x3 convolutions for the LOW-RESOLUTION levels of RegNetUS0
// (mvsnet/cnn_wrapper/mvsnetworks.py:131-150: 3dconv2_0, 3_0 stride 2; 2_1, 3_1 stride 1; 4_0, 5_0 transposed;
// tf.layers.conv3d / conv3d_transpose SAME, network.py:210,327).
//
// Why a second kernel family: at 1/4 and 1/8 resolution the volumes are 61 440 and 7 680 voxels.  The plane-march
// kernels (conv3d_mfma.hip) pay there (rocprofv3 counters, round 1) 2x padded MFMA work -- (kd, cout) rows that do
// not fill a 16-row tile, halo planes swept for 3-8 plane ranges -- plus a 55 KB weight upload into LDS at the head
// of every workgroup: 0.12-0.26 of the fp32 MFMA peak.  Here a workgroup owns a small 3D BLOCK of output voxels:
//   * GEMM rows = 16 output channels (a full tile, no (kd, cout) packing), columns = a 4x4 (h x w) patch of voxels,
//     K = (tap, ci); accumulators stay in registers for the whole K loop (27 * Cin / 4 MFMAs per tile), so the
//     executed MFMA work equals the algorithmic work;
//   * the block's input neighbourhood (halo included) is staged ONCE into LDS with the producer's BatchNorm + ReLU
//     (+ additive skip) applied on the way in; the B operand is one ds_read_b128 per (tap, 16-channel group, tile)
//     at lane base + immediate offset -- any patch shape and any stride is free;
//   * the A operand (weights) never touches LDS: the pre-laid-out weights (mvs_regnet_prepare_f32) are read straight
//     from L2 into registers, one coalesced 1 KB global_load_dwordx4 per wave and (tap, 16-channel group), eight steps
//     ahead of use.  No weight upload, no prologue barrier for it, and the LDS footprint (20-61 KB) lets several
//     workgroups -- of this kernel and of the branch layers on the side stream -- share a CU;
//   * the four waves split the output channels (and the voxel tiles when Cout < 64), never K: no reduction.
// Transposed convolution: per axis an even output 2m takes (i = m, k = 0) and (i = m-1, k = 2), an odd output 2m+1
// takes (i = m, k = 1): the 8 output parity classes are 8 accumulators over the same staged input block.
#include "conv_common.h"
#include <cstdlib>

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int OS_OOB = (int)0x80000000u;   // buffer byte offset with bit 31 set: loads return 0

enum { OS_S1 = 0, OS_S2 = 1, OS_DECONV = 2 };


template <int KIND, int CIN, int NCW, int VT, int BD, int BHT, int BWT>
struct OsGeom {
    static constexpr int NVW = 4 / NCW;                    // waves along the voxel tiles
    static constexpr int NT = NVW * VT;                    // voxel tiles (4x4 patches) per block
    static_assert(NT == BD * BHT * BWT, "block shape must hold the workgroup's voxel tiles");
    static constexpr int BH = 4 * BHT, BW = 4 * BWT;
    static constexpr int PD = KIND == OS_S1 ? BD + 2 : KIND == OS_S2 ? 2 * BD + 1 : BD + 1;
    static constexpr int PH = KIND == OS_S1 ? BH + 2 : KIND == OS_S2 ? 2 * BH + 1 : BH + 1;
    static constexpr int PW = KIND == OS_S1 ? BW + 2 : KIND == OS_S2 ? 2 * BW + 1 : BW + 1;
    static constexpr int S = CIN + 4;                      // floats per staged position (16-B aligned, odd slot count)
    static constexpr int NPOS = PD * PH * PW;
    static constexpr int CQ = CIN / 4;
    static constexpr int NF4 = NPOS * CQ;
    static constexpr int NIT = (NF4 + 255) / 256;
    static constexpr int G = CIN / 16;
    static constexpr int NS = 27 * G;                      // K steps: (tap, 16-channel group)
    static constexpr int NACC = KIND == OS_DECONV ? 8 : 1;
    static constexpr int LDS_BYTES = NPOS * S * 4;
    static_assert(LDS_BYTES >= 4 * 2 * 64 * 4, "the BatchNorm reduction reuses the block's LDS");
    static_assert(256 % CQ == 0, "channel quad per thread must be loop invariant");
};

// 27 taps in TensorFlow order tap = kd*9 + kh*3 + kw
__host__ __device__ constexpr int tap_kd(int t) { return t / 9; }
__host__ __device__ constexpr int tap_kh(int t) { return (t / 3) % 3; }
__host__ __device__ constexpr int tap_kw(int t) { return t % 3; }

// One workgroup's block.  `blk` of `nblk` = the block's index in the layer's block grid (a launch's blockIdx.x of gridDim.x, or
// a slice of another layer's launch: conv3d_os_filled_kernel), `cog` = the group of 16 * NCW output channels (blockIdx.y).
template <int KIND, int CIN, int NCW, int VT, int BD, int BHT, int BWT, bool HAS_X2, bool PREP>
__device__ __forceinline__ void os_block(const ConvArgs& a, int nbh, int nbw, int blk, int nblk, int cog, float* lds) {
    using Gm = OsGeom<KIND, CIN, NCW, VT, BD, BHT, BWT>;
    constexpr int S = Gm::S, PH = Gm::PH, PW = Gm::PW, CQ = Gm::CQ, NIT = Gm::NIT, G = Gm::G, NS = Gm::NS;
    constexpr int NACC = Gm::NACC;
    constexpr int PF = 8;                                   // weight loads in flight per wave (K steps ahead; 4 / 8 / 12 measured: 8)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int ct = wave % NCW, wv = wave / NCW;             // cout tile / voxel-tile group of this wave

    // block -> origin in OUTPUT voxels (S1, S2) or INPUT voxels (transposed)
    const int bid = xcd_swizzle(blk, nblk);
    const int bx = bid % nbw, by = (bid / nbw) % nbh, bz = bid / (nbw * nbh);
    const int z0 = bz * BD, y0 = by * Gm::BH, x0 = bx * Gm::BW;
    const int co_base = (cog * NCW + ct) * 16;              // first output channel of this wave's tile
    // input coordinate of staged position (0,0,0)
    const int iz0 = KIND == OS_S1 ? z0 - 1 : KIND == OS_S2 ? 2 * z0 - a.pd : z0 - 1;
    const int iy0 = KIND == OS_S1 ? y0 - 1 : KIND == OS_S2 ? 2 * y0 - a.ph : y0 - 1;
    const int ix0 = KIND == OS_S1 ? x0 - 1 : KIND == OS_S2 ? 2 * x0 - a.pw : x0 - 1;

    // ---- weights: this wave's stream of A fragments -------------------------------------------------------------
    // prepared: [cout tile][step][lane][4] floats, lane (r = lane&15, q) <-> w[tap][16g + 4q + i][16*tile + r]
    const float* wp = PREP ? a.wprep + ((size_t)(cog * NCW + ct) * NS) * 256 + lane * 4 : nullptr;
    auto load_a = [&](int s) __attribute__((always_inline)) -> float4 {
        if (PREP) return *reinterpret_cast<const float4*>(wp + (size_t)s * 256);
        const int tap = s / G, g = s - tap * G;
        const int ci = 16 * g + 4 * q, co = co_base + n;
        if (KIND == OS_DECONV)                              // TensorFlow (3,3,3,Cout,Cin)
            return *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.cout_total + co) * CIN + ci);
        const float* p = a.w + ((size_t)tap * CIN + ci) * a.cout_total + co;      // TensorFlow (3,3,3,Cin,Cout)
        return make_float4(p[0], p[a.cout_total], p[2 * a.cout_total], p[3 * a.cout_total]);
    };
    // the producers' float64 BatchNorm sums first: they come from memory-side atomics (an L2 miss) and head the longest
    // dependent chain of the prologue (sums -> float64 scale / shift -> staged values -> LDS -> barrier)
    BnSums4 bs1 = {}, bs2 = {};
    if (!a.xs && a.bn.stats) bs1 = bn_sums4(a.bn, 4 * (tid % CQ));
    if (HAS_X2 && !a.x2s && a.bn2.stats) bs2 = bn_sums4(a.bn2, 4 * (tid % CQ));
    float4 areg[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) areg[s] = load_a(s);       // in flight under the staging below

    // ---- stage the input block: BN + ReLU (+ skip) on load, zeros outside the volume (SAME padding) -----------
    {
        const int c4 = tid % CQ;
        const int xbytes = a.D * a.H * a.W * CIN * 4;       // host-checked < 2^31
        const auto xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xbytes, 0x00020000);
        const auto x2r = __builtin_amdgcn_make_buffer_rsrc((void*)(HAS_X2 ? a.x2 : a.x), 0, xbytes, 0x00020000);
        u32x4_t v[NIT], v2[HAS_X2 ? NIT : 1];
        unsigned ok_mask = 0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = tid + 256 * i;
            const int pos = f / CQ;
            const int pz = pos / (PH * PW), pr = pos - pz * (PH * PW);
            const int py = pr / PW, px = pr - py * PW;
            const int gz = iz0 + pz, gy = iy0 + py, gx = ix0 + px;
            const bool ok = f < Gm::NF4 && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const int off = ok ? (((gz * a.H + gy) * a.W + gx) * CIN + 4 * c4) * 4 : OS_OOB;
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
            if (HAS_X2) v2[i] = __builtin_amdgcn_raw_buffer_load_b128(x2r, off, 0, 0);
            ok_mask |= ok ? (1u << i) : 0u;
        }
        // the producers' BatchNorm (float64 sums -> scale, shift) while the loads above are in flight
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 sc2 = sc, sh2 = sh;
        const bool aff = a.xs != nullptr || a.bn.stats != nullptr;
        if (a.xs) { sc = *(const float4*)(a.xs + 4 * c4); sh = *(const float4*)(a.xb + 4 * c4); }
        else if (a.bn.stats) bn_affine4_from(a.bn, bs1, sc, sh);
        const bool aff2 = HAS_X2 && (a.x2s != nullptr || a.bn2.stats != nullptr);
        if (HAS_X2 && a.x2s) { sc2 = *(const float4*)(a.x2s + 4 * c4); sh2 = *(const float4*)(a.x2b + 4 * c4); }
        else if (HAS_X2 && a.bn2.stats) bn_affine4_from(a.bn2, bs2, sc2, sh2);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int f = tid + 256 * i;
            if (f < Gm::NF4) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok_mask & (1u << i)) {
                    o = bn_relu4(make_float4(__uint_as_float(v[i][0]), __uint_as_float(v[i][1]), __uint_as_float(v[i][2]), __uint_as_float(v[i][3])), sc, sh, aff);
                    if (HAS_X2) {
                        const float4 o2 = bn_relu4(make_float4(__uint_as_float(v2[i][0]), __uint_as_float(v2[i][1]), __uint_as_float(v2[i][2]), __uint_as_float(v2[i][3])), sc2, sh2, aff2);
                        o.x += o2.x; o.y += o2.y; o.z += o2.z; o.w += o2.w;
                    }
                }
                *reinterpret_cast<float4*>(lds + (f / CQ) * S + 4 * c4) = o;
            }
        }
    }
    __syncthreads();

    // ---- K loop ---------------------------------------------------------------------------------------------------
    // lane base of the B operand per voxel tile: position of voxel n of the tile for tap (0,0,0) (transposed: for the
    // tap shifted by -1 on every axis), channel quad q
    const float* bp[VT];
    int tz[VT], ty[VT], tx[VT];
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int t = wv * VT + i;
        tz[i] = t / (BHT * BWT); ty[i] = 4 * ((t / BWT) % BHT) + (n >> 2); tx[i] = 4 * (t % BWT) + (n & 3);
        const int pos = KIND == OS_S2 ? (2 * tz[i] * PH + 2 * ty[i]) * PW + 2 * tx[i] : (tz[i] * PH + ty[i]) * PW + tx[i];
        bp[i] = lds + pos * S + 4 * q;
    }
    // VT = 1 (one voxel tile per wave: small workgroups, several per CU): two consecutive K steps run interleaved on
    // two accumulator chains, summed at the end -- four MFMAs back to back on ONE accumulator would issue every 40
    // cycles (dependent latency) instead of every 32.
    constexpr int NCH = VT == 1 ? 2 : 1;
    f32x4 acc[NACC][VT][NCH];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < VT; ++i)
#pragma unroll
            for (int h = 0; h < NCH; ++h) acc[c][i][h] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline, pinned with scheduling fences (left alone, the scheduler sinks the weight loads to one step
    // before their use and runs the four MFMAs of an accumulator back to back): a group of NCH steps issues the weight
    // loads PF steps ahead and the LDS reads of the next group, then its MFMAs with the accumulators interleaved.
    auto b_off = [&](int s) __attribute__((always_inline)) -> int {
        const int tap = s / G, g = s % G;
        const int kd = tap_kd(tap), kh = tap_kh(tap), kw = tap_kw(tap);
        // transposed: k = 0 -> (even, i = m), k = 2 -> (even, i = m-1), k = 1 -> (odd, i = m); staged position 0 is m0-1
        const int dz = KIND == OS_DECONV ? (kd == 2 ? 0 : 1) : kd;
        const int dy = KIND == OS_DECONV ? (kh == 2 ? 0 : 1) : kh;
        const int dx = KIND == OS_DECONV ? (kw == 2 ? 0 : 1) : kw;
        return ((dz * PH + dy) * PW + dx) * S + 16 * g;
    };
    float4 bv[2][NCH][VT];
#pragma unroll
    for (int h = 0; h < NCH; ++h)
#pragma unroll
        for (int i = 0; i < VT; ++i)
            if (h < NS) bv[0][h][i] = *reinterpret_cast<const float4*>(bp[i] + b_off(h));
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += NCH) {
        const int grp = s0 / NCH;
        float4 av[NCH];
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            const int s = s0 + h;
            if (s < NS) {
                av[h] = areg[s % PF];
                if (s + PF < NS) areg[s % PF] = load_a(s + PF);
            }
        }
#pragma unroll
        for (int h = 0; h < NCH; ++h)
#pragma unroll
            for (int i = 0; i < VT; ++i)
                if (s0 + NCH + h < NS) bv[(grp + 1) & 1][h][i] = *reinterpret_cast<const float4*>(bp[i] + b_off(s0 + NCH + h));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                const int s = s0 + h;
                if (s < NS) {
                    const int tap = s / G;
                    const int cls = KIND == OS_DECONV ? ((tap_kd(tap) == 1) * 4 + (tap_kh(tap) == 1) * 2 + (tap_kw(tap) == 1)) : 0;
                    const float ae = k == 0 ? av[h].x : k == 1 ? av[h].y : k == 2 ? av[h].z : av[h].w;
#pragma unroll
                    for (int i = 0; i < VT; ++i) {
                        const float4 b4 = bv[grp & 1][h][i];
                        const float be = k == 0 ? b4.x : k == 1 ? b4.y : k == 2 ? b4.z : b4.w;
                        acc[cls][i][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae, be, acc[cls][i][h], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (NCH == 2) {
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int i = 0; i < VT; ++i) acc[c][i][0] += acc[c][i][NCH - 1];
    }

    // ---- epilogue: raw outputs (float4 = 4 consecutive couts of voxel n) + BatchNorm sums -------------------------
    const int Do = KIND == OS_S1 ? a.D : KIND == OS_S2 ? (a.D + 1) / 2 : 2 * a.D;
    const int Ho = KIND == OS_S1 ? a.H : KIND == OS_S2 ? (a.H + 1) / 2 : 2 * a.H;
    const int Wo = KIND == OS_S1 ? a.W : KIND == OS_S2 ? (a.W + 1) / 2 : 2 * a.W;
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VT; ++i) {
        const int vz = z0 + tz[i], vy = y0 + ty[i], vx = x0 + tx[i];
#pragma unroll
        for (int c = 0; c < NACC; ++c) {
            const int oz = KIND == OS_DECONV ? 2 * vz + (c >> 2) : vz;
            const int oy = KIND == OS_DECONV ? 2 * vy + ((c >> 1) & 1) : vy;
            const int ox = KIND == OS_DECONV ? 2 * vx + (c & 1) : vx;
            if (oz < Do && oy < Ho && ox < Wo) {
                const f32x4 r = acc[c][i][0];
                *reinterpret_cast<float4*>(a.y + ((size_t)(oz * Ho + oy) * Wo + ox) * a.cout_total + co_base + 4 * q) =
                    make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) { st_s[k] += r[k]; st_q[k] += r[k] * r[k]; }
            }
        }
    }
    if (a.stats) {
        // fold the 16 voxel lanes, then the waves that share a cout tile, then one f64 atomic per channel and workgroup
        // (one atomic instruction of 32 * NCW lanes: per-wave atomics without the barriers -- 8 instructions of 4 lanes per
        // wave -- measured 15 us SLOWER on the 240-workgroup layers)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { st_s[k] += __shfl_xor(st_s[k], o, 64); st_q[k] += __shfl_xor(st_q[k], o, 64); }
        __syncthreads();                                    // every wave is done reading the staged block
        float* red = lds;                                   // [wave][2][16]
        if (n == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[(wave * 2 + 0) * 16 + 4 * q + k] = st_s[k]; red[(wave * 2 + 1) * 16 + 4 * q + k] = st_q[k]; }
        }
        __syncthreads();
        if (tid < 2 * 16 * NCW) {
            const int c = tid & 15, kk = (tid >> 4) & 1, t = tid >> 5;       // channel in tile, sum / sum of squares, cout tile
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < Gm::NVW; ++w) tot += (double)red[((w * NCW + t) * 2 + kk) * 16 + c];
            atomicAdd(&conv_stats_row(a)[(size_t)kk * a.cout_total + (cog * NCW + t) * 16 + c], tot);
        }
    }
}

template <int KIND, int CIN, int NCW, int VT, int BD, int BHT, int BWT, bool HAS_X2, bool PREP>
__global__ void __launch_bounds__(256, 2)
conv3d_os_kernel(ConvArgs a, int nbh, int nbw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    os_block<KIND, CIN, NCW, VT, BD, BHT, BWT, HAS_X2, PREP>(a, nbh, nbw, blockIdx.x, gridDim.x, blockIdx.y, lds);
}

// A chain layer of the 1/8 level + FILLER blocks of 3dconv2_1.  3dconv3_0, 3_1 and 4_0 (mvsnetworks.py:133-147) are 240
// workgroups each -- one per CU, four waves -- that spend half of their 16-23 us in fixed cost (BatchNorm sums, staging, the
// atomics tail) with the matrix pipe idle, and 3dconv2_1 (960 blocks, 3.4 GFLOP), which only the decoder's 3dconv5_0 needs,
// used to run alone before them.  Here every launch of the chain carries a slice of 3dconv2_1's blocks behind its own: the
// first `na` workgroups are layer A's blocks, the rest blocks [b_first, b_first + gridDim.x - na) of layer B's `nb`.
template <class CA, class CB>
__global__ void __launch_bounds__(256, 2)
conv3d_os_filled_kernel(ConvArgs a, int nbh_a, int nbw_a, int na, ConvArgs b, int nbh_b, int nbw_b, int nb, int b_first) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < na)
        os_block<CA::KIND, CA::CIN, CA::NCW, CA::VT, CA::BD, CA::BHT, CA::BWT, false, true>(a, nbh_a, nbw_a, blockIdx.x, na, 0, lds);
    else
        os_block<CB::KIND, CB::CIN, CB::NCW, CB::VT, CB::BD, CB::BHT, CB::BWT, false, true>(b, nbh_b, nbw_b, b_first + blockIdx.x - na, nb, 0, lds);
}

template <int KIND_, int CIN_, int NCW_, int VT_, int BD_, int BHT_, int BWT_>
struct OsCfg {
    static constexpr int KIND = KIND_, CIN = CIN_, NCW = NCW_, VT = VT_, BD = BD_, BHT = BHT_, BWT = BWT_;
    using Gm = OsGeom<KIND_, CIN_, NCW_, VT_, BD_, BHT_, BWT_>;
    // block grid over OUTPUT voxels (S1, S2) or INPUT voxels (transposed)
    static void grid(const ConvArgs& a, int& nbd, int& nbh, int& nbw) {
        const int Dg = KIND == OS_S2 ? (a.D + 1) / 2 : a.D, Hg = KIND == OS_S2 ? (a.H + 1) / 2 : a.H,
                  Wg = KIND == OS_S2 ? (a.W + 1) / 2 : a.W;
        nbd = (Dg + BD - 1) / BD; nbh = (Hg + Gm::BH - 1) / Gm::BH; nbw = (Wg + Gm::BW - 1) / Gm::BW;
    }
};
using Os21 = OsCfg<OS_S1, 32, 2, 2, 2, 1, 2>;           // 3dconv2_1
using Os30 = OsCfg<OS_S2, 32, 4, 2, 2, 1, 1>;           // 3dconv3_0
using Os31 = OsCfg<OS_S1, 64, 4, 2, 2, 1, 1>;           // 3dconv3_1
using Os40 = OsCfg<OS_DECONV, 64, 2, 1, 2, 1, 1>;       // 3dconv4_0

template <class CA, class CB>
int launch_os_filled(const ConvArgs& a, int CoutA, const ConvArgs& b, int CoutB, int b_first, int b_count, hipStream_t st) {
    if (CoutA != 16 * CA::NCW || CoutB != 16 * CB::NCW) return MVS_E_SHAPE;          // one output-channel group each
    if (!a.wprep || !b.wprep || a.x2 || b.x2) return MVS_E_SHAPE;
    const long long via = (long long)a.D * a.H * a.W, voa = CA::KIND == OS_DECONV ? 8 * via : via, vb = (long long)b.D * b.H * b.W;
    if (via * CA::CIN * 4 >= (1LL << 31) || voa * CoutA * 4 >= (1LL << 31) || vb * CB::CIN * 4 >= (1LL << 31) ||
        vb * CoutB * 4 >= (1LL << 31)) return MVS_E_SHAPE;                            // 32-bit offsets
    int nbd, nbh_a, nbw_a, nbh_b, nbw_b;
    CA::grid(a, nbd, nbh_a, nbw_a);
    const int na = nbd * nbh_a * nbw_a;
    CB::grid(b, nbd, nbh_b, nbw_b);
    const int nb = nbd * nbh_b * nbw_b;
    if (b_first < 0 || b_count < 0 || b_first + b_count > nb) return MVS_E_SHAPE;
    const size_t smem = CA::Gm::LDS_BYTES > CB::Gm::LDS_BYTES ? CA::Gm::LDS_BYTES : CB::Gm::LDS_BYTES;
    static bool attr_done = false;
    auto* kern = conv3d_os_filled_kernel<CA, CB>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    kern<<<dim3(na + b_count), 256, smem, st>>>(a, nbh_a, nbw_a, na, b, nbh_b, nbw_b, nb, b_first);
    return (int)hipGetLastError();
}

template <int KIND, int CIN, int NCW, int VT, int BD, int BHT, int BWT>
int launch_os(const ConvArgs& a0, int Cout, hipStream_t st) {
    using Gm = OsGeom<KIND, CIN, NCW, VT, BD, BHT, BWT>;
    ConvArgs a = a0;
    if (Cout % (16 * NCW)) return MVS_E_SHAPE;
    const long long vin = (long long)a.D * a.H * a.W, vout = KIND == OS_DECONV ? 8 * vin : vin;
    if (vin * CIN * 4 >= (1LL << 31) || vout * Cout * 4 >= (1LL << 31)) return MVS_E_SHAPE;      // 32-bit offsets
    // block grid over OUTPUT voxels (S1, S2) or INPUT voxels (transposed)
    const int Dg = KIND == OS_S2 ? (a.D + 1) / 2 : a.D, Hg = KIND == OS_S2 ? (a.H + 1) / 2 : a.H,
              Wg = KIND == OS_S2 ? (a.W + 1) / 2 : a.W;
    const int nbd = (Dg + BD - 1) / BD, nbh = (Hg + Gm::BH - 1) / Gm::BH, nbw = (Wg + Gm::BW - 1) / Gm::BW;
    dim3 grid(nbd * nbh * nbw, Cout / (16 * NCW), 1);
    const size_t smem = Gm::LDS_BYTES;
    const bool prep = a.wprep != nullptr;
#define OS_LAUNCH(X2, PR)                                                                                          \
    do {                                                                                                           \
        static bool attr_done = false;                                                                             \
        auto* kern = conv3d_os_kernel<KIND, CIN, NCW, VT, BD, BHT, BWT, X2, PR>;                                   \
        if (!attr_done) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess) return (int)e;                                                                    \
            attr_done = true;                                                                                      \
        }                                                                                                          \
        kern<<<grid, 256, smem, st>>>(a, nbh, nbw);                                                                \
    } while (0)
    if (a.x2) { if (prep) OS_LAUNCH(true, true); else OS_LAUNCH(true, false); }
    else { if (prep) OS_LAUNCH(false, true); else OS_LAUNCH(false, false); }
#undef OS_LAUNCH
    return (int)hipGetLastError();
}

// out[((tile*NS + tap*G + g)*64 + lane)*4 + i] = W[tap][ci = 16g + 4(lane>>4) + i][co = 16*tile + (lane&15)]
__global__ void os_weight_layout_kernel(const float* __restrict__ w, int transposed, int Cin, int Cout, float* __restrict__ out) {
    const int total = 27 * Cin * Cout;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = idx & 3, lane = (idx >> 2) & 63;
    int r = idx >> 8;
    const int G = Cin / 16, NS = 27 * G;
    const int s = r % NS, tile = r / NS;
    const int tap = s / G, g = s - tap * G;
    const int ci = 16 * g + 4 * (lane >> 4) + i, co = 16 * tile + (lane & 15);
    out[idx] = transposed ? w[((size_t)tap * Cout + co) * Cin + ci] : w[((size_t)tap * Cin + ci) * Cout + co];
}

}  // namespace

// Which layers take this family: the faster one of the two at the metric workload (round 2, in-pipeline microseconds
// plane-march kernel of conv3d_mfma.hip / conv3d_s2_mfma.hip / deconv3d_*.hip -> block kernel): 3dconv3_1 38.9 -> 27.5, 2_1 46.3 -> 42.4,
// 3_0 28.1 -> 21.3, 4_0 35.4 -> 21.5, 5_0 43.9 -> 34.0; 3dconv2_0 (16 -> 32, K = 432) 37.0 -> 41.3 and 3dconv1_1
// (16 -> 16) 75 -> 108 stay on the plane march: with 16 input channels the work per staged byte is too small for a
// stage-then-compute block.  One voxel tile per wave (smaller, more numerous workgroups) measured slower everywhere.
static int os_variant_of(int kind, int Cin, int Cout) {
    constexpr int v31 = 0, v21 = 0, v11 = -1, v30 = 0, v20 = -1, v40 = 0, v50 = 0;      // 0 = this family, -1 = plane march
    // network_mode 'fat' (base_filter 16, network.py:82-83): the four layers no plane-march kernel is built for
    if ((kind == 0 && Cin == 128 && Cout == 128) || (kind == 1 && Cin == 64 && (Cout == 32 || Cout == 128)) ||
        (kind == 2 && Cin == 128 && Cout == 64)) return 0;
    if (kind == 0) return (Cin == 64 && Cout == 64) ? v31 : (Cin == 32 && Cout == 32) ? v21 : (Cin == 16 && Cout == 16) ? v11 : -1;
    if (kind == 1) return (Cin == 32 && Cout == 64) ? v30 : (Cin == 16 && Cout == 32) ? v20 : -1;
    return (Cin == 64 && Cout == 32) ? v40 : (Cin == 32 && Cout == 16) ? v50 : -1;
}

// Which (kind, Cin, Cout) layers run on this family; the weight pre-layout (regnet.hip) asks the same question.
// kind: 0 = stride 1, 1 = stride 2, 2 = transposed.
bool mvs_conv3d_os_covers(int kind, int Cin, int Cout) { return os_variant_of(kind, Cin, Cout) >= 0; }

int mvs_conv3d_os_weight_layout(const float* w, int kind, int Cin, int Cout, float* out, hipStream_t st) {
    if ((Cin % 16) || (Cout % 16)) return MVS_E_SHAPE;
    const int total = 27 * Cin * Cout;
    os_weight_layout_kernel<<<mvs_cdiv(total, 256), 256, 0, st>>>(w, kind == 2, Cin, Cout, out);
    return (int)hipGetLastError();
}

// Blocks of the 3dconv2_1-shaped layer (32 -> 32, stride 1) a caller can deal out as fillers.
int mvs_conv3d_os_filler_blocks(int D, int H, int W) {
    ConvArgs b{}; b.D = D; b.H = H; b.W = W;
    int nbd, nbh, nbw;
    Os21::grid(b, nbd, nbh, nbw);
    return nbd * nbh * nbw;
}

// Layer A (kind / Cin / Cout of 3dconv3_0, 3_1 or 4_0) with blocks [b_first, b_first + b_count) of the 32 -> 32 stride-1 layer b.
// MVS_E_SHAPE: not one of the three pairs (the caller launches the layers apart).
int mvs_conv3d_os_filled_launch(const ConvArgs& a0, int kind, int Cin, int Cout, const ConvArgs& b, int b_first, int b_count,
                                hipStream_t st) {
    ConvArgs a = a0;
    if (kind == 1) {
        auto pad_before = [](int n) { int o = (n + 1) / 2; int t = (o - 1) * 2 + 3 - n; return t < 0 ? 0 : t / 2; };
        a.pd = pad_before(a.D); a.ph = pad_before(a.H); a.pw = pad_before(a.W);
    }
    if (kind == 1 && Cin == 32 && Cout == 64) return launch_os_filled<Os30, Os21>(a, Cout, b, 32, b_first, b_count, st);
    if (kind == 0 && Cin == 64 && Cout == 64) return launch_os_filled<Os31, Os21>(a, Cout, b, 32, b_first, b_count, st);
    if (kind == 2 && Cin == 64 && Cout == 32) return launch_os_filled<Os40, Os21>(a, Cout, b, 32, b_first, b_count, st);
    return MVS_E_SHAPE;
}

int mvs_conv3d_os_launch(const ConvArgs& a, int kind, int Cin, int Cout, hipStream_t st) {
    //                                                          KIND    CIN NCW VT BD BHT BWT
    if (kind == 0 && Cin == 64 && Cout == 64) return launch_os<OS_S1, 64, 4, 2, 2, 1, 1>(a, Cout, st);      // 3dconv3_1
    if (kind == 0 && Cin == 32 && Cout == 32) return launch_os<OS_S1, 32, 2, 2, 2, 1, 2>(a, Cout, st);      // 3dconv2_1
    if (kind == 1 && Cin == 32 && Cout == 64) return launch_os<OS_S2, 32, 4, 2, 2, 1, 1>(a, Cout, st);      // 3dconv3_0
    if (kind == 2 && Cin == 64 && Cout == 32) return launch_os<OS_DECONV, 64, 2, 1, 2, 1, 1>(a, Cout, st);  // 3dconv4_0
    if (kind == 2 && Cin == 32 && Cout == 16) return launch_os<OS_DECONV, 32, 1, 2, 2, 2, 2>(a, Cout, st);  // 3dconv5_0
    // 'fat' (64-channel volume, base_filter 16): 3dconv1_0, 3_0, 3_1, 4_0
    if (kind == 1 && Cin == 64 && Cout == 32) return launch_os<OS_S2, 64, 2, 1, 2, 1, 1>(a, Cout, st);
    if (kind == 1 && Cin == 64 && Cout == 128) return launch_os<OS_S2, 64, 4, 2, 2, 1, 1>(a, Cout, st);
    if (kind == 0 && Cin == 128 && Cout == 128) return launch_os<OS_S1, 128, 4, 2, 2, 1, 1>(a, Cout, st);
    if (kind == 2 && Cin == 128 && Cout == 64) return launch_os<OS_DECONV, 128, 2, 1, 2, 1, 1>(a, Cout, st);
    return MVS_E_SHAPE;
}
