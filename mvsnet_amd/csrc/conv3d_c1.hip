// The two one-channel operations on the training path of 3dconv6_2 (8 -> 1, no BN; mvsnetworks.py:156-158):
//   * its input gradient, a 3x3x3 stride-1 convolution 1 -> Cout (the output gradient g_reg has ONE channel);
//   * its weight gradient  dW(tap, cb) = sum_v s6(v + tap - 1, cb) * g_reg(v)  with one "small" channel.
// One channel fills 1/16 of an MFMA tile either way (the generic MFMA kernels took 0.33 ms and 0.50 ms at
// D=192, 120x160) and both are HBM/LDS-bound: VALU kernels with the input-stationary plane march of
// conv3d_out.hip (8 x 32 tile, planes staged once in LDS, weights through the scalar cache).
#include "conv_common.h"

namespace {

constexpr int TH = 8, TW = 32, PW = TW + 2, PH = TH + 2;
typedef const __attribute__((address_space(4))) float cfloat;

// ---- y(v, co) = sum_tap x(v + tap - 1) * w(tap, 0, co):  x (D,H,W,1), w (3,3,3,1,COUT), y (D,H,W,COUT) -------------
template <int COUT>
__global__ void __launch_bounds__(256)
conv3d_in1_kernel(ConvArgs a) {
    __shared__ float slab[2][PH * PW];
    cfloat* wsh = (cfloat*)(a.w);
    const int tid = threadIdx.x, row = tid >> 5, col = tid & 31;
    const int tiles_w = (a.W + TW - 1) / TW;
    const int tile_h = blockIdx.x / tiles_w, tile_w = blockIdx.x - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TW;
    const int d0 = blockIdx.z * a.planes_per_wg, d1 = min(d0 + a.planes_per_wg, a.D);
    const int T = d1 - d0 + 2;
    constexpr int NPOS = PH * PW, NIT = (NPOS + 255) / 256;
    int goff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int f = tid + 256 * i, r = f / PW, c = f - r * PW;
        const int gh = h0 - 1 + r, gw = w0 - 1 + c;
        goff[i] = (f < NPOS && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) ? gh * a.W + gw : -1;
    }
    const size_t plane = (size_t)a.H * a.W;
    float pre[NIT];
    auto issue = [&](int q) __attribute__((always_inline)) {
        const bool ok = q >= 0 && q < a.D;
#pragma unroll
        for (int i = 0; i < NIT; ++i) pre[i] = (ok && goff[i] >= 0) ? a.x[(size_t)q * plane + goff[i]] : 0.f;
    };
    auto stage = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) { const int f = tid + 256 * i; if (f < NPOS) buf[f] = pre[i]; }
    };
    float acc[3][COUT];                                  // kd = 0 (-> q+1), 1 (-> q), 2 (-> q-1)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[k][co] = 0.f;

    issue(d0 - 1); stage(slab[0]);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int q = d0 - 1 + t;
        const float* cur = slab[t & 1];
        const bool more = t + 1 < T;
        if (more) issue(q + 1);
        if (q >= 0 && q < a.D) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float x = cur[(row + kh) * PW + col + kw];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                        for (int co = 0; co < COUT; ++co) acc[kd][co] += x * wsh[(kd * 9 + kh * 3 + kw) * COUT + co];
                }
        }
        const int o = q - 1, h = h0 + row, w = w0 + col;
        if (o >= d0 && o < d1 && h < a.H && w < a.W) {
            float* dst = a.y + (((size_t)o * a.H + h) * a.W + w) * COUT;
#pragma unroll
            for (int c4 = 0; c4 < COUT; c4 += 4)
                *(float4*)(dst + c4) = make_float4(acc[2][c4], acc[2][c4 + 1], acc[2][c4 + 2], acc[2][c4 + 3]);
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) { acc[2][co] = acc[1][co]; acc[1][co] = acc[0][co]; acc[0][co] = 0.f; }
        if (more) stage(slab[(t + 1) & 1]);
        __syncthreads();
    }
}

// ---- dW(kd, kh, kw, cb) = sum_v big(v + tap - 1, cb) * small(v):  big (D,H,W,CB), small (D,H,W,1) -----------------
// blockIdx.y = kd: a thread (one voxel of the 8 x 32 tile per plane) keeps the 9 x CB in-plane sums of its kd,
// the workgroup folds them through wave shuffles + LDS and writes one partial row; wgrad_reduce adds the rows.
template <int CB>
__global__ void __launch_bounds__(256)
wgrad_c1_kernel(const float* __restrict__ big, const float* __restrict__ small, float* __restrict__ partial,
                int D, int H, int W, int planes_per_wg) {
    constexpr int S = CB + 4, CQ = CB / 4, NPOS = PH * PW, NF4 = NPOS * CQ, NIT = (NF4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float slab[NPOS * S];
    __shared__ float red[4][9 * CB];
    const int tid = threadIdx.x, row = tid >> 5, col = tid & 31, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (W + TW - 1) / TW;
    const int tile_h = blockIdx.x / tiles_w, tile_w = blockIdx.x - tile_h * tiles_w;
    const int h0 = tile_h * TH, w0 = tile_w * TW;
    const int kd = blockIdx.y;
    const int d0 = blockIdx.z * planes_per_wg, d1 = min(d0 + planes_per_wg, D);
    int goff[NIT], loff[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int f = tid + 256 * i, pos = f / CQ, c4 = f - pos * CQ;
        const int r = pos / PW, c = pos - r * PW, gh = h0 - 1 + r, gw = w0 - 1 + c;
        goff[i] = (f < NF4 && gh >= 0 && gh < H && gw >= 0 && gw < W) ? (gh * W + gw) * CB + 4 * c4 : -1;
        loff[i] = f < NF4 ? pos * S + 4 * c4 : -1;
    }
    const size_t bplane = (size_t)H * W * CB, splane = (size_t)H * W;
    const int h = h0 + row, w = w0 + col;
    const bool vox_ok = h < H && w < W;
    float acc[9][CB];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[t9][c] = 0.f;

    for (int od = d0; od < d1; ++od) {
        const int q = od + kd - 1;                       // the big plane this kd pairs with small plane od
        if (q < 0 || q >= D) continue;                   // uniform over the workgroup
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (loff[i] >= 0)
                *(float4*)(slab + loff[i]) = goff[i] >= 0 ? *(const float4*)(big + (size_t)q * bplane + goff[i])
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        const float g = vox_ok ? small[(size_t)od * splane + (size_t)h * W + w] : 0.f;
        __syncthreads();
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int cq = 0; cq < CQ; ++cq) {
                    const float4 x = *(const float4*)(slab + ((row + kh) * PW + col + kw) * S + 4 * cq);
                    float* p = &acc[kh * 3 + kw][4 * cq];
                    p[0] += g * x.x; p[1] += g * x.y; p[2] += g * x.z; p[3] += g * x.w;
                }
    }
    // fold the 256 lanes' sums: shuffles inside a wave, LDS across the four waves, fixed order
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            float v = acc[t9][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[wave][t9 * CB + c] = v;
        }
    __syncthreads();
    float* out = partial + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * (27 * CB) + kd * 9 * CB;
    for (int i = tid; i < 9 * CB; i += 256) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

}  // namespace

int mvs_conv3d_in1_launch(const ConvArgs& a0, int Cout, hipStream_t st) {
    if (a0.xs || a0.x2 || a0.bn.stats || a0.stats || a0.wprep) return MVS_E_SHAPE;     // plain convolution only
    ConvArgs a = a0;
    const int tiles = ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    a.planes_per_wg = conv_pick_planes(a.D, tiles, 2, 1024);
    dim3 grid(tiles, 1, (a.D + a.planes_per_wg - 1) / a.planes_per_wg);
    if (Cout == 8) conv3d_in1_kernel<8><<<grid, 256, 0, st>>>(a);
    else if (Cout == 4) conv3d_in1_kernel<4><<<grid, 256, 0, st>>>(a);
    else return MVS_E_SHAPE;
    return (int)hipGetLastError();
}

// partial rows: (tiles * chunks) x (27 * CB) floats; returns the row count through *rows
int mvs_wgrad_c1_plan(int D, int H, int W, int CB, int* rows, int* planes_per_wg) {
    if (CB != 8) return MVS_E_SHAPE;
    const int tiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    int chunks = 1024 / (3 * tiles); if (chunks < 1) chunks = 1; if (chunks > D) chunks = D;
    const int ppw = (D + chunks - 1) / chunks;
    *planes_per_wg = ppw; *rows = tiles * ((D + ppw - 1) / ppw);
    return 0;
}

int mvs_wgrad_c1_launch(const float* big, const float* small, int D, int H, int W, int CB, float* partial, hipStream_t st) {
    int rows, ppw;
    int rc = mvs_wgrad_c1_plan(D, H, W, CB, &rows, &ppw);
    if (rc) return rc;
    const int tiles = ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    dim3 grid(tiles, 3, (D + ppw - 1) / ppw);
    wgrad_c1_kernel<8><<<grid, 256, 0, st>>>(big, small, partial, D, H, W, ppw);
    return (int)hipGetLastError();
}
