// Weight gradient of a 3x3 SAME stride-1 2D convolution over a batch of planes on the fp32 matrix cores:
//
//      dW[kh][kw][ci][co] = sum over n, y, x of  X[n][y + kh - 1][x + kw - 1][ci] * G[n][y][x][co]
//
// (TensorFlow's Conv2DBackpropFilter behind opt.compute_gradients, mvsnet/train.py:428-429, of the tf.layers.conv2d
// calls in mvsnet/convgru.py:92,110 when the recurrent model trains: the planes of the depth sweep are the batch.)
// The pixels are the contraction: v_mfma_f32_16x16x4_f32 takes 4 consecutive pixels of a row per instruction, rows =
// 16 input channels of one tap, columns = 16 output channels.  A (tap, input-channel tile) pair is a unit; the 9*CI_T
// units are dealt over the 4 waves, every unit keeps CO_T accumulator tiles.  A workgroup stages a 4 x 32 pixel tile
// of G and its 6 x 34 halo tile of X in LDS (C + 8 floats per position: the 64 b32 reads of an operand, 16 channels x
// 4 pixels, then fall 2 per bank), walks its 32 pixel quads, and moves on to the next tile of its grid-stride
// sequence with the accumulators still in registers; one partial dW per workgroup, summed in float64 in a fixed
// order by a second kernel: deterministic, no atomics.  G may be a channel slice of a wider tensor (g_stride, g_off).
// Roofline: MFMA, 2*9*Cin*Cout flops per pixel.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int W2_TH = 4, W2_TW = 32, W2_PH = W2_TH + 2, W2_PW = W2_TW + 2;

template <int CI_T, int CO_T>
__global__ void __launch_bounds__(256)
wgrad2d_kernel(const float* __restrict__ x, const float* __restrict__ g, int g_stride, int g_off, int N, int H, int W,
               float* __restrict__ partial) {
    constexpr int CIN = 16 * CI_T, COUT = 16 * CO_T, SX = CIN + 8, SG = COUT + 8;
    constexpr int UNITS = 9 * CI_T, UPW = (UNITS + 3) / 4;
    __shared__ __attribute__((aligned(16))) float xs[W2_PH * W2_PW * SX];
    __shared__ __attribute__((aligned(16))) float gs[W2_TH * W2_TW * SG];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;
    f32x4 acc[UPW][CO_T];
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[u][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave's units: LDS offset of the operand's tap and channel tile
    int a_off[UPW];
#pragma unroll
    for (int ui = 0; ui < UPW; ++ui) {
        const int u = wave + 4 * ui, tap = u / CI_T, cit = u - tap * CI_T, kh = tap / 3, kw = tap - 3 * kh;
        a_off[ui] = u < UNITS ? (kh * W2_PW + kw) * SX + cit * 16 : -1;
    }
    const int tiles_w = (W + W2_TW - 1) / W2_TW, tiles_h = (H + W2_TH - 1) / W2_TH;
    const int ntiles = N * tiles_h * tiles_w;
    // staging map of this thread (the same for every tile): NX float4 of the X halo tile, NG float4 of the G tile
    constexpr int XQ = CIN / 4, GQ = COUT / 4;
    constexpr int NX = (W2_PH * W2_PW * XQ + 255) / 256, NG = (W2_TH * W2_TW * GQ + 255) / 256;
    float4 vx[NX], vg[NG];
    // the next tile's operands are requested before the current tile's MFMAs and land in LDS after them
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int n = t / (tiles_h * tiles_w), r0 = t - n * tiles_h * tiles_w;
        const int y0 = (r0 / tiles_w) * W2_TH, x0 = (r0 % tiles_w) * W2_TW;
        const float* xn = x + (size_t)n * H * W * CIN;
        const float* gn = g + (size_t)n * H * W * g_stride + g_off;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int f = tid + 256 * i, pos = f / XQ, c4 = f - pos * XQ;
            const int r = pos / W2_PW, c = pos - r * W2_PW, gy = y0 - 1 + r, gx = x0 - 1 + c;
            vx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < W2_PH * W2_PW * XQ && gy >= 0 && gy < H && gx >= 0 && gx < W)
                vx[i] = *(const float4*)(xn + ((size_t)gy * W + gx) * CIN + 4 * c4);
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int f = tid + 256 * i, pos = f / GQ, c4 = f - pos * GQ;
            const int r = pos / W2_TW, c = pos - r * W2_TW, gy = y0 + r, gx = x0 + c;
            vg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < W2_TH * W2_TW * GQ && gy < H && gx < W)
                vg[i] = *(const float4*)(gn + ((size_t)gy * W + gx) * g_stride + 4 * c4);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int f = tid + 256 * i, pos = f / XQ, c4 = f - pos * XQ;
            if (f < W2_PH * W2_PW * XQ) *(float4*)(xs + pos * SX + 4 * c4) = vx[i];
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int f = tid + 256 * i, pos = f / GQ, c4 = f - pos * GQ;
            if (f < W2_TH * W2_TW * GQ) *(float4*)(gs + pos * SG + 4 * c4) = vg[i];
        }
    };
    if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        commit();
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) issue(t + gridDim.x);
#pragma unroll 2
        for (int q = 0; q < W2_TH * W2_TW / 4; ++q) {
            const int row = q / (W2_TW / 4), w0 = (q - row * (W2_TW / 4)) * 4;
            float b[CO_T];
#pragma unroll
            for (int c = 0; c < CO_T; ++c) b[c] = gs[(row * W2_TW + w0 + kq) * SG + c * 16 + n16];
            const float* xa = xs + (row * W2_PW + w0 + kq) * SX + n16;
#pragma unroll
            for (int ui = 0; ui < UPW; ++ui) {
                if (a_off[ui] < 0) continue;                               // wave-uniform
                const float a = xa[a_off[ui]];
#pragma unroll
                for (int c = 0; c < CO_T; ++c) acc[ui][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[ui][c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* out = partial + (size_t)blockIdx.x * 9 * CIN * COUT;
#pragma unroll
    for (int ui = 0; ui < UPW; ++ui) {
        const int u = wave + 4 * ui;
        if (u >= UNITS) continue;
        const int tap = u / CI_T, cit = u - tap * CI_T;
#pragma unroll
        for (int c = 0; c < CO_T; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[((size_t)tap * CIN + cit * 16 + 4 * kq + r) * COUT + c * 16 + n16] = acc[ui][c][r];
    }
}

__global__ void __launch_bounds__(256)
wgrad2d_reduce_kernel(const float* __restrict__ partial, int blocks, int n, float* __restrict__ dw) {
    __shared__ double fold[4][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (o < n) {
        int b = part;
        for (; b + 12 < blocks; b += 16) {                          // four loads in flight
            s0 += (double)partial[(size_t)b * n + o];
            s1 += (double)partial[(size_t)(b + 4) * n + o];
            s2 += (double)partial[(size_t)(b + 8) * n + o];
            s3 += (double)partial[(size_t)(b + 12) * n + o];
        }
        for (; b < blocks; b += 4) s0 += (double)partial[(size_t)b * n + o];
    }
    fold[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0 && o < n) dw[o] = (float)((fold[0][threadIdx.x] + fold[1][threadIdx.x]) + (fold[2][threadIdx.x] + fold[3][threadIdx.x]));
}

template <int CI_T, int CO_T>
int w2_per_cu() {
    static int v = 0;                                               // workgroups of this instance one CU holds
    if (!v) {
        int k = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, wgrad2d_kernel<CI_T, CO_T>, 256, 0) != hipSuccess || k < 1) k = 1;
        v = k > 4 ? 4 : k;
    }
    return v;
}

constexpr int W2_MAX_BLOCKS = 1024;                                // partial-sum slots the workspace is sized for
int w2_blocks(int N, int H, int W, int per_cu) {
    const long long ntiles = (long long)N * ((H + W2_TH - 1) / W2_TH) * ((W + W2_TW - 1) / W2_TW);
    const long long want = 256LL * per_cu;                          // one resident round: no tail of half-empty CUs
    return (int)(ntiles < want ? ntiles : want);
}

}  // namespace

extern "C" size_t mvs_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)w2_blocks(N, H, W, W2_MAX_BLOCKS / 256) * 9 * Cin * Cout * sizeof(float);
}

extern "C" int mvs_conv2d_wgrad_f32(const float* x, const float* g, int g_stride, int g_off, int N, int H, int W,
                                    int Cin, int Cout, void* workspace, size_t workspace_bytes, float* dw,
                                    void* stream) {
    MVS_CHECK_ARG(x && g && workspace && dw && N > 0 && H > 0 && W > 0 && g_stride >= g_off + Cout && g_off >= 0);
    if (g_stride % 4 || g_off % 4) return MVS_E_SHAPE;
    if (workspace_bytes < mvs_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout)) return MVS_E_WORKSPACE;
    hipStream_t st = mvs_stream(stream);
    int blocks = 0;
    float* partial = (float*)workspace;
#define W2(CI, CO) if (Cin == 16 * CI && Cout == 16 * CO) { \
        blocks = w2_blocks(N, H, W, w2_per_cu<CI, CO>()); \
        wgrad2d_kernel<CI, CO><<<blocks, 256, 0, st>>>(x, g, g_stride, g_off, N, H, W, partial); } else
    W2(2, 3) W2(1, 2) W2(1, 1) W2(2, 2) W2(2, 1) W2(1, 3)
        return MVS_E_SHAPE;
#undef W2
    const int n = 9 * Cin * Cout;
    wgrad2d_reduce_kernel<<<mvs_cdiv(n, 64), 256, 0, st>>>(partial, blocks, n, dw);
    MVS_LAUNCH_RET();
}
