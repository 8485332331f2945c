// Per-image, per-channel standardisation of the input images on the device (the towers' input side):
// the reference centres every image on the host before it enters the graph (mvs_data_generation/utils.py:33-38,
//   img.astype(float32); (img - mean(img, (0,1))) / (sqrt(var(img, (0,1))) + 1e-8)   ),
// a session uploads the decoded uint8 images (a quarter of the bytes) and this file does the rest in two launches per batch,
// writing straight into the towers' 4-channel input (channel 3 = 0; mvs_conv2d_gn_f32 reads the image padded 3 -> 4 channels).
//
// Arithmetic: the moments of uint8 data are INTEGER sums (sum <= 255 n, sum of squares <= 65025 n: exact in uint64), so
// mean = S / n and var = Q / n - mean^2 are formed once per image in float64 from exact totals -- no float reduction order to
// argue about; the output is (float(x) - float(mean)) / (float(sqrt(max(var, 0))) + 1e-8f) in float32, the same expression
// (and the same roundings) as the PyTorch restatement inference.center_images_device.  The reference's numpy float32
// reductions run over the two leading axes, where numpy keeps running float32 sums: within ~2e-6 of these outputs on a few
// thousand pixels, ~1e-3 away at 640 x 512 and above (and dependent on the numpy build); the parity test states both.
//
// Both kernels are HBM streams: 3 bytes per pixel read twice (the second read hits the 256 MB infinity cache for a batch of
// 16 x 640 x 512), 16 bytes per pixel written.
#include "common.h"

namespace {

constexpr int SUM_THREADS = 256;

// sums (V, 3, 2) uint64 [S, Q], zeroed before the launch.  One workgroup reduces a contiguous run of 4-pixel packets
// (3 aligned 32-bit words each) of one image; a thread's partial sums fit 32 bits (Q <= 65025 * 4 pixels * packets: the launcher keeps
// the run of a workgroup below 2^14 packets per thread).
__global__ __launch_bounds__(SUM_THREADS) void center_sums_kernel(const uint32_t* __restrict__ img, size_t words_per_image,
                                                                   int packets_per_wg, unsigned long long* __restrict__ sums) {
    const int view = blockIdx.y;
    const uint32_t* src = img + (size_t)view * words_per_image;
    const size_t npk = words_per_image / 3;
    const size_t p0 = (size_t)blockIdx.x * packets_per_wg;
    const size_t p1 = p0 + packets_per_wg < npk ? p0 + packets_per_wg : npk;
    uint32_t s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
    for (size_t pk = p0 + threadIdx.x; pk < p1; pk += SUM_THREADS) {
        const uint32_t w0 = src[3 * pk], w1 = src[3 * pk + 1], w2 = src[3 * pk + 2];
        // bytes in memory order: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        const uint32_t b[12] = {w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, w0 >> 24, w1 & 255, (w1 >> 8) & 255, (w1 >> 16) & 255, w1 >> 24,
                                w2 & 255, (w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24};
#pragma unroll
        for (int i = 0; i < 12; ++i) { s[i % 3] += b[i]; q[i % 3] += b[i] * b[i]; }
    }
    __shared__ unsigned long long part[SUM_THREADS / 64][6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned long long a = s[c], b2 = q[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b2 += __shfl_xor(b2, o, 64); }
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][2 * c] = a; part[threadIdx.x >> 6][2 * c + 1] = b2; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < SUM_THREADS / 64; ++w) t += part[w][threadIdx.x];
        atomicAdd(&sums[(size_t)view * 6 + threadIdx.x], t);
    }
}

// One thread = one 4-pixel packet: 3 words in, 4 float4 out.
__global__ __launch_bounds__(256) void center_apply_kernel(const uint32_t* __restrict__ img, size_t words_per_image,
                                                            const unsigned long long* __restrict__ sums, float4* __restrict__ out) {
    const int view = blockIdx.y;
    const size_t npk = words_per_image / 3;
    const size_t pk = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pk >= npk) return;
    const double n = (double)(npk * 4);
    float mean[3], den[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma clang fp contract(off)      // Q / n - m * m with the product rounded on its own: the PyTorch restatement has no fused multiply-add
        const double m = (double)sums[(size_t)view * 6 + 2 * c] / n;
        double var = (double)sums[(size_t)view * 6 + 2 * c + 1] / n - m * m;
        var = var > 0.0 ? var : 0.0;
        mean[c] = (float)m;
        den[c] = (float)sqrt(var) + 0.00000001f;       // the DIVISOR (kept as one: the division below is IEEE, as the reference's)
    }
    const uint32_t* src = img + (size_t)view * words_per_image + 3 * pk;
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    const float b[12] = {(float)(w0 & 255), (float)((w0 >> 8) & 255), (float)((w0 >> 16) & 255), (float)(w0 >> 24),
                         (float)(w1 & 255), (float)((w1 >> 8) & 255), (float)((w1 >> 16) & 255), (float)(w1 >> 24),
                         (float)(w2 & 255), (float)((w2 >> 8) & 255), (float)((w2 >> 16) & 255), (float)(w2 >> 24)};
    float4* dst = out + ((size_t)view * npk + pk) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        dst[i] = make_float4((b[3 * i] - mean[0]) / den[0], (b[3 * i + 1] - mean[1]) / den[1], (b[3 * i + 2] - mean[2]) / den[2], 0.f);
}

}  // namespace

extern "C" size_t mvs_center_images_workspace_bytes(int V) { return V > 0 ? (size_t)V * 6 * sizeof(unsigned long long) : 0; }

extern "C" int mvs_center_images_u8_f32(const uint8_t* images, int V, int H, int W, float* out4, void* workspace, void* stream) {
    MVS_CHECK_ARG(images && out4 && workspace && V > 0 && H > 0 && W > 0);
    const size_t px = (size_t)H * W;
    // 4-pixel packets of 3 aligned words: an image's byte count has to be a multiple of 12 (every size the towers take is: they
    // need H and W divisible by 16) and the batch pointer 4-byte aligned
    if (px % 4 != 0 || (reinterpret_cast<uintptr_t>(images) & 3) || (reinterpret_cast<uintptr_t>(out4) & 15)) return MVS_E_SHAPE;
    if (V > 65535 || px / 4 > (size_t)0x7fffffff * 256) return MVS_E_SHAPE;
    const size_t words = px * 3 / 4, npk = px / 4;
    hipStream_t st = mvs_stream(stream);
    hipError_t e = hipMemsetAsync(workspace, 0, mvs_center_images_workspace_bytes(V), st);
    if (e != hipSuccess) return (int)e;
    // >= 1024 workgroups over the batch where the images are large enough; a thread never sees more than 2^14 packets
    // (Q <= 65025 * 4 * 2^14 < 2^32)
    long long per_wg = (long long)SUM_THREADS * 4;
    const long long want = (long long)((npk * V + 1023) / 1024);
    if (per_wg < want) per_wg = (want + SUM_THREADS - 1) / SUM_THREADS * SUM_THREADS;
    const long long cap = (long long)SUM_THREADS << 14;
    if (per_wg > cap) per_wg = cap;
    const long long gx = (long long)((npk + per_wg - 1) / per_wg);
    if (gx > 0x7fffffff) return MVS_E_SHAPE;
    auto* sums = static_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(center_sums_kernel, dim3((unsigned)gx, V), dim3(SUM_THREADS), 0, st,
                       reinterpret_cast<const uint32_t*>(images), words, (int)per_wg, sums);
    hipLaunchKernelGGL(center_apply_kernel, dim3((unsigned)((npk + 255) / 256), V), dim3(256), 0, st,
                       reinterpret_cast<const uint32_t*>(images), words, sums, reinterpret_cast<float4*>(out4));
    MVS_LAUNCH_RET();
}
