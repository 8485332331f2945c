// Many small tensors in ONE launch (the training step's host side): the towers' backward ends with ~94 parameter
// gradients -- 32 kernels in ATen's (Cout, Cin, k, k) layout that the flat gradient buffer wants in TensorFlow's (k, k, Cin, Cout),
// 62 GroupNorm gamma / beta rows accumulated in float64 -- which used to travel through one permute-copy and one autograd
// accumulation launch EACH (~190 launches of a step the launching thread bounds).  The jobs ride in the kernel arguments
// (a by-value table: no device-side table to upload or keep alive); blockIdx.y = job.
#include "common.h"

namespace {

constexpr int TR_MAX = 48;
struct TransposeJob { const float* src; float* dst; int B, A, KK, keep; };
struct TransposeJobs { TransposeJob j[TR_MAX]; };

// dst (KK, keep, B) += src (B, A, KK) with the axes reversed; a >= keep is dropped (the image's padding channel)
__global__ __launch_bounds__(256) void transpose_add_kernel(TransposeJobs jobs) {
    const TransposeJob J = jobs.j[blockIdx.y];
    const long long total = (long long)J.KK * J.keep * J.B;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i % J.B);
        const long long r = i / J.B;
        const int a = (int)(r % J.keep), kk = (int)(r / J.keep);
        J.dst[i] += J.src[((size_t)b * J.A + a) * J.KK + kk];
    }
}

constexpr int ADD_MAX = 96;
struct AddJob { const double* src; float* dst; int n; };
struct AddJobs { AddJob j[ADD_MAX]; };

__global__ __launch_bounds__(128) void add_f64_kernel(AddJobs jobs) {
    const AddJob J = jobs.j[blockIdx.x];
    for (int i = threadIdx.x; i < J.n; i += 128) J.dst[i] += (float)J.src[i];
}

constexpr int ADDF_MAX = 96;
struct AddFJob { const float* src; float* dst; long long n; };
struct AddFJobs { AddFJob j[ADDF_MAX]; };

__global__ __launch_bounds__(256) void add_f32_kernel(AddFJobs jobs) {
    const AddFJob J = jobs.j[blockIdx.y];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < J.n; i += (long long)gridDim.x * 256) J.dst[i] += J.src[i];
}

}  // namespace

extern "C" int mvs_transpose_add_many_f32(int n, const float* const* src, float* const* dst, const int* dims, void* stream) {
    MVS_CHECK_ARG(n > 0 && src && dst && dims);
    for (int i = 0; i < n; ++i)
        MVS_CHECK_ARG(src[i] && dst[i] && dims[4 * i] > 0 && dims[4 * i + 1] > 0 && dims[4 * i + 2] > 0 && dims[4 * i + 3] > 0 &&
                      dims[4 * i + 3] <= dims[4 * i + 1]);
    for (int first = 0; first < n; first += TR_MAX) {
        const int m = n - first < TR_MAX ? n - first : TR_MAX;
        TransposeJobs jobs;
        long long largest = 1;
        for (int i = 0; i < m; ++i) {
            const int* d = dims + 4 * (first + i);
            jobs.j[i] = {src[first + i], dst[first + i], d[0], d[1], d[2], d[3]};
            const long long t = (long long)d[0] * d[2] * d[3];
            if (t > largest) largest = t;
        }
        const int gx = (int)((largest + 1023) / 1024 < 64 ? (largest + 1023) / 1024 : 64);
        hipLaunchKernelGGL(transpose_add_kernel, dim3(gx, m), dim3(256), 0, mvs_stream(stream), jobs);
    }
    MVS_LAUNCH_RET();
}

extern "C" int mvs_add_f64_many_f32(int n, const double* const* src, float* const* dst, const int* counts, void* stream) {
    MVS_CHECK_ARG(n > 0 && src && dst && counts);
    for (int i = 0; i < n; ++i) MVS_CHECK_ARG(src[i] && dst[i] && counts[i] > 0);
    for (int first = 0; first < n; first += ADD_MAX) {
        const int m = n - first < ADD_MAX ? n - first : ADD_MAX;
        AddJobs jobs;
        for (int i = 0; i < m; ++i) jobs.j[i] = {src[first + i], dst[first + i], counts[first + i]};
        hipLaunchKernelGGL(add_f64_kernel, dim3(m), dim3(128), 0, mvs_stream(stream), jobs);
    }
    MVS_LAUNCH_RET();
}

extern "C" int mvs_add_many_f32(int n, const float* const* src, float* const* dst, const long long* counts, void* stream) {
    MVS_CHECK_ARG(n > 0 && src && dst && counts);
    for (int i = 0; i < n; ++i) MVS_CHECK_ARG(src[i] && dst[i] && counts[i] > 0);
    for (int first = 0; first < n; first += ADDF_MAX) {
        const int m = n - first < ADDF_MAX ? n - first : ADDF_MAX;
        AddFJobs jobs;
        long long largest = 1;
        for (int i = 0; i < m; ++i) {
            jobs.j[i] = {src[first + i], dst[first + i], counts[first + i]};
            if (counts[first + i] > largest) largest = counts[first + i];
        }
        const int gx = (int)((largest + 2047) / 2048 < 32 ? (largest + 2047) / 2048 : 32);
        hipLaunchKernelGGL(add_f32_kernel, dim3(gx, m), dim3(256), 0, mvs_stream(stream), jobs);
    }
    MVS_LAUNCH_RET();
}
