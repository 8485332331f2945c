// Calibration: MFMA stream with software-pipelined LDS operand reads and some VALU work mixed in,
// mimicking the conv inner loop (3 ds_read_b128 + NV VALU ops per 12 MFMAs, operands one step ahead).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NRD>
__global__ void __launch_bounds__(256, 2) mix_loop(float* out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i & 7);
    __syncthreads();
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    f32x4 op[2][3];
    int addr = lane * 4;
    const int addr0 = lane * 4;
    float vv = seed;
#pragma unroll
    for (int r = 0; r < 3; ++r) op[0][r] = *(const f32x4*)(lds + ((addr + 256 * r) & 8191));
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < NRD; ++r) op[h ^ 1][r % 3] = *(const f32x4*)(lds + addr0 + 256 * r + 1024 * h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) vv = vv * 1.0001f + seed;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[h][i][j], op[h][(i + 1) % 3][j], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        addr += 64;
    }
    f32x4 s = acc[0] + acc[1] + acc[2];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + vv;
}

template <int NV, int NRD>
void run(float* out, int lds_bytes) {
    const int iters = 20000, grid = 512;
    hipFuncSetAttribute((const void*)mix_loop<NV, NRD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix_loop<NV, NRD><<<grid, 256, lds_bytes>>>(out, 100, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_loop<NV, NRD><<<grid, 256, lds_bytes>>>(out, iters, 1e-3f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 12 * 2048.0;
    printf("VALU/step %2d  reads/step %d : %.3f ms  %.1f TFLOP/s\n", NV, NRD, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 512 * 256 * sizeof(float));
    const int L = 70 * 1024;   // 2 workgroups per CU
    run<0, 0>(out, L); run<0, 1>(out, L); run<0, 3>(out, L); run<0, 6>(out, L); run<0, 9>(out, L); run<8, 0>(out, L); run<16, 0>(out, L); run<8, 3>(out, L);
    return 0;
}
