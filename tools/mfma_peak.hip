// Calibration micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate on this GPU.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
// Variants: waves per SIMD (1|2), independent accumulators per wave (2|3|4), with / without an LDS
// read between MFMA groups.  Prints TFLOP/s for each; the best is the practical fp32-MFMA ceiling
// (clocks under matrix load are below the 2.4 GHz the 157.3 TFLOP/s spec figure assumes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * (i & 7);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a = {seed, seed * 2, seed * 3, seed * 4}, b = {1.f, 0.5f, 0.25f, 0.125f};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
            f32x4 n = *(const f32x4*)(lds + ((lane * 4 + it * 16) & 4095 & ~3));
            b = n;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r], acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, bool LDS>
void run(int wgs_per_cu, float* out) {
    const int iters = 20000;
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC, LDS><<<grid, 256>>>(out, 100, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC, LDS><<<grid, 256>>>(out, iters, 1e-3f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 4 * NACC * 2048.0;
    printf("waves/SIMD %d  acc %d  lds %d : %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, NACC, (int)LDS, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<2, false>(w, out); run<3, false>(w, out); run<4, false>(w, out);
        run<2, true>(w, out); run<3, true>(w, out); run<4, true>(w, out);
    }
    return 0;
}
