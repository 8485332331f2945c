// Does s_setprio protect a latency-bound wave from a matrix-bound wave on the same SIMD?  (in the recurrent sweep a gate
// kernel of the chain takes 35 us alone and 79 us beside a launch of the x-part producer, although the chain's waves run at priority 3.)
// One workgroup of 8 waves per CU = two waves per SIMD: waves 0-3 ("chain") run a DEPENDENT chain of vector FMAs (one instruction
// in flight at a time, as the staging / LayerNorm / activation code of a per-plane kernel), waves 4-7 ("producer") issue
// back-to-back matrix instructions on independent accumulators.  Printed: clocks per dependent FMA of the chain waves, alone and
// beside the producer, for chain priorities 0 and 3, and for the producer on v_mfma_f32_16x16x4_f32 (32 cycles in the pipe) or
// v_mfma_f32_4x4x1_16B_f32 (8 cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/prio_probe.hip -o /tmp/prio_probe && /tmp/prio_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PRIO, int MF>      // MF: 0 no producer, 1 = 16x16x4, 2 = 4x4x1
__global__ void __launch_bounds__(512, 1) probe(float* out, long long* clk, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float v = seed * (threadIdx.x & 7);
        const long long t0 = clock64();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v = __builtin_fmaf(v, 1.0001f, seed);       // 16 dependent FMAs
        }
        const long long t1 = clock64();
        if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
        out[blockIdx.x * 512 + threadIdx.x] = v;
    } else {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float a = seed * (threadIdx.x & 3), b = seed + (threadIdx.x & 15);
        // about as long as the chain waves: 16 FMAs of ~8 clocks against 4 (or 16) matrix instructions of 32 (8) clocks per iteration
        if (MF == 1) for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
        }
        if (MF == 2) for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[k & 3], 0, 0, 0);
        }
        const f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
        out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    }
}

template <int PRIO, int MF>
void run(const char* what, float* out, long long* clk, long long* h) {
    const int iters = 20000, nb = 256;
    probe<PRIO, MF><<<nb, 512>>>(out, clk, 100, 1e-3f);
    hipDeviceSynchronize();
    probe<PRIO, MF><<<nb, 512>>>(out, clk, iters, 1e-3f);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, nb * 4 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nb * 4; ++i) s += (double)h[i];
    printf("%-64s %6.2f clocks per dependent FMA\n", what, s / (nb * 4) / ((double)iters * 16));
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&clk, 1024 * sizeof(long long));
    static long long h[1024];
    run<0, 0>("chain waves alone", out, clk, h);
    run<0, 1>("beside v_mfma_f32_16x16x4_f32 waves, same priority", out, clk, h);
    run<1, 1>("beside v_mfma_f32_16x16x4_f32 waves, chain at s_setprio 3", out, clk, h);
    run<0, 2>("beside v_mfma_f32_4x4x1_16B_f32 waves, same priority", out, clk, h);
    run<1, 2>("beside v_mfma_f32_4x4x1_16B_f32 waves, chain at s_setprio 3", out, clk, h);
    return 0;
}
