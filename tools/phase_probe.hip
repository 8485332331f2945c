// gfx950: LONG pure phases with a workgroup barrier between them.  One workgroup of 8 waves per CU (two per SIMD: wave w and w + 4).
// Every wave alternates a matrix phase (NM independent v_mfma_f32_16x16x4_f32) and a vector phase (NV independent v_fma_f32 and,
// optionally, transcendentals), `s_barrier` after each phase.  lockstep: all eight waves in the same phase (what gru_fused.hip does);
// anti-phase: waves 4-7 start with the vector phase, so the two waves of a SIMD are always in DIFFERENT phases.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/phase_probe tools/phase_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int NT>
__global__ void __launch_bounds__(512, 1) phases(int anti, int iters, long long* out, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
    const float a = seed * lane, b = seed + lane, m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
    const bool second = anti && wave >= 4;
    auto matrix = [&]() __attribute__((always_inline)) {
        for (int r = 0; r < NM / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    };
    auto vector = [&]() __attribute__((always_inline)) {
        for (int r = 0; r < NV / 16; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
        }
        for (int r = 0; r < NT / 16; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 1e-30f + v[i];
        }
    };
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (!second) matrix(); else vector();
        __builtin_amdgcn_s_barrier();
        if (!second) vector(); else matrix();
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[4000] = 1;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NM, int NV, int NT>
void run(const char* what) {
    long long* out; hipMalloc(&out, 8192 * 8);
    const int iters = 200;
    long long h[2048];
    double res[2];
    for (int anti = 0; anti < 2; ++anti) {
        phases<NM, NV, NT><<<256, 512>>>(anti, iters, out, 1.0f);
        phases<NM, NV, NT><<<256, 512>>>(anti, iters, out, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
        res[anti] = s / 2048 / iters;
    }
    // per SIMD and iteration: both waves' matrix instructions = 2 * NM * 32 clk of the matrix pipe (s_memtime ticks at 100 MHz: scaled below)
    printf("%-44s lockstep %8.0f  anti-phase %8.0f  wall-clock ticks per iteration -> anti-phase / lockstep = %.2f\n", what, res[0], res[1], res[1] / res[0]);
    hipFree(out);
}

int main() {
    // matrix pipe per wave and phase: NM * 32 clk; vector: NV * 4 clk (+ NT * 16)
    run<216, 512, 0>("216 matrix | 512 vector (2048 clk)");
    run<216, 1024, 0>("216 matrix | 1024 vector (4096 clk)");
    run<216, 1728, 0>("216 matrix | 1728 vector (6912 = matrix time)");
    run<432, 1024, 0>("432 matrix | 1024 vector");
    run<432, 768, 64>("432 matrix | 768 vector + 64 exp2");
    run<432, 0, 0>("432 matrix | nothing");
    run<0, 1024, 0>("nothing | 1024 vector");
    return 0;
}
