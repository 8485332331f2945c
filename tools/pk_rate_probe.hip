// Issue rate of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against v_fma_f32 on gfx950: register-only loops of independent instructions,
// 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 -o tools/bin/pk_rate_probe tools/pk_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float seed) {
    f32x2 a[16]; float s[16];
    const f32x2 m = {1.0000001f, 0.9999999f}, c = {seed, -seed};
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = (f32x2){seed + i, seed - i}; s[i] = seed * i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m[0]), "v"(c[0]));
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i][0] + a[i][1] + s[i];
    if (r == 123.456f) out[0] = r;
}

int main() {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[4] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_fma_f32"};
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int k = 0; k < 4; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                const dim3 grid(256 * wps);               // 256-thread workgroups = one wave per SIMD each
                if (k == 0) rate_kernel<0><<<grid, 256>>>(out, iters, 1.5f);
                if (k == 1) rate_kernel<1><<<grid, 256>>>(out, iters, 1.5f);
                if (k == 2) rate_kernel<2><<<grid, 256>>>(out, iters, 1.5f);
                if (k == 3) rate_kernel<3><<<grid, 256>>>(out, iters, 1.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)iters * 16 * wps;
            printf("%d wave(s) per SIMD  %-13s %8.1f us  -> %.2f ns = %.1f clk (2.4 GHz) per wave-instruction per SIMD\n", wps, names[k], best * 1e3,
                   best * 1e6 / instr_per_simd, best * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}
