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
    // (wave index through readfirstlane: the role branches must be SCALAR branches -- as a per-lane value the compiler predicates both
    //  sides and every wave walks through both phases with an empty EXEC mask, which costs the same pipe time)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
    const float a = seed * lane, b = seed + lane, m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
    // which waves start with the vector phase: anti = 1: waves 4-7, 2: odd waves, 3: waves 2, 3, 6, 7 (which pairing shares a SIMD?)
    const bool second = anti == 1 ? wave >= 4 : anti == 2 ? (wave & 1) != 0 : anti == 3 ? ((wave >> 1) & 1) != 0 : false;
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
    __shared__ unsigned gcount[2];
    if (threadIdx.x < 2) gcount[threadIdx.x] = 0;
    unsigned gtarget = 0;
    const int grp = wave >> 2;
    // barrier among the four waves of one group only (LDS counter + spin): the two groups drift freely against each other
    auto group_sync = [&]() __attribute__((always_inline)) {
        gtarget += 4;
        if (lane == 0) __hip_atomic_fetch_add(&gcount[grp], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&gcount[grp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < gtarget) __builtin_amdgcn_s_sleep(1);
    };
    __syncthreads();
    const long long t0 = clock64();
    // anti = 4: waves 0-3 ONLY matrix phases, waves 4-7 ONLY vector phases (pure roles, barriers kept); anti = 5: as 1 without the barriers
    for (int it = 0; it < iters; ++it) {
        if (anti == 4) {
            if (wave < 4) matrix(); else vector();
            __builtin_amdgcn_s_barrier();
            if (wave < 4) matrix(); else vector();
            __builtin_amdgcn_s_barrier();
        } else if (anti == 6) {      // as 4, the vector waves start their burst ~1500 clocks into the phase
            if (wave < 4) matrix(); else { for (int z = 0; z < 12; ++z) __builtin_amdgcn_s_sleep(2); vector(); }
            __builtin_amdgcn_s_barrier();
            if (wave < 4) matrix(); else { for (int z = 0; z < 12; ++z) __builtin_amdgcn_s_sleep(2); vector(); }
            __builtin_amdgcn_s_barrier();
        } else if (anti == 7) {      // as 4, but the matrix waves raise their priority
            if (wave < 4) { __builtin_amdgcn_s_setprio(3); matrix(); __builtin_amdgcn_s_setprio(0); } else vector();
            __builtin_amdgcn_s_barrier();
            if (wave < 4) { __builtin_amdgcn_s_setprio(3); matrix(); __builtin_amdgcn_s_setprio(0); } else vector();
            __builtin_amdgcn_s_barrier();
        } else if (anti == 8) {      // two groups (waves 0-3, 4-7), each alternating matrix / vector phases with a GROUP barrier after each phase
            matrix(); group_sync(); vector(); group_sync();
        } else if (anti == 9) {      // the same, group B starts with the vector phase
            if (grp == 0) { matrix(); group_sync(); vector(); group_sync(); } else { vector(); group_sync(); matrix(); group_sync(); }
        } else if (anti == 5) {
            if (wave < 4) { matrix(); vector(); } else { vector(); matrix(); }
        } else {
            if (!second) matrix(); else vector();
            __builtin_amdgcn_s_barrier();
            if (!second) vector(); else matrix();
            __builtin_amdgcn_s_barrier();
        }
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
    double res[10];
    for (int anti = 0; anti < 10; ++anti) {
        phases<NM, NV, NT><<<256, 512>>>(anti, iters, out, 1.0f);
        phases<NM, NV, NT><<<256, 512>>>(anti, iters, out, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
        res[anti] = s / 2048 / iters;
    }
    // per SIMD and iteration: both waves' matrix instructions = 2 * NM * 32 clk of the matrix pipe (s_memtime ticks at 100 MHz: scaled below)
    printf("%-44s lockstep %8.0f | opposite phases: waves 4-7 %8.0f, odd waves %8.0f, waves 2 3 6 7 %8.0f | pure roles (0-3 matrix twice, 4-7 vector twice) %8.0f | waves 4-7 opposite, no barriers %8.0f | pure roles + barriers, vector burst delayed %8.0f | pure roles + barriers, matrix waves s_setprio 3 %8.0f | group barriers only: same start %8.0f, opposite start %8.0f  shader clocks per iteration\n", what, res[0], res[1], res[2], res[3], res[4], res[5], res[6], res[7], res[8], res[9]);
    hipFree(out);
}


// NGRP groups of four waves (wave w, w + 4, w + 8 share a SIMD), each group alternating a vector phase and a matrix phase with a GROUP
// barrier (LDS counter) after each: how busy does the matrix pipe get with two and with three waves per SIMD?
template <int NGRP, int NM, int NV, int NT>
__global__ void __launch_bounds__(256 * NGRP, 1) groups_kernel(int iters, long long* out, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int grp = wave >> 2;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
    const float a = seed * lane, b = seed + lane, m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
    __shared__ unsigned gcount[4];
    if (threadIdx.x < 4) gcount[threadIdx.x] = 0;
    unsigned gtarget = 0;
    auto group_sync = [&]() __attribute__((always_inline)) {
        gtarget += 4;
        if (lane == 0) __hip_atomic_fetch_add(&gcount[grp], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&gcount[grp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < gtarget) __builtin_amdgcn_s_sleep(1);
    };
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < NV / 16; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
        }
        for (int r = 0; r < NT / 16; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 1e-30f + v[i];
        }
        group_sync();
        for (int r = 0; r < NM / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        group_sync();
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[4000] = 1;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}
template <int NGRP, int NM, int NV, int NT>
void run_groups(const char* what) {
    long long* out; hipMalloc(&out, 8192 * 8);
    const int iters = 200;
    static long long h[4096];
    groups_kernel<NGRP, NM, NV, NT><<<256, 256 * NGRP>>>(iters, out, 1.0f);
    groups_kernel<NGRP, NM, NV, NT><<<256, 256 * NGRP>>>(iters, out, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h, out, sizeof(long long) * 4096, hipMemcpyDeviceToHost);
    double s = 0; int cnt = 0;
    for (int b2 = 0; b2 < 256; ++b2) for (int w = 0; w < 4 * NGRP; ++w) { s += (double)h[b2 * 16 + w]; ++cnt; }
    const double per_iter = s / cnt / iters, pipe = (double)NGRP * NM * 32;
    printf("%-58s %d waves per SIMD: %8.0f clocks per iteration of every group; matrix pipe needs %6.0f -> %.0f %% busy\n", what, NGRP, per_iter, pipe, 100.0 * pipe / per_iter);
    hipFree(out);
}

int main() {
    run_groups<2, 432, 1280, 96>("two groups: 432 matrix | 1280 vector + 96 exp2 per wave");
    run_groups<3, 216, 768, 56>("three groups: 216 matrix | 768 vector + 56 exp2 per wave");
    run_groups<3, 216, 1024, 64>("three groups: 216 matrix | 1024 vector + 64 exp2 per wave");
    run_groups<2, 216, 768, 56>("two groups: 216 matrix | 768 vector + 56 exp2 per wave");
    run_groups<1, 216, 768, 56>("one group: 216 matrix | 768 vector + 56 exp2 per wave");
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
