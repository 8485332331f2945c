// gfx950: do fp32 vector instructions of one wave overlap with fp32 matrix instructions of ANOTHER wave of the same SIMD?
// One workgroup of 8 waves per CU (two per SIMD: wave w and w + 4).  Waves 0-3 run a stream of independent
// v_mfma_f32_16x16x4_f32, waves 4-7 a stream of independent v_fma_f32 (or v_exp_f32 / ds_read_b128), each alone and together;
// the shader-clock time of each role is reported per instruction.
// Build: hipcc --offload-arch=gfx950 -O2 tools/coissue_probe.hip -o tools/bin/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2000;
// mode bits: 1 = matrix waves run, 2 = vector waves run; kind: 0 v_fma_f32, 1 v_exp_f32, 2 ds_read_b128, 3 v_pk_fma_f32
template <int KIND>
__global__ void __launch_bounds__(512, 1) k(int mode, long long* out, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed;
    __syncthreads();
    const bool matrix = wave < 4;
    if ((matrix && !(mode & 1)) || (!matrix && !(mode & 2))) return;
    long long t0 = 0, t1 = 0;
    if (matrix) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
        const float a = seed * lane, b = seed + lane;
        t0 = clock64();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        t1 = clock64();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 12345.678f) out[1000] = 1;
    } else {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
        const float m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
        t0 = clock64();
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) v[i] = __builtin_fmaf(v[i], m, c);
                if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.0f + v[i];
                if (KIND == 2) v[i] += (*(const f32x4*)(lds + ((lane * 4 + i * 256 + it) & 4092)))[0];
                if (KIND == 3) { typedef float f32x2 __attribute__((ext_vector_type(2))); }
            }
        }
        t1 = clock64();
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += v[i];
        if (s == 12345.678f) out[1001] = 1;
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
// Both roles in EVERY wave, as in a real kernel: per iteration a burst of NV vector instructions and a burst of 8 matrix
// instructions.  phase = 0: all eight waves run (vector burst, matrix burst) in lockstep; phase = 1: waves 4-7 run (matrix burst,
// vector burst) -- the two waves of a SIMD in anti-phase.  chain = 1: the 8 matrix instructions of a burst use ONE accumulator
// (dependent chain) instead of eight.
template <int NV, int CHAIN>
__global__ void __launch_bounds__(512, 1) k2(int phase, long long* out, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
    const float a = seed * lane, b = seed + lane, m = 1.0f + seed * 1e-7f, c = seed * 1e-9f;
    const bool flip = phase && wave >= 4;
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (!flip) {
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[CHAIN ? 0 : i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[CHAIN ? 0 : i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (flip) {
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678f) out[1000] = 1;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NV, int CHAIN> void run2(long long* d) {
    const int wgs = 256;
    std::vector<long long> h(wgs * 8);
    double res[2];
    for (int phase = 0; phase < 2; ++phase) {
        k2<NV, CHAIN><<<wgs, 512>>>(phase, d, 1.0f);
        k2<NV, CHAIN><<<wgs, 512>>>(phase, d, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(long long) * wgs * 8, hipMemcpyDeviceToHost);
        double t = 0;
        for (int i = 0; i < wgs * 8; ++i) t += h[i];
        res[phase] = t / (wgs * 8.0) / ITERS;
    }
    printf("8 waves, per iteration %3d v_fma_f32 + 8 v_mfma_f32_16x16x4 (%s): lockstep %6.1f clk per iteration, anti-phase %6.1f clk (matrix pipe alone: %d clk for the SIMD's two waves)\n",
           NV, CHAIN ? "one accumulator " : "8 accumulators  ", res[0], res[1], 2 * 8 * 32);
}
template <int KIND> void run(const char* name, long long* d) {
    const int wgs = 256;
    std::vector<long long> h(wgs * 8);
    double res[4][2] = {};
    for (int mode = 1; mode <= 3; ++mode) {
        hipMemset(d, 0, sizeof(long long) * 4096);
        k<KIND><<<wgs, 512>>>(mode, d, 1.0f);
        k<KIND><<<wgs, 512>>>(mode, d, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(long long) * wgs * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < wgs; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) v += h[b * 8 + w]; }
        res[mode][0] = m / (wgs * 4.0) / (ITERS * 8.0); res[mode][1] = v / (wgs * 4.0) / (ITERS * 16.0);
    }
    const int per = KIND == 1 ? 2 : 1;
    printf("%-14s: matrix alone %5.1f clk per v_mfma_f32_16x16x4 | vector alone %5.2f clk per instruction | together: matrix %5.1f (x%.2f), vector %5.2f (x%.2f)\n",
           name, res[1][0], res[2][1] / per, res[3][0], res[3][0] / res[1][0], res[3][1] / per, res[3][1] / res[2][1]);
}
int main() {
    long long* d; if (hipMalloc(&d, sizeof(long long) * 4096) != hipSuccess) return 1;
    run<0>("v_fma_f32", d); run<1>("v_exp_f32+fma", d); run<2>("ds_read_b128", d);
    run2<32, 0>(d); run2<64, 0>(d); run2<128, 0>(d); run2<64, 1>(d); run2<0, 1>(d);
    return 0;
}
