// Timing probe behind DESIGN 4.1's "generate the cost-volume planes inside the fused 3dconv0_1 + 3dconv1_0 pass?" (VERDICT r2 item 5).
// The pair kernel issues ~294 v_mfma_f32_16x16x4_f32 per wave and plane (2 waves per SIMD) with ~100 vector instructions beside
// them; building a staged plane from the feature maps inside its staging would add, per wave and plane, ~800-900 vector
// instructions (the warp + variance kernel's 143 per 64 (pixel, channel-quad) items x 22.5 items of the 10 x 18 x 8 slab)
// and ~50 16-byte gathers.  This probe runs the pair kernel's instruction mix -- an MFMA stream with LDS operand reads one
// step ahead -- with NV INDEPENDENT vector FMAs (8 chains) and NL global 16-byte loads mixed into every group of 12 MFMAs,
// and prints the slowdown: 37 vector instructions + 2 loads per 12 MFMAs is the fused kernel's ratio.
//   hipcc --offload-arch=gfx950 -O3 tools/fuse_probe.hip -o /tmp/fuse_probe && /tmp/fuse_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL>
__global__ void __launch_bounds__(256, 2) mix_loop(float* out, const float4* __restrict__ src, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i & 7);
    __syncthreads();
    f32x4 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    f32x4 op[2][3];
    const int addr0 = lane * 4;
    float vv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) vv[k] = seed * (k + 1);
    float4 ld[NL > 0 ? NL : 1];
#pragma unroll
    for (int k = 0; k < (NL > 0 ? NL : 1); ++k) ld[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned goff = (blockIdx.x * 256 + threadIdx.x) * 37u;
#pragma unroll
    for (int r = 0; r < 3; ++r) op[0][r] = *(const f32x4*)(lds + ((addr0 + 256 * r) & 8191));
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 3; ++r) op[h ^ 1][r] = *(const f32x4*)(lds + addr0 + 256 * r + 1024 * h);
#pragma unroll
            for (int k = 0; k < NL; ++k) {                 // gathers from a 13 MB (L2 / Infinity Cache resident) buffer, consumed one step later
                vv[k & 7] += ld[k].x + ld[k].w;
                ld[k] = src[(goff + 977u * k) & 0xFFFFF];
                goff += 4099u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < NV; ++v) vv[v & 7] = vv[v & 7] * 1.0001f + seed;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[h][i][j], op[h][(i + 1) % 3][j], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += vv[k];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + t;
}

static float base_ms = 0.f;
template <int NV, int NL>
void run(float* out, const float4* src, int lds_bytes) {
    const int iters = 20000, grid = 512;
    hipFuncSetAttribute((const void*)mix_loop<NV, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix_loop<NV, NL><<<grid, 256, lds_bytes>>>(out, src, 100, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_loop<NV, NL><<<grid, 256, lds_bytes>>>(out, src, iters, 1e-3f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (NV == 0 && NL == 0) base_ms = ms;
    double flops = (double)grid * 4 * iters * 12 * 2048.0;
    printf("per 12 MFMAs: %2d vector FMAs, %d gathers : %7.3f ms  %6.1f TFLOP/s  x%.3f of the pure stream\n", NV, NL, ms, flops / ms / 1e9, ms / base_ms);
}

int main() {
    float* out; hipMalloc(&out, 512 * 256 * sizeof(float));
    float4* src; hipMalloc(&src, (size_t)(1 << 20) * sizeof(float4)); hipMemset(src, 0, (size_t)(1 << 20) * sizeof(float4));
    const int L = 70 * 1024;   // 2 workgroups per CU, as the pair kernel
    run<0, 0>(out, src, L); run<4, 0>(out, src, L); run<12, 0>(out, src, L); run<24, 0>(out, src, L); run<36, 0>(out, src, L); run<48, 0>(out, src, L);
    run<0, 2>(out, src, L); run<36, 2>(out, src, L); run<36, 4>(out, src, L);
    return 0;
}
