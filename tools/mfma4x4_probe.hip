// Probe of v_mfma_f32_4x4x1_16b_f32's operand layout on gfx950 (used by the MFMA-blend warp kernel, csrc/cost_volume.hip):
// 16 independent blocks, D_b (4x4) += A_b (4x1) * B_b (1x4).  Prints, for every lane and result register, which
// (A lane, B lane) pair produced it.   hipcc --offload-arch=gfx950 -O2 tools/mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
    const int l = threadIdx.x;
    // A = 1 + lane, B = 1000 + 67 * lane: a product a*b identifies both source lanes (checked unique below)
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), (float)(1000 + 67 * l), d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
    float* dv = nullptr;
    if (hipMalloc(&dv, 256 * sizeof(float)) != hipSuccess) return 1;
    probe<<<1, 64>>>(dv);
    float h[256];
    if (hipMemcpy(h, dv, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float v = h[l * 4 + r];
            int fa = -1, fb = -1;
            for (int a = 0; a < 64 && fa < 0; ++a)
                for (int b = 0; b < 64; ++b)
                    if (v == (float)(1 + a) * (float)(1000 + 67 * b)) { fa = a; fb = b; break; }
            // expected: block = l / 4, D[i = r][j = l % 4] = A[lane 4*block + r] * B[lane l]
            const int ea = 4 * (l / 4) + r, eb = l;
            if (fa != ea || fb != eb) { ok = 0; printf("lane %d reg %d: A lane %d B lane %d (expected %d, %d)\n", l, r, fa, fb, ea, eb); }
        }
    printf(ok ? "layout as expected: D[reg r][lane l] = A[lane 4*(l/4)+r] * B[lane l]\n" : "layout differs\n");
    return 0;
}
