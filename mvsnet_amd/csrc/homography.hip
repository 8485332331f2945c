// Plane-induced homographies (R1, R1') and their pixel-coordinate 8-vectors (R2 prep).
// Reference behaviour: mvsnet/homography_warping.py:10-106 and :216-250.
// One thread per (source view, depth plane); the arithmetic is a few dozen flops, so the
// only design goal is: same operation order as the reference, one launch instead of ~15
// TensorFlow ops per view + 11 slices per (view, plane).
#include "common.h"

namespace {

struct M3 { float m[3][3]; };

__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}

__device__ __forceinline__ M3 inverse(const M3& a) {
    const float (*m)[3] = a.m;
    float c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1];
    float c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2];
    float c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
    float det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02;
    float id = 1.0f / det;
    M3 r;
    r.m[0][0] = c00 * id;
    r.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
    r.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
    r.m[1][0] = c01 * id;
    r.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
    r.m[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
    r.m[2][0] = c02 * id;
    r.m[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
    r.m[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
    return r;
}

// cams: (N,2,4,4).  cam[0] rows 0..2 = [R | t]; cam[1] rows 0..2 cols 0..2 = K.
__global__ void homography_kernel(const float* __restrict__ cams, int n_src, int D,
                                  float depth_start, float depth_interval, float depth_end,
                                  int inverse_depth, float* __restrict__ Hout,
                                  float* __restrict__ Tout, double* __restrict__ zero, int zero_n) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // the fused features -> depth entry also clears the regulariser's BatchNorm sums here (one launch less per depth map)
    for (int i = idx; i < zero_n; i += gridDim.x * blockDim.x) zero[i] = 0.0;
    if (idx >= n_src * D) return;
    int v = idx / D, d = idx - v * D;
    const float* L = cams;                    // reference ("left") camera
    const float* Rc = cams + (size_t)(v + 1) * 32;  // source ("right") camera

    M3 Rl, Rr, Kl, Kr;
    float tl[3], tr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Rl.m[i][j] = L[i * 4 + j];
            Rr.m[i][j] = Rc[i * 4 + j];
            Kl.m[i][j] = L[16 + i * 4 + j];
            Kr.m[i][j] = Rc[16 + i * 4 + j];
        }
        tl[i] = L[i * 4 + 3];
        tr[i] = Rc[i * 4 + 3];
    }
    // depth of this plane (homography_warping.py:28-30 / :74-77)
    float depth;
    if (inverse_depth) {
        float a = 1.0f / depth_start, b = 1.0f / depth_end;
        float step = (b - a) / (float)(D > 1 ? D - 1 : 1);
        depth = 1.0f / (a + (float)d * step);
    } else {
        depth = (float)d * depth_interval + depth_start;
    }
    M3 Kli = inverse(Kl);                                        // :33
    M3 Rlt, Rrt;                                                 // :34-35
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { Rlt.m[i][j] = Rl.m[j][i]; Rrt.m[i][j] = Rr.m[j][i]; }
    float crel[3];                                               // :39-41
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float cl = -(Rlt.m[i][0] * tl[0] + Rlt.m[i][1] * tl[1] + Rlt.m[i][2] * tl[2]);
        float cr = -(Rrt.m[i][0] * tr[0] + Rrt.m[i][1] * tr[1] + Rrt.m[i][2] * tr[2]);
        crel[i] = cr - cl;
    }
    M3 mid0;                                                     // :45-50  I - c_rel n^T / depth
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            mid0.m[i][j] = (i == j ? 1.0f : 0.0f) - (crel[i] * Rl.m[2][j]) / depth;
    M3 mid1 = mul(Rlt, Kli);                                     // :51
    M3 mid2 = mul(mid0, mid1);                                   // :52
    M3 Hm = mul(Kr, mul(Rr, mid2));                              // :54-56

    if (Hout) {
        float* h = Hout + (size_t)idx * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) h[i * 3 + j] = Hm.m[i][j];
    }
    if (Tout) {
        // pixel-centre conversion, homography_warping.py:226-234, then / c2' (:248-250)
        float a0 = Hm.m[0][0], a1 = Hm.m[0][1], a2 = Hm.m[0][2];
        float b0 = Hm.m[1][0], b1 = Hm.m[1][1], b2 = Hm.m[1][2];
        float c0 = Hm.m[2][0], c1 = Hm.m[2][1], c2 = Hm.m[2][2];
        float a_0 = a0 - c0 / 2.f;
        float a_1 = a1 - c1 / 2.f;
        float a_2 = (a0 + a1) / 2.f + a2 - (c0 + c1) / 4.f - c2 / 2.f;
        float b_0 = b0 - c0 / 2.f;
        float b_1 = b1 - c1 / 2.f;
        float b_2 = (b0 + b1) / 2.f + b2 - (c0 + c1) / 4.f - c2 / 2.f;
        float c_2 = c2 + (c0 + c1) / 2.f;
        float* t = Tout + (size_t)idx * 8;
        t[0] = a_0 / c_2; t[1] = a_1 / c_2; t[2] = a_2 / c_2;
        t[3] = b_0 / c_2; t[4] = b_1 / c_2; t[5] = b_2 / c_2;
        t[6] = c0 / c_2;  t[7] = c1 / c_2;
    }
}

}  // namespace

extern "C" int mvs_homography_transforms_f32(const float* cams, int view_num, int depth_num,
                                             float depth_start, float depth_interval,
                                             float depth_end, int inverse_depth,
                                             float* homographies, float* transforms,
                                             void* stream) {
    MVS_CHECK_ARG(cams && view_num >= 2 && depth_num >= 1);
    MVS_CHECK_ARG(homographies || transforms);
    int n = (view_num - 1) * depth_num;
    homography_kernel<<<mvs_cdiv(n, 128), 128, 0, mvs_stream(stream)>>>(
        cams, view_num - 1, depth_num, depth_start, depth_interval, depth_end, inverse_depth,
        homographies, transforms, nullptr, 0);
    MVS_LAUNCH_RET();
}

// same + zero-fill of `zero_n` doubles (regnet.hip: the BatchNorm sums of the depth map about to be computed)
int mvs_homography_transforms_zero(const float* cams, int view_num, int depth_num, float depth_start,
                                   float depth_interval, float depth_end, int inverse_depth, float* transforms,
                                   double* zero, int zero_n, hipStream_t st) {
    int n = (view_num - 1) * depth_num;
    homography_kernel<<<mvs_cdiv(n, 128), 128, 0, st>>>(cams, view_num - 1, depth_num, depth_start, depth_interval,
                                                       depth_end, inverse_depth, nullptr, transforms, zero, zero_n);
    return (int)hipGetLastError();
}
