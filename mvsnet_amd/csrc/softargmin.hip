// softmax(-reg) over depth + soft-argmin depth + 4-bucket probability map in one kernel (R6 + R7).
// Reference behaviour: mvsnet/model.py:471-498 and get_probability_map_slice (:45-144); the
// reference makes 4 passes over the (D,H,W) volume plus 4 gather_nd; here the volume is read
// twice (max, then exp-sum / weighted sum), both coalesced along W, and the four bucket
// probabilities are recomputed from 4 point reads.
//
// Roofline: HBM, D*H*W*4 bytes in (+8 bytes/pixel out).  Block = 64 consecutive pixels x 4 depth
// groups (wave g sweeps planes g, g+4, ...), combined through LDS.
#include "common.h"

namespace {

__device__ __forceinline__ float depth_at(int d, int D, float start, float interval, int inverse) {
    float end = start + ((float)D - 1.0f) * interval;                    // model.py:378-379
    float denom = (float)(D > 1 ? D - 1 : 1);
    if (inverse) {                                                        // :481-485
        float a = 1.0f / start, b = 1.0f / end;
        return 1.0f / (a + (float)d * ((b - a) / denom));
    }
    return start + (float)d * ((end - start) / denom);                    // :487-488
}

// Tile variant: the block's 32 pixel columns (32 x D floats) are pulled into LDS with every load
// in flight at once (the plain kernel below walks depth with a few dependent loads per thread and
// is latency-bound at ~1 workgroup per CU); max / exp-sum / weighted sum / bucket reads then run
// out of LDS, so the volume is read from memory exactly once.
constexpr int SA_PX = 32, SA_G = 16;
__global__ void __launch_bounds__(SA_PX * SA_G)
softargmin_prob_tile_kernel(const float* __restrict__ reg, int D, int HW, float start, float interval,
                            int inverse, float* __restrict__ depth_out, float* __restrict__ prob_out) {
    extern __shared__ float tile[];                  // [D][SA_PX] z = -reg, later exp(z - max)
    __shared__ float sh_a[SA_G][SA_PX];
    __shared__ float sh_b[SA_G][SA_PX];
    const int px = threadIdx.x % SA_PX, g = threadIdx.x / SA_PX;
    const int pix = blockIdx.x * SA_PX + px;
    const bool valid = pix < HW;
    const float* col = reg + (valid ? pix : 0);

    int d = g;
    for (; d + 3 * SA_G < D; d += 4 * SA_G) {        // four independent loads per trip
        float v0 = col[(size_t)d * HW], v1 = col[(size_t)(d + SA_G) * HW];
        float v2 = col[(size_t)(d + 2 * SA_G) * HW], v3 = col[(size_t)(d + 3 * SA_G) * HW];
        tile[d * SA_PX + px] = -v0; tile[(d + SA_G) * SA_PX + px] = -v1;
        tile[(d + 2 * SA_G) * SA_PX + px] = -v2; tile[(d + 3 * SA_G) * SA_PX + px] = -v3;
    }
    for (; d < D; d += SA_G) tile[d * SA_PX + px] = -col[(size_t)d * HW];

    // pass 1: max of z (own planes: no barrier needed before reading them back)
    float m = -INFINITY;
    for (d = g; d < D; d += SA_G) m = fmaxf(m, tile[d * SA_PX + px]);
    sh_a[g][px] = m;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SA_G; ++i) m = fmaxf(m, sh_a[i][px]);
    __syncthreads();

    // pass 2: exp, sum of exp and depth-weighted sum (same per-thread order as the plain kernel's
    // strided walk, then a fixed-order combine)
    float se = 0.f, sz = 0.f;
    for (d = g; d < D; d += SA_G) {
        float e = expf(tile[d * SA_PX + px] - m);
        tile[d * SA_PX + px] = e;
        se += e;
        sz += e * depth_at(d, D, start, interval, inverse);
    }
    sh_a[g][px] = se; sh_b[g][px] = sz;
    __syncthreads();
    if (g != 0 || !valid) return;
    se = 0.f; sz = 0.f;
#pragma unroll
    for (int i = 0; i < SA_G; ++i) { se += sh_a[i][px]; sz += sh_b[i][px]; }
    const float dep = sz / se;                                            // model.py:493-494
    depth_out[pix] = dep;

    int l0, r0;                                                           // model.py:83-140
    if (inverse) {
        float end = start + ((float)D - 1.0f) * interval;
        float inv_s = 1.0f / start, inv_e = 1.0f / end;
        float inv_int = (inv_s - inv_e) / ((float)D - 1.0f);
        float idx = (1.0f / dep - inv_e) / inv_int;
        l0 = D - (int)ceilf(idx) - 1;
        r0 = D - (int)floorf(idx) - 1;
    } else {
        float idx = (dep - start) / interval;
        l0 = (int)floorf(idx);
        r0 = (int)ceilf(idx);
    }
    l0 = min(max(l0, 0), D - 1);
    r0 = min(max(r0, 0), D - 1);
    int l1 = min(max(l0 - 1, 0), D - 1);
    int r1 = min(max(r0 + 1, 0), D - 1);
    float pl0 = tile[l0 * SA_PX + px] / se, pr0 = tile[r0 * SA_PX + px] / se;
    float pl1 = tile[l1 * SA_PX + px] / se, pr1 = tile[r1 * SA_PX + px] / se;
    prob_out[pix] = (pl0 + pr0) + (pl1 + pr1);
}

__global__ void __launch_bounds__(256)
softargmin_prob_kernel(const float* __restrict__ reg, int D, int HW, float start, float interval,
                       int inverse, float* __restrict__ depth_out, float* __restrict__ prob_out) {
    __shared__ float sh_a[4][64];
    __shared__ float sh_b[4][64];
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int pix = blockIdx.x * 64 + px;
    const bool valid = pix < HW;
    const float* col = reg + (valid ? pix : 0);

    // pass 1: max of z = -reg
    float m = -INFINITY;
    for (int d = g; d < D; d += 4) m = fmaxf(m, -col[(size_t)d * HW]);
    sh_a[g][px] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh_a[0][px], sh_a[1][px]), fmaxf(sh_a[2][px], sh_a[3][px]));
    __syncthreads();

    // pass 2: sum of exp and depth-weighted sum
    float se = 0.f, sz = 0.f;
    for (int d = g; d < D; d += 4) {
        float e = expf(-col[(size_t)d * HW] - m);
        se += e;
        sz += e * depth_at(d, D, start, interval, inverse);
    }
    sh_a[g][px] = se; sh_b[g][px] = sz;
    __syncthreads();
    if (g != 0 || !valid) return;
    se = (sh_a[0][px] + sh_a[1][px]) + (sh_a[2][px] + sh_a[3][px]);
    sz = (sh_b[0][px] + sh_b[1][px]) + (sh_b[2][px] + sh_b[3][px]);
    const float dep = sz / se;                                            // model.py:493-494
    depth_out[pix] = dep;

    // probability map: P[l0] + P[r0] + P[l1] + P[r1]   (model.py:83-140)
    int l0, r0;
    if (inverse) {
        float end = start + ((float)D - 1.0f) * interval;
        float inv_s = 1.0f / start, inv_e = 1.0f / end;
        float inv_int = (inv_s - inv_e) / ((float)D - 1.0f);
        float idx = (1.0f / dep - inv_e) / inv_int;
        l0 = D - (int)ceilf(idx) - 1;
        r0 = D - (int)floorf(idx) - 1;
    } else {
        float idx = (dep - start) / interval;
        l0 = (int)floorf(idx);
        r0 = (int)ceilf(idx);
    }
    l0 = min(max(l0, 0), D - 1);
    r0 = min(max(r0, 0), D - 1);
    int l1 = min(max(l0 - 1, 0), D - 1);
    int r1 = min(max(r0 + 1, 0), D - 1);
    float pl0 = expf(-col[(size_t)l0 * HW] - m) / se;
    float pr0 = expf(-col[(size_t)r0 * HW] - m) / se;
    float pl1 = expf(-col[(size_t)l1 * HW] - m) / se;
    float pr1 = expf(-col[(size_t)r1 * HW] - m) / se;
    prob_out[pix] = (pl0 + pr0) + (pl1 + pr1);
}

}  // namespace

extern "C" int mvs_softargmin_prob_f32(const float* reg, int D, int H, int W, float depth_start,
                                       float depth_interval, int inverse_depth, float* depth,
                                       float* prob, void* stream) {
    MVS_CHECK_ARG(reg && depth && prob && D > 0 && H > 0 && W > 0);
    int HW = H * W;
    const size_t tile_bytes = (size_t)D * SA_PX * sizeof(float);
    if (tile_bytes <= 60 * 1024) {                   // D <= 480: the whole column tile fits in LDS
        softargmin_prob_tile_kernel<<<mvs_cdiv(HW, SA_PX), SA_PX * SA_G, tile_bytes, mvs_stream(stream)>>>(
            reg, D, HW, depth_start, depth_interval, inverse_depth, depth, prob);
        MVS_LAUNCH_RET();
    }
    softargmin_prob_kernel<<<mvs_cdiv(HW, 64), 256, 0, mvs_stream(stream)>>>(
        reg, D, HW, depth_start, depth_interval, inverse_depth, depth, prob);
    MVS_LAUNCH_RET();
}
