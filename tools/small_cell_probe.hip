// Probe for VERDICT r3 item 1(a): the 3x3 convolution of ConvGRU cell 2's gates (20 -> 8 channels, 16 x 16 pixel tile staged in
// LDS as the product kernel stages it) on three instruction forms, everything else stripped away:
//   V0  v_pk_fma_f32, weight pairs straight from SGPRs, one pixel per lane        (what csrc/gru.hip conv2d_small_kernel does)
//   V1  v_mfma_f32_4x4x1_16B_f32: 16 blocks x (4 couts x 4 pixels), one pixel per lane, A operand (weights) read from LDS
//   V2  v_mfma_f32_16x16x4_f32: rows = 16 couts (8 real + 8 zero padding), columns = 16 pixels, K = (tap, channel quad)
// Every workgroup convolves its LDS tile REPS times (inputs perturbed per repetition so nothing is hoisted), 4096 workgroups of
// 256 threads: time per (tile, repetition) = the convolution's own cost at full occupancy.  All three give the same sums.
//   hipcc --offload-arch=gfx950 -O3 tools/small_cell_probe.hip -o /tmp/scp && /tmp/scp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int CT = 20, CO = 8, TS = 16, PS = TS + 2, REPS = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float cfloat;

__device__ __forceinline__ void stage(float* tile, const float* x, int tid, float bump) {
    for (int i = tid; i < PS * PS * CT; i += 256) tile[i] = x[i] + bump;
}

__global__ void __launch_bounds__(256) v0_pkfma(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float tile[PS * PS * CT];
    const int tid = threadIdx.x, ly = tid >> 4, lx = tid & 15;
    cfloat* wsh = (cfloat*)w;
    f32x2 tot[CO / 2];
    for (int j = 0; j < CO / 2; ++j) tot[j] = (f32x2){0.f, 0.f};
    for (int r = 0; r < REPS; ++r) {
        stage(tile, x, tid, 1e-3f * r);
        __syncthreads();
        f32x2 acc[CO / 2];
        for (int j = 0; j < CO / 2; ++j) acc[j] = (f32x2){0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* p = tile + ((ly + kh) * PS + lx + kw) * CT;
                float v[CT];
#pragma unroll
                for (int q = 0; q < CT / 4; ++q) { const float4 t = *(const float4*)(p + 4 * q); v[4*q] = t.x; v[4*q+1] = t.y; v[4*q+2] = t.z; v[4*q+3] = t.w; }
                cfloat* wt = wsh + (kh * 3 + kw) * CT * CO;
#pragma unroll
                for (int ci = 0; ci < CT; ++ci)
#pragma unroll
                    for (int j = 0; j < CO / 2; ++j)
                        acc[j] += (f32x2){v[ci], v[ci]} * (f32x2){wt[ci * CO + 2 * j], wt[ci * CO + 2 * j + 1]};
            }
        for (int j = 0; j < CO / 2; ++j) tot[j] += acc[j];
        __syncthreads();
    }
    float* d = y + ((size_t)blockIdx.x * 256 + tid) * CO;
    for (int j = 0; j < CO / 2; ++j) { d[2 * j] = tot[j][0]; d[2 * j + 1] = tot[j][1]; }
}

// V3: the product's vector form as written in csrc/gru.hip (scalar accumulators, left to the vectoriser)
__global__ void __launch_bounds__(256) v3_scalar(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float tile[PS * PS * CT];
    const int tid = threadIdx.x, ly = tid >> 4, lx = tid & 15;
    cfloat* wsh = (cfloat*)w;
    float tot[CO];
    for (int j = 0; j < CO; ++j) tot[j] = 0.f;
    for (int r = 0; r < REPS; ++r) {
        stage(tile, x, tid, 1e-3f * r);
        __syncthreads();
        float acc[CO];
#pragma unroll
        for (int j = 0; j < CO; ++j) acc[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float v[CT];
                const float* p = tile + ((ly + kh) * PS + lx + kw) * CT;
#pragma unroll
                for (int q = 0; q < CT / 4; ++q) { const float4 t = *(const float4*)(p + 4 * q); v[4*q] = t.x; v[4*q+1] = t.y; v[4*q+2] = t.z; v[4*q+3] = t.w; }
                cfloat* wt = wsh + (kh * 3 + kw) * CT * CO;
#pragma unroll
                for (int ci = 0; ci < CT; ++ci)
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j] += v[ci] * wt[ci * CO + j];
            }
        }
        for (int j = 0; j < CO; ++j) tot[j] += acc[j];
        __syncthreads();
    }
    float* d = y + ((size_t)blockIdx.x * 256 + tid) * CO;
    for (int j = 0; j < CO; ++j) d[j] = tot[j];
}

// V1: lane = pixel (wave w covers tile rows 4w .. 4w+3: lane l -> row 4w + l/16, column l%16).  Per (tap, ci): B = the lane's own
// input value, A = W[co = lane & 3 (+4)][k] from LDS, D[reg r] = cout r of the lane's pixel (tools/mfma4x4_probe.hip).
__global__ void __launch_bounds__(256) v1_mfma4x4(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float tile[PS * PS * CT];
    __shared__ __attribute__((aligned(16))) float wl[9 * CT * CO];          // [k][co]
    const int tid = threadIdx.x, ly = tid >> 4, lx = tid & 15, lane = tid & 63;
    for (int i = tid; i < 9 * CT * CO; i += 256) wl[i] = w[i];
    f32x4 tot0 = {0.f, 0.f, 0.f, 0.f}, tot1 = tot0;
    const int arow = lane & 3;
    for (int r = 0; r < REPS; ++r) {
        stage(tile, x, tid, 1e-3f * r);
        __syncthreads();
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float* p = tile + ((ly + kh) * PS + lx + kw) * CT;
                float v[CT];
#pragma unroll
                for (int q = 0; q < CT / 4; ++q) { const float4 t = *(const float4*)(p + 4 * q); v[4*q] = t.x; v[4*q+1] = t.y; v[4*q+2] = t.z; v[4*q+3] = t.w; }
                const float* wt = wl + (kh * 3 + kw) * CT * CO + arow;
#pragma unroll
                for (int ci = 0; ci < CT; ++ci) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wt[ci * CO], v[ci], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wt[ci * CO + 4], v[ci], a1, 0, 0, 0);
                }
            }
        tot0 += a0; tot1 += a1;
        __syncthreads();
    }
    float* d = y + ((size_t)blockIdx.x * 256 + tid) * CO;
    for (int j = 0; j < 4; ++j) { d[j] = tot0[j]; d[4 + j] = tot1[j]; }
}

// V2: wave w covers tile rows 4w .. 4w+3, one 16-pixel row per column tile; lane (n = lane & 15, kq = lane >> 4).  Per (tap,
// channel quad g): B = x[pixel n][4g + kq], A = Wpad[co = n][k = 4g + kq] (rows 8..15 zero), D[reg j] = cout 4*kq + j of pixel n.
__global__ void __launch_bounds__(256) v2_mfma16(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float tile[PS * PS * CT];
    __shared__ __attribute__((aligned(16))) float wl[9 * CT * 16];          // [k][16 couts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 9 * CT * 16; i += 256) { const int co = i & 15, k = i >> 4; wl[i] = co < CO ? w[k * CO + co] : 0.f; }
    f32x4 tot[4];
    for (int t = 0; t < 4; ++t) tot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < REPS; ++r) {
        stage(tile, x, tid, 1e-3f * r);
        __syncthreads();
        f32x4 acc[4];
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int g = 0; g < CT / 4; ++g) {
                    const float aval = wl[((kh * 3 + kw) * CT + 4 * g + kq) * 16 + n];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float bval = tile[((4 * wave + t + kh) * PS + n + kw) * CT + 4 * g + kq];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, bval, acc[t], 0, 0, 0);
                    }
                }
        for (int t = 0; t < 4; ++t) tot[t] += acc[t];
        __syncthreads();
    }
    // D[reg j][lane] = cout 4*kq + j of pixel column n; couts 0..7 live in kq 0, 1
    if (kq < 2)
        for (int t = 0; t < 4; ++t) {
            float* d = y + ((size_t)blockIdx.x * 256 + (4 * wave + t) * 16 + n) * CO + 4 * kq;
            for (int j = 0; j < 4; ++j) d[j] = tot[t][j];
        }
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int WGS = 4096;
    std::vector<float> hx(PS * PS * CT), hw(9 * CT * CO);
    srand(1);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
    float *x, *w, *y[4];
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4));
    for (int i = 0; i < 4; ++i) CK(hipMalloc(&y[i], (size_t)WGS * 256 * CO * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"V0 v_pk_fma_f32, SGPR weights          ", "V1 v_mfma_f32_4x4x1_16B, A from LDS      ", "V2 v_mfma_f32_16x16x4, couts padded to 16",
                            "V3 scalar accumulators (product source)  "};
    std::vector<float> ref((size_t)256 * CO), got((size_t)256 * CO);
    for (int v = 0; v < 4; ++v) {
        float best = 1e30f;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            if (v == 0) v0_pkfma<<<WGS, 256>>>(x, w, y[0]);
            else if (v == 1) v1_mfma4x4<<<WGS, 256>>>(x, w, y[1]);
            else if (v == 2) v2_mfma16<<<WGS, 256>>>(x, w, y[2]);
            else v3_scalar<<<WGS, 256>>>(x, w, y[3]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best) best = ms;
        }
        CK(hipMemcpy(v == 0 ? ref.data() : got.data(), y[v], ref.size() * 4, hipMemcpyDeviceToHost));
        double err = 0.0, mag = 0.0;
        if (v > 0) for (size_t i = 0; i < ref.size(); ++i) { err = fmax(err, fabs((double)got[i] - ref[i])); mag = fmax(mag, fabs((double)ref[i])); }
        const double flop = 2.0 * 9 * CT * CO * 256.0 * WGS * REPS;
        printf("%s  %8.3f ms  %6.1f TFLOP/s of real work  (%.2f ns per tile and repetition per CU-slot)  max |diff| vs V0 %.2e of %.1f\n",
               names[v], best, flop / best / 1e9, best * 1e6 / ((double)WGS * REPS / 256.0), err, mag);
    }
    return 0;
}
