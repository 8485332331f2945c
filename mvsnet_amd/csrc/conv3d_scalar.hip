// Shape-generic scalar (VALU) 3x3x3 convolution / transposed convolution, BatchNorm finalise.
// This is the slow HIP cross-check for the MFMA kernels in conv3d_mfma.hip and the path for
// channel counts the MFMA tiling does not cover (e.g. network_mode 'semilite': 6/12/24/48).
// Reference behaviour: tf.layers.conv3d / conv3d_transpose with padding='SAME', no bias
// (mvsnet/cnn_wrapper/network.py:203-215,300-329) and training-mode batch normalisation
// (network.py:492-509) folded into the consumer's load.
#include "common.h"

namespace {

struct InXform {
    const float* x; const float* s; const float* b;      // primary input and its BN affine (or null)
    const float* x2; const float* s2; const float* b2;   // optional skip input
};

__device__ __forceinline__ float load_in(const InXform& in, size_t off, int c) {
    float v = in.x[off];
    if (in.s) v = relu(v * in.s[c] + in.b[c]);
    if (in.x2) {
        float v2 = in.x2[off];
        if (in.s2) v2 = relu(v2 * in.s2[c] + in.b2[c]);
        v += v2;
    }
    return v;
}

constexpr int CO_CHUNK = 16;

// Block-level reduction of per-thread channel sums into the float64 accumulators.
__device__ __forceinline__ void stats_accumulate(const float (&acc)[CO_CHUNK], bool valid, int co0,
                                                 int Cout, double* stats) {
    __shared__ float red[4][2][CO_CHUNK];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CO_CHUNK; ++j) {
        float v = valid ? acc[j] : 0.f;
        float s = wave_sum(v), q = wave_sum(v * v);
        if (lane == 0) { red[wv][0][j] = s; red[wv][1][j] = q; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CO_CHUNK) {
        int k = threadIdx.x / CO_CHUNK, j = threadIdx.x % CO_CHUNK;
        if (co0 + j < Cout) {
            double t = (double)red[0][k][j] + (double)red[1][k][j] + (double)red[2][k][j] + (double)red[3][k][j];
            atomicAdd(&stats[(size_t)k * Cout + co0 + j], t);
        }
    }
    __syncthreads();
}

// Forward conv, stride 1 or 2, TensorFlow SAME padding (pad_before given per axis).
__global__ void __launch_bounds__(256)
conv3d_scalar_kernel(InXform in, const float* __restrict__ w, int D, int H, int W, int Cin, int Cout,
                     int stride, int Do, int Ho, int Wo, int pd, int ph, int pw,
                     float* __restrict__ y, double* __restrict__ stats) {
    long long vox = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nvox = (long long)Do * Ho * Wo;
    const bool valid = vox < nvox;
    long long vv = valid ? vox : 0;
    int ow = (int)(vv % Wo); int oh = (int)((vv / Wo) % Ho); int od = (int)(vv / ((long long)Wo * Ho));
    for (int co0 = 0; co0 < Cout; co0 += CO_CHUNK) {
        float acc[CO_CHUNK];
#pragma unroll
        for (int j = 0; j < CO_CHUNK; ++j) acc[j] = 0.f;
        if (valid) {
            for (int kd = 0; kd < 3; ++kd) {
                int id = od * stride + kd - pd; if (id < 0 || id >= D) continue;
                for (int kh = 0; kh < 3; ++kh) {
                    int ih = oh * stride + kh - ph; if (ih < 0 || ih >= H) continue;
                    for (int kw = 0; kw < 3; ++kw) {
                        int iw = ow * stride + kw - pw; if (iw < 0 || iw >= W) continue;
                        size_t base = (((size_t)id * H + ih) * W + iw) * Cin;
                        const float* wt = w + (size_t)((kd * 3 + kh) * 3 + kw) * Cin * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            float xv = load_in(in, base + ci, ci);
                            const float* wr = wt + (size_t)ci * Cout + co0;
#pragma unroll
                            for (int j = 0; j < CO_CHUNK; ++j)
                                if (co0 + j < Cout) acc[j] += xv * wr[j];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < CO_CHUNK; ++j)
                if (co0 + j < Cout) y[(size_t)vox * Cout + co0 + j] = acc[j];
        }
        if (stats) stats_accumulate(acc, valid, co0, Cout, stats);
    }
}

// Transposed conv, stride 2, SAME: out[o = 2i + k] += in[i] * w[k][co][ci], o in [0, 2n).
__global__ void __launch_bounds__(256)
deconv3d_scalar_kernel(InXform in, const float* __restrict__ w, int D, int H, int W, int Cin,
                       int Cout, float* __restrict__ y, double* __restrict__ stats) {
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
    long long vox = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nvox = (long long)Do * Ho * Wo;
    const bool valid = vox < nvox;
    long long vv = valid ? vox : 0;
    int ow = (int)(vv % Wo); int oh = (int)((vv / Wo) % Ho); int od = (int)(vv / ((long long)Wo * Ho));
    for (int co0 = 0; co0 < Cout; co0 += CO_CHUNK) {
        float acc[CO_CHUNK];
#pragma unroll
        for (int j = 0; j < CO_CHUNK; ++j) acc[j] = 0.f;
        if (valid) {
            for (int kd = 0; kd < 3; ++kd) {
                int td = od - kd; if (td < 0 || (td & 1)) continue; int id = td >> 1; if (id >= D) continue;
                for (int kh = 0; kh < 3; ++kh) {
                    int th = oh - kh; if (th < 0 || (th & 1)) continue; int ih = th >> 1; if (ih >= H) continue;
                    for (int kw = 0; kw < 3; ++kw) {
                        int tw = ow - kw; if (tw < 0 || (tw & 1)) continue; int iw = tw >> 1; if (iw >= W) continue;
                        size_t base = (((size_t)id * H + ih) * W + iw) * Cin;
                        const float* wt = w + (size_t)((kd * 3 + kh) * 3 + kw) * Cout * Cin;
                        for (int ci = 0; ci < Cin; ++ci) {
                            float xv = load_in(in, base + ci, ci);
#pragma unroll
                            for (int j = 0; j < CO_CHUNK; ++j)
                                if (co0 + j < Cout) acc[j] += xv * wt[(size_t)(co0 + j) * Cin + ci];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < CO_CHUNK; ++j)
                if (co0 + j < Cout) y[(size_t)vox * Cout + co0 + j] = acc[j];
        }
        if (stats) stats_accumulate(acc, valid, co0, Cout, stats);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float* __restrict__ scale, float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double mean = stats[c] / count;
    double var = stats[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    double inv = (double)gamma[c] / sqrt(var + (double)eps);
    scale[c] = (float)inv;
    shift[c] = (float)((double)beta[c] - mean * inv);
}

__global__ void zero_f64_kernel(double* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

}  // namespace

// ---- entry points used by the dispatcher in conv3d_mfma.hip -----------------------------------

int mvs_conv3d_scalar(const float* x, const float* xs, const float* xb, const float* x2,
                      const float* x2s, const float* x2b, const float* w, int D, int H, int W,
                      int Cin, int Cout, int stride, float* y, double* stats, hipStream_t st) {
    InXform in{x, xs, xb, x2, x2s, x2b};
    auto same = [](int n, int s, int& out, int& pb) {
        out = (n + s - 1) / s;
        int total = (out - 1) * s + 3 - n; if (total < 0) total = 0;
        pb = total / 2;
    };
    int Do, Ho, Wo, pd, ph, pw;
    same(D, stride, Do, pd); same(H, stride, Ho, ph); same(W, stride, Wo, pw);
    long long nvox = (long long)Do * Ho * Wo;
    conv3d_scalar_kernel<<<mvs_cdiv(nvox, 256), 256, 0, st>>>(in, w, D, H, W, Cin, Cout, stride,
                                                             Do, Ho, Wo, pd, ph, pw, y, stats);
    return (int)hipGetLastError();
}

int mvs_deconv3d_scalar(const float* x, const float* xs, const float* xb, const float* x2,
                        const float* x2s, const float* x2b, const float* w, int D, int H, int W,
                        int Cin, int Cout, float* y, double* stats, hipStream_t st) {
    InXform in{x, xs, xb, x2, x2s, x2b};
    long long nvox = 8LL * D * H * W;
    deconv3d_scalar_kernel<<<mvs_cdiv(nvox, 256), 256, 0, st>>>(in, w, D, H, W, Cin, Cout, y, stats);
    return (int)hipGetLastError();
}

extern "C" int mvs_bn_finalize_f32(const double* stats, int C, double count, const float* gamma,
                                   const float* beta, float eps, float* scale, float* shift,
                                   void* stream) {
    MVS_CHECK_ARG(stats && gamma && beta && scale && shift && C > 0 && count > 0);
    bn_finalize_kernel<<<mvs_cdiv(C, 64), 64, 0, mvs_stream(stream)>>>(stats, C, count, gamma, beta,
                                                                     eps, scale, shift);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_zero_f64(double* p, size_t n, void* stream) {
    MVS_CHECK_ARG(p && n > 0);
    zero_f64_kernel<<<mvs_cdiv((long long)n, 256), 256, 0, mvs_stream(stream)>>>(p, n);
    MVS_LAUNCH_RET();
}
