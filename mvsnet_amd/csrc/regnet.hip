// Host-side composition: conv dispatch (scalar vs MFMA), the RegNetUS0 3D U-Net launch sequence
// (mvsnet/cnn_wrapper/mvsnetworks.py:122-158) and library bookkeeping entry points.
// No kernels here; every launch goes to the caller's stream and nothing allocates or syncs.
#include "common.h"

// scalar path (conv3d_scalar.hip)
int mvs_conv3d_scalar(const float*, const float*, const float*, const float*, const float*,
                      const float*, const float*, int, int, int, int, int, int, float*, double*,
                      hipStream_t);
int mvs_deconv3d_scalar(const float*, const float*, const float*, const float*, const float*,
                        const float*, const float*, int, int, int, int, int, float*, double*,
                        hipStream_t);
// MFMA path (conv3d_mfma.hip); returns MVS_E_SHAPE when the shape is outside its tiling
int mvs_conv3d_mfma(const float*, const float*, const float*, const float*, const float*,
                    const float*, const float*, int, int, int, int, int, int, float*, double*,
                    hipStream_t);
int mvs_deconv3d_mfma(const float*, const float*, const float*, const float*, const float*,
                      const float*, const float*, int, int, int, int, int, float*, double*,
                      hipStream_t);

static int g_conv_impl = MVS_CONV_IMPL_AUTO;

extern "C" int mvs_abi_version(void) { return MVS_ABI_VERSION; }

extern "C" const char* mvs_error_string(int code) {
    if (code == 0) return "success";
    if (code == MVS_E_BADARG) return "mvsnet_hip: bad argument (null pointer or non-positive size)";
    if (code == MVS_E_SHAPE) return "mvsnet_hip: shape not supported by this kernel";
    if (code == MVS_E_WORKSPACE) return "mvsnet_hip: workspace too small";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "mvsnet_hip: unknown error";
}

extern "C" int mvs_set_conv_impl(int impl) {
    if (impl < MVS_CONV_IMPL_AUTO || impl > MVS_CONV_IMPL_MFMA) return MVS_E_BADARG;
    g_conv_impl = impl;
    return 0;
}
extern "C" int mvs_get_conv_impl(void) { return g_conv_impl; }

extern "C" int mvs_conv3d_f32(const float* x, const float* xs, const float* xb, const float* x2,
                              const float* x2s, const float* x2b, const float* w, int D, int H,
                              int W, int Cin, int Cout, int stride, float* y, double* stats,
                              void* stream) {
    MVS_CHECK_ARG(x && w && y && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    MVS_CHECK_ARG((xs == nullptr) == (xb == nullptr) && (x2s == nullptr) == (x2b == nullptr));
    MVS_CHECK_ARG(stride == 1 || stride == 2);
    hipStream_t st = mvs_stream(stream);
    if (g_conv_impl != MVS_CONV_IMPL_SCALAR) {
        int rc = mvs_conv3d_mfma(x, xs, xb, x2, x2s, x2b, w, D, H, W, Cin, Cout, stride, y, stats, st);
        if (rc != MVS_E_SHAPE || g_conv_impl == MVS_CONV_IMPL_MFMA) return rc;
    }
    return mvs_conv3d_scalar(x, xs, xb, x2, x2s, x2b, w, D, H, W, Cin, Cout, stride, y, stats, st);
}

extern "C" int mvs_deconv3d_f32(const float* x, const float* xs, const float* xb, const float* x2,
                                const float* x2s, const float* x2b, const float* w, int D, int H,
                                int W, int Cin, int Cout, float* y, double* stats, void* stream) {
    MVS_CHECK_ARG(x && w && y && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    MVS_CHECK_ARG((xs == nullptr) == (xb == nullptr) && (x2s == nullptr) == (x2b == nullptr));
    hipStream_t st = mvs_stream(stream);
    if (g_conv_impl != MVS_CONV_IMPL_SCALAR) {
        int rc = mvs_deconv3d_mfma(x, xs, xb, x2, x2s, x2b, w, D, H, W, Cin, Cout, y, stats, st);
        if (rc != MVS_E_SHAPE || g_conv_impl == MVS_CONV_IMPL_MFMA) return rc;
    }
    return mvs_deconv3d_scalar(x, xs, xb, x2, x2s, x2b, w, D, H, W, Cin, Cout, y, stats, st);
}

// ---- RegNetUS0 -----------------------------------------------------------------------------------

namespace {

constexpr int N_BN = 10;   // layers with BatchNorm, order: 1_0 2_0 3_0 0_1 1_1 2_1 3_1 4_0 5_0 6_0
enum { L10, L20, L30, L01, L11, L21, L31, L40, L50, L60, L62 };

struct RegnetWs {
    float* y[N_BN];        // raw (pre-BN) outputs
    float* scale[N_BN];
    float* shift[N_BN];
    double* stats;         // N_BN x 2 x cmax
    size_t bytes;
};

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

RegnetWs carve(char* base, int D, int H, int W, int cin, int b) {
    (void)cin;
    const size_t v0 = (size_t)D * H * W, v1 = v0 / 8, v2 = v1 / 8, v3 = v2 / 8;
    const size_t vox[N_BN] = {v1, v2, v3, v0, v1, v2, v3, v2, v1, v0};
    const int ch[N_BN] = {2 * b, 4 * b, 8 * b, b, 2 * b, 4 * b, 8 * b, 4 * b, 2 * b, b};
    const int cmax = 8 * b;
    size_t off = 0;
    RegnetWs w;
    for (int i = 0; i < N_BN; ++i) {
        w.y[i] = (float*)(base ? base + off : nullptr);
        off += align256(vox[i] * ch[i] * sizeof(float));
    }
    for (int i = 0; i < N_BN; ++i) {
        w.scale[i] = (float*)(base ? base + off : nullptr); off += align256(cmax * sizeof(float));
        w.shift[i] = (float*)(base ? base + off : nullptr); off += align256(cmax * sizeof(float));
    }
    w.stats = (double*)(base ? base + off : nullptr);
    off += align256((size_t)N_BN * 2 * cmax * sizeof(double));
    w.bytes = off;
    return w;
}

}  // namespace

extern "C" size_t mvs_regnet_workspace_bytes(int D, int H, int W, int cin, int base) {
    if (D <= 0 || H <= 0 || W <= 0 || cin <= 0 || base <= 0) return 0;
    return carve(nullptr, D, H, W, cin, base).bytes;
}

extern "C" int mvs_regnet_us0_f32(const float* cost, int D, int H, int W, int cin, int base,
                                  const float* const* weights, const float* const* gammas,
                                  const float* const* betas, float eps, void* workspace,
                                  size_t workspace_bytes, float* reg, void* stream) {
    MVS_CHECK_ARG(cost && weights && gammas && betas && workspace && reg);
    MVS_CHECK_ARG(D > 0 && H > 0 && W > 0 && cin > 0 && base > 0);
    if ((D % 8) || (H % 8) || (W % 8)) return MVS_E_SHAPE;
    RegnetWs ws = carve((char*)workspace, D, H, W, cin, base);
    if (workspace_bytes < ws.bytes) return MVS_E_WORKSPACE;
    const int b = base, cmax = 8 * b;
    const int D1 = D / 2, H1 = H / 2, W1 = W / 2, D2 = D / 4, H2 = H / 4, W2 = W / 4,
              D3 = D / 8, H3 = H / 8, W3 = W / 8;
    const double v0 = (double)D * H * W, v1 = v0 / 8, v2 = v1 / 8, v3 = v2 / 8;
    int rc;
    if ((rc = mvs_zero_f64(ws.stats, (size_t)N_BN * 2 * cmax, stream))) return rc;
    auto st = [&](int i) { return ws.stats + (size_t)i * 2 * cmax; };
    auto fin = [&](int i, int C, double cnt) {
        return mvs_bn_finalize_f32(st(i), C, cnt, gammas[i], betas[i], eps, ws.scale[i], ws.shift[i], stream);
    };
#define RUN(call) do { if ((rc = (call))) return rc; } while (0)
    // encoder on the raw cost volume (mvsnetworks.py:130-136)
    RUN(mvs_conv3d_f32(cost, 0, 0, 0, 0, 0, weights[L10], D, H, W, cin, 2 * b, 2, ws.y[L10], st(L10), stream));
    RUN(fin(L10, 2 * b, v1));
    RUN(mvs_conv3d_f32(cost, 0, 0, 0, 0, 0, weights[L01], D, H, W, cin, b, 1, ws.y[L01], st(L01), stream));
    RUN(fin(L01, b, v0));
    RUN(mvs_conv3d_f32(ws.y[L10], ws.scale[L10], ws.shift[L10], 0, 0, 0, weights[L20], D1, H1, W1, 2 * b, 4 * b, 2, ws.y[L20], st(L20), stream));
    RUN(fin(L20, 4 * b, v2));
    RUN(mvs_conv3d_f32(ws.y[L20], ws.scale[L20], ws.shift[L20], 0, 0, 0, weights[L30], D2, H2, W2, 4 * b, 8 * b, 2, ws.y[L30], st(L30), stream));
    RUN(fin(L30, 8 * b, v3));
    // same-resolution branches (mvsnetworks.py:138-145)
    RUN(mvs_conv3d_f32(ws.y[L10], ws.scale[L10], ws.shift[L10], 0, 0, 0, weights[L11], D1, H1, W1, 2 * b, 2 * b, 1, ws.y[L11], st(L11), stream));
    RUN(fin(L11, 2 * b, v1));
    RUN(mvs_conv3d_f32(ws.y[L20], ws.scale[L20], ws.shift[L20], 0, 0, 0, weights[L21], D2, H2, W2, 4 * b, 4 * b, 1, ws.y[L21], st(L21), stream));
    RUN(fin(L21, 4 * b, v2));
    RUN(mvs_conv3d_f32(ws.y[L30], ws.scale[L30], ws.shift[L30], 0, 0, 0, weights[L31], D3, H3, W3, 8 * b, 8 * b, 1, ws.y[L31], st(L31), stream));
    RUN(fin(L31, 8 * b, v3));
    // decoder with additive skips (mvsnetworks.py:146-157)
    RUN(mvs_deconv3d_f32(ws.y[L31], ws.scale[L31], ws.shift[L31], 0, 0, 0, weights[L40], D3, H3, W3, 8 * b, 4 * b, ws.y[L40], st(L40), stream));
    RUN(fin(L40, 4 * b, v2));
    RUN(mvs_deconv3d_f32(ws.y[L40], ws.scale[L40], ws.shift[L40], ws.y[L21], ws.scale[L21], ws.shift[L21], weights[L50], D2, H2, W2, 4 * b, 2 * b, ws.y[L50], st(L50), stream));
    RUN(fin(L50, 2 * b, v1));
    RUN(mvs_deconv3d_f32(ws.y[L50], ws.scale[L50], ws.shift[L50], ws.y[L11], ws.scale[L11], ws.shift[L11], weights[L60], D1, H1, W1, 2 * b, b, ws.y[L60], st(L60), stream));
    RUN(fin(L60, b, v0));
    // output conv, no BN / ReLU / bias (mvsnetworks.py:158)
    RUN(mvs_conv3d_f32(ws.y[L60], ws.scale[L60], ws.shift[L60], ws.y[L01], ws.scale[L01], ws.shift[L01], weights[L62], D, H, W, b, 1, 1, reg, nullptr, stream));
#undef RUN
    return 0;
}
