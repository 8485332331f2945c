// Shared by the UNetDS2GN tower kernels (unet2d.hip: one tile per workgroup; unet2d_p.hip: persistent workgroups).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// producer GroupNorm of one source, given as raw float64 group sums (or stats == null: identity)
struct GnSrc {
    const float* x;           // (V,H,W,C) raw producer output (or the image)
    const double* stats;      // (V, C/8, NSLOT, 2) [sum, sumsq] partial sums per view and group, or null
    const float* gamma; const float* beta;
    double count;             // elements per (view, group) = H*W*8
    int C;                    // channels of this source
    int relu;                 // ReLU after the affine (conv_gn) or not (deconv_gn, network.py:357)
};

struct Conv2dArgs {
    GnSrc a, b;               // b.x == null: single source
    const float* wprep;       // [cout group][chunk][tap][CK/4][16*MT][4]
    const float* wpair;       // pixel-pair layout behind it (unet2d_p.hip PAIR instances) or null
    float* y;                 // (V,Ho,Wo,Cout) raw
    double* stats;            // (V, Cout/8, NSLOT, 2) or null
    int V, H, W, Ho, Wo, Cout, pad_h, pad_w;
};

// GroupNorm sums are spread over GN_NSLOT partial accumulators per (view, group): a full-resolution layer of the one-tile
// kernel has ~13 000 workgroups adding into ~10 (view, group) pairs, and that many float64 atomics on one address serialise
// in L2 (measured: 287 us for a 20 us layer).  Consumers add the slots up.
constexpr int GN_NSLOT = 32;

// Persistent form (unet2d_p.hip): index of the instance for this layer shape (cg, mt as conv2d_tiling chose them: the two
// kernels share the prepared weight layout) or -1; mvs_conv2d_p_run launches it (MVS_E_SHAPE: sizes beyond its 32-bit offsets).
int mvs_conv2d_p_find(int ks, int stride, int cin, int cg, int mt, int cout);
// 3 x 3 layers with at most 8 output channels carry a second, pixel-pair weight layout behind the plain one in the prepared buffer
size_t mvs_conv2d_pair_floats(int ks, int stride, int cin, int cout);
int mvs_conv2d_pair_prepare(const float* w, int cin, int cout, int ck, float* out, hipStream_t st, int flipT, int cin_src = 0);
int mvs_conv2d_p_run(int inst, const Conv2dArgs& p, hipStream_t st);
