// Device helpers shared by the warp + variance kernels (cost_volume.hip and the lab variants under csrc/lab/): 16-byte
// loads, the per-tap projective bilinear sample of tf.contrib.image.transform (mvsnet/homography_warping.py:216-252).
// Included inside the including file's anonymous namespace.
#pragma once
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <typename R>
__device__ __forceinline__ float4 ldb(R rsrc, int byte_off) {      // buffer_load_dwordx4 ... offen
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

template <typename R>
__device__ __forceinline__ float4 ldbs(R rsrc, int byte_off, int uniform_off) {      // buffer_load_dwordx4 ... , s_off offen
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, uniform_off, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

__device__ __forceinline__ float4 fma4(float w, float4 a, float4 acc) {
    acc.x += w * a.x; acc.y += w * a.y; acc.z += w * a.z; acc.w += w * a.w;
    return acc;
}

// Bilinear sample of `img` (H,W,C) at the projective image of pixel (x,y); channels [c, c+4).
template <int BORDER>
__device__ __forceinline__ float4 warp_sample(const float* __restrict__ img, const float* __restrict__ t,
                                              float xf, float yf, int H, int W, int C, int c) {
    float proj = t[6] * xf + t[7] * yf + 1.0f;
    float sx = (t[0] * xf + t[1] * yf + t[2]) / proj;
    float sy = (t[3] * xf + t[4] * yf + t[5]) / proj;
    float x0 = floorf(sx), y0 = floorf(sy);
    float x1 = x0 + 1.0f, y1 = y0 + 1.0f;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BORDER == 0) {
        // zero fill: each tap individually reads 0 outside [0,W)x[0,H)
        bool okx0 = (x0 >= 0.0f) && (x0 < (float)W);
        bool okx1 = (x1 >= 0.0f) && (x1 < (float)W);
        bool oky0 = (y0 >= 0.0f) && (y0 < (float)H);
        bool oky1 = (y1 >= 0.0f) && (y1 < (float)H);
        int ix0 = okx0 ? (int)x0 : 0, ix1 = okx1 ? (int)x1 : 0;
        int iy0 = oky0 ? (int)y0 : 0, iy1 = oky1 ? (int)y1 : 0;
        float4 v00 = (okx0 && oky0) ? ld4(img + ((size_t)iy0 * W + ix0) * C + c) : z;
        float4 v01 = (okx1 && oky0) ? ld4(img + ((size_t)iy0 * W + ix1) * C + c) : z;
        float4 v10 = (okx0 && oky1) ? ld4(img + ((size_t)iy1 * W + ix0) * C + c) : z;
        float4 v11 = (okx1 && oky1) ? ld4(img + ((size_t)iy1 * W + ix1) * C + c) : z;
        float wx1 = x1 - sx, wx0 = sx - x0, wy1 = y1 - sy, wy0 = sy - y0;
        float4 vf, vc, o;
        vf.x = wx1 * v00.x + wx0 * v01.x; vf.y = wx1 * v00.y + wx0 * v01.y;
        vf.z = wx1 * v00.z + wx0 * v01.z; vf.w = wx1 * v00.w + wx0 * v01.w;
        vc.x = wx1 * v10.x + wx0 * v11.x; vc.y = wx1 * v10.y + wx0 * v11.y;
        vc.z = wx1 * v10.z + wx0 * v11.z; vc.w = wx1 * v10.w + wx0 * v11.w;
        o.x = wy1 * vf.x + wy0 * vc.x; o.y = wy1 * vf.y + wy0 * vc.y;
        o.z = wy1 * vf.z + wy0 * vc.z; o.w = wy1 * vf.w + wy0 * vc.w;
        return o;
    } else {
        // clamp-to-border variant (reference dead code, homography_warping.py:140-173)
        float mx = (float)(W - 1), my = (float)(H - 1);
        float cx0 = fminf(fmaxf(x0, 0.f), mx), cx1 = fminf(fmaxf(x1, 0.f), mx);
        float cy0 = fminf(fmaxf(y0, 0.f), my), cy1 = fminf(fmaxf(y1, 0.f), my);
        int ix0 = (int)cx0, ix1 = (int)cx1, iy0 = (int)cy0, iy1 = (int)cy1;
        float4 v00 = ld4(img + ((size_t)iy0 * W + ix0) * C + c);
        float4 v01 = ld4(img + ((size_t)iy0 * W + ix1) * C + c);
        float4 v10 = ld4(img + ((size_t)iy1 * W + ix0) * C + c);
        float4 v11 = ld4(img + ((size_t)iy1 * W + ix1) * C + c);
        float wa = (cy1 - sy) * (cx1 - sx), wb = (cy1 - sy) * (sx - cx0);
        float wc = (sy - cy0) * (cx1 - sx), wd = (sy - cy0) * (sx - cx0);
        float4 o = z;
        o = fma4(wa, v00, o); o = fma4(wb, v01, o); o = fma4(wc, v10, o); o = fma4(wd, v11, o);
        return o;
    }
}

// grid: x = ceil(H*W*(C/4) / 256), y = planes.  One lane = (pixel, 4 channels) of one plane.
