// Warp + variance cost volume with the bilinear blend on the matrix cores (C = 32, zero fill): R2 + R3 of SURVEY 8a,
// mvsnet/homography_warping.py:251-252 inside the cost loops of mvsnet/model.py:422-463, :315-334, :680-693.
//
// Why: the register-tap-cache sweep (cost_volume.hip) spends 70 M vector instructions per cost volume (143 per wave and
// plane: 60 for the blend itself, 83 for tap addresses, cache compares and the per-plane projective bookkeeping) and its
// 16-byte tap loads keep the texture path 70 % busy: vector ALU ~130 us, TA ~130 us, the 503 MB write ~80 us, kernel
// 186 us.  The matrix pipe idles.  A bilinear sample is a 1 x K by K x C product (K taps), and FOUR CONSECUTIVE PLANES of
// one pixel read (almost) the same source texels with different weights -- a 4 x K by K x C product per pixel and view.
// v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products per instruction (tools/mfma4x4_probe.hip pins its
// layout: D[reg i][lane l] = A[lane 4*(l/4) + i] * B[lane l]):
//      block  b = l / 4   <->  one of the wave's 16 pixels
//      row    i           <->  plane d0 + i            A: lane (b, i) holds ITS OWN bilinear weight of the current texel
//      column j = l % 4   <->  channels 16h + 4j + r   B: lane (b, j) holds the texel's channel 16h + 4j + r: per half h of
//                                                      the channels one 16-byte read and 4 MFMAs (r = 0..3)
// so a lane does the projective bookkeeping of exactly one (pixel, plane) -- no table, no redundancy -- reads 16 contiguous
// bytes per texel and channel half (the 4 lanes of a pixel read 64 contiguous bytes; no tap cache, no address compares),
// and the blend costs NO vector instructions: per view and group of 4 planes the vector ALU only adds the warped values
// into the running sums (S += w, Q += w*w: 32 packed instructions for 8 x 4 values per lane).
// The window of a pixel = the texels any of its 4 planes touches: origin = min over the planes of floor(sample point)
// (packed 16-bit min / max over the quad by DPP), a plane's weight for a window texel is its bilinear weight if the texel
// is one of its 4 taps and an exact zero otherwise (fma(0, t, acc) = acc: the nonzero taps accumulate in the order 00, 01,
// 10, 11 of the other kernels).  Sample points outside the image enter the window clamped to the border ring: their
// weights are zero anyway.  The window shape is made wave-uniform (ballots); 2x2 .. 4x2 and 2x3, 3x3 are unrolled (at the
// metric workload 63 % of the (wave, view, group)s are 3x2, 13 % 2x2, 12 % 4x2, 5 % 3x3); any other shape -- wide
// baselines, discontinuities, non-finite sample points -- takes the per-plane form (4 windows of 2x2, one per plane, the
// other planes' weights zero): exact for every geometry, bounded work.
// The reference view enters the sums through the same pipe: S = mfma(1, ref, 0) is the reference feature in all 4 planes.
//
// Two kernels:
//  * cost_volume_mfma_lds_kernel (n_src <= 4): north_star's LDS staging under the MFMA blend.  A workgroup = a 4 x 16
//    pixel tile x 8 planes; per view the bounding box of the tile's footprint over the plane run (8 corner samples, as
//    cost_volume_lds2_kernel) is staged into LDS once -- wide coalesced loads, ZEROS outside the image, so the zero fill
//    needs no weight masks -- together with the run's transforms; the B operands are ds_read_b128 (latency ~100 cycles
//    instead of an L2 / Infinity Cache round trip per view).  Exact by construction: a wave whose windows do not lie
//    inside the staged box takes the direct path for that view; a run whose box exceeds the LDS budget is retried as two
//    runs of 4 planes and otherwise takes the direct path entirely.
//  * cost_volume_mfma_kernel (any n_src): the direct path alone -- range-checked buffer loads, weight masks.
// Roofline: HBM write of the volume.
#include "../common.h"
#include <climits>
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <typename R>
__device__ __forceinline__ f32x4 ldb4(R rsrc, unsigned byte_off) {      // buffer_load_dwordx4 ... offen, range-checked
    u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, 0, 0);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// value of lane (4 * (l / 4) + SRC) in every lane of the quad
template <int SRC>
__device__ __forceinline__ int quad_bcast(int v) {
    return __builtin_amdgcn_update_dpp(v, v, SRC * 0x55, 0xf, 0xf, false);
}
__device__ __forceinline__ u16x2 as_u16x2(int v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ int as_int(u16x2 v) { return __builtin_bit_cast(int, v); }

// One (pixel, plane, view): sample point, its floor, the pixel's window over its 4 planes, the wave's window shape.
struct Sample {
    float sx, sy, x0, y0;
    int ix0, iy0;        // floor of the sample point (v_cvt saturates, NaN -> 0)
    int cx, cy;          // ... clamped to the border ring [-1, W-1] x [-1, H-1]
    int ox, oy;          // window origin of the pixel: min over its 4 planes of (cx, cy)
    bool x3, x4, y3;     // wave-uniform: some pixel's window is >= 3 / >= 4 texels wide, >= 3 texels high
    bool wide;           // wave-uniform: some window exceeds the unrolled shapes -> per-plane form
};

__device__ __forceinline__ Sample make_sample(const float4 ta, const float4 tb, float xf, float yf, int H, int W) {
    Sample s;
    const float proj = tb.z * xf + tb.w * yf + 1.0f;
    const float inv = __builtin_amdgcn_rcpf(proj);                      // v_rcp_f32: 1 ulp, exact for proj = 1
    s.sx = (ta.x * xf + ta.y * yf + ta.z) * inv;
    s.sy = (ta.w * xf + tb.x * yf + tb.y) * inv;
    s.x0 = floorf(s.sx); s.y0 = floorf(s.sy);
    s.ix0 = (int)s.x0; s.iy0 = (int)s.y0;
    s.cx = min(max(s.ix0, -1), W - 1); s.cy = min(max(s.iy0, -1), H - 1);
    // min and max of (cx + 1, cy + 1) over the quad, both coordinates at once (H, W < 65535: host-checked)
    const int p = ((s.cy + 1) << 16) | (s.cx + 1);
    const int p1 = __builtin_amdgcn_update_dpp(p, p, 0xB1, 0xf, 0xf, false);             // quad_perm [1,0,3,2]
    const int lo1 = as_int(__builtin_elementwise_min(as_u16x2(p), as_u16x2(p1)));
    const int hi1 = as_int(__builtin_elementwise_max(as_u16x2(p), as_u16x2(p1)));
    const int lo2 = __builtin_amdgcn_update_dpp(lo1, lo1, 0x4E, 0xf, 0xf, false);        // quad_perm [2,3,0,1]
    const int hi2 = __builtin_amdgcn_update_dpp(hi1, hi1, 0x4E, 0xf, 0xf, false);
    const u16x2 lo = __builtin_elementwise_min(as_u16x2(lo1), as_u16x2(lo2));
    const u16x2 hi = __builtin_elementwise_max(as_u16x2(hi1), as_u16x2(hi2));
    s.ox = (int)lo[0] - 1; s.oy = (int)lo[1] - 1;
    const int wx = (int)hi[0] - (int)lo[0], wy = (int)hi[1] - (int)lo[1];               // window = (wx + 2) x (wy + 2) texels
    s.x3 = __builtin_amdgcn_ballot_w64(wx > 0) != 0; s.x4 = __builtin_amdgcn_ballot_w64(wx > 1) != 0;
    s.y3 = __builtin_amdgcn_ballot_w64(wy > 0) != 0;
    const bool x5 = __builtin_amdgcn_ballot_w64(wx > 2) != 0, y4 = __builtin_amdgcn_ballot_w64(wy > 1) != 0;
    s.wide = x5 || y4 || (s.x4 && s.y3);
    return s;
}

__device__ __forceinline__ void add_view(const f32x4 (&D)[4], f32x4* S, f32x4* Q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { S[r] += D[r]; Q[r] += D[r] * D[r]; }
}

// ---- direct path: texels through range-checked buffer loads ---------------------------------------------------------
// One channel half of one window: WX x WY texels from byte offset `base` (this lane's 16 bytes of the window's first
// texel); wxs / wys are this lane's (= its plane's) weights of the window's columns and rows.  All loads are issued
// before the first MFMA.  `fresh`: the accumulators start from zero with the first texel.
template <int WX, int WY, typename R>
__device__ __forceinline__ void blend_half(R rsrc, unsigned base, unsigned row_bytes, const float (&wxs)[WX],
                                           const float (&wys)[WY], bool fresh, f32x4 (&D)[4]) {
    f32x4 tp[WY][WX];
#pragma unroll
    for (int b = 0; b < WY; ++b)
#pragma unroll
        for (int a = 0; a < WX; ++a) tp[b][a] = ldb4(rsrc, base + (unsigned)b * row_bytes + (unsigned)a * 128u);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < WY; ++b)
#pragma unroll
        for (int a = 0; a < WX; ++a) {
            const float w = wys[b] * wxs[a];
            const bool first = fresh && a == 0 && b == 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                D[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, tp[b][a][r], first ? zero : D[r], 0, 0, 0);
        }
}

template <int WX, int WY, typename R>
__device__ __forceinline__ void view_shared(R rsrc, unsigned base, unsigned row_bytes, int relx, int rely, float wx1,
                                            float wx0, float wy1, float wy0, f32x4 (&S)[8], f32x4 (&Q)[8]) {
    float wxs[WX], wys[WY];
#pragma unroll
    for (int a = 0; a < WX; ++a) wxs[a] = a == relx ? wx1 : (a == relx + 1 ? wx0 : 0.0f);
#pragma unroll
    for (int b = 0; b < WY; ++b) wys[b] = b == rely ? wy1 : (b == rely + 1 ? wy0 : 0.0f);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 D[4];
        blend_half<WX, WY>(rsrc, base + 64u * h, row_bytes, wxs, wys, true, D);
        add_view(D, S + 4 * h, Q + 4 * h);
    }
}

// One view of one group of 4 planes through buffer loads: zero fill by weight masks (as cost_volume_sweep_kernel).
// vbase = byte offset of texel (-1, -1) of the view + this lane's 16 bytes.
template <typename R>
__device__ __forceinline__ void view_direct(R rsrc, const Sample& s, unsigned vbase, unsigned row_bytes, int t, int H, int W,
                                            f32x4 (&S)[8], f32x4 (&Q)[8]) {
    const float wx1 = (unsigned)s.ix0 < (unsigned)W ? (s.x0 + 1.0f) - s.sx : 0.0f;      // column ix0 inside the image
    const float wx0 = (unsigned)(s.ix0 + 1) < (unsigned)W ? s.sx - s.x0 : 0.0f;
    const float wy1 = (unsigned)s.iy0 < (unsigned)H ? (s.y0 + 1.0f) - s.sy : 0.0f;
    const float wy0 = (unsigned)(s.iy0 + 1) < (unsigned)H ? s.sy - s.y0 : 0.0f;
    if (!s.wide) {                                                      // shared window, wave-uniform shape
        const unsigned base = vbase + __umul24((unsigned)(s.oy + 1), row_bytes) + ((unsigned)(s.ox + 1) << 7);
        const int relx = s.ix0 - s.ox, rely = s.iy0 - s.oy;
        if (!s.y3) {
            if (!s.x3) view_shared<2, 2>(rsrc, base, row_bytes, relx, rely, wx1, wx0, wy1, wy0, S, Q);
            else if (!s.x4) view_shared<3, 2>(rsrc, base, row_bytes, relx, rely, wx1, wx0, wy1, wy0, S, Q);
            else view_shared<4, 2>(rsrc, base, row_bytes, relx, rely, wx1, wx0, wy1, wy0, S, Q);
        } else {
            if (!s.x3) view_shared<2, 3>(rsrc, base, row_bytes, relx, rely, wx1, wx0, wy1, wy0, S, Q);
            else view_shared<3, 3>(rsrc, base, row_bytes, relx, rely, wx1, wx0, wy1, wy0, S, Q);
        }
    } else {                                                            // per-plane windows: exact for any geometry
        // (a clamped sample has zero weights in the clamped axis, so its taps may sit anywhere)
        const unsigned own = (unsigned)t * 16u;
        const int pb = (int)(vbase - own + __umul24((unsigned)(s.cy + 1), row_bytes) + ((unsigned)(s.cx + 1) << 7));
        const unsigned pbs[4] = {(unsigned)quad_bcast<0>(pb) + own, (unsigned)quad_bcast<1>(pb) + own,
                                 (unsigned)quad_bcast<2>(pb) + own, (unsigned)quad_bcast<3>(pb) + own};
        const float wys[2] = {wy1, wy0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 D[4];
#pragma unroll
            for (int ip = 0; ip < 4; ++ip) {
                const float wxs[2] = {t == ip ? wx1 : 0.0f, t == ip ? wx0 : 0.0f};
                blend_half<2, 2>(rsrc, pbs[ip] + 64u * h, row_bytes, wxs, wys, ip == 0, D);
            }
            add_view(D, S + 4 * h, Q + 4 * h);
        }
    }
}

// ---- LDS path: texels from the staged box (zeros outside the image: plain tent weights, no masks) ---------------------
template <int WX, int WY>
__device__ __forceinline__ void view_lds(const char* lbase, int pitch, float sxo, float syo, f32x4 (&S)[8], f32x4 (&Q)[8]) {
    float wxs[WX], wys[WY];
    // bilinear weight of window column a for a sample at sxo (relative to the window origin): the tent max(0, 1 - |a - sxo|)
    // = (x0 + 1) - sx on the sample's left tap column, sx - x0 on its right one (both differences are exact in float32
    // for in-image coordinates), 0 elsewhere and for non-finite sample points
#pragma unroll
    for (int a = 0; a < WX; ++a) wxs[a] = fmaxf(1.0f - fabsf((float)a - sxo), 0.0f);
#pragma unroll
    for (int b = 0; b < WY; ++b) wys[b] = fmaxf(1.0f - fabsf((float)b - syo), 0.0f);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 tp[WY][WX];
#pragma unroll
        for (int b = 0; b < WY; ++b)
#pragma unroll
            for (int a = 0; a < WX; ++a) tp[b][a] = *reinterpret_cast<const f32x4*>(lbase + b * pitch + a * 128 + h * 64);
        f32x4 D[4];
#pragma unroll
        for (int b = 0; b < WY; ++b)
#pragma unroll
            for (int a = 0; a < WX; ++a) {
                const float w = wys[b] * wxs[a];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    D[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, tp[b][a][r], (a == 0 && b == 0) ? zero : D[r], 0, 0, 0);
            }
        add_view(D, S + 4 * h, Q + 4 * h);
    }
}

// variance over the views (in place in S) and the store of a group: lane (pixel, t) holds channels 16h + 4t + r of planes
// dlb .. dlb+3
__device__ __forceinline__ void finish_group(f32x4 (&S)[8], f32x4 (&Q)[8], float inv_n, float inv_nn, int variant, int negate,
                                             bool pix_ok, int dlb, int dl1, float* dst, size_t plane_stride) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        if (variant == 0) S[r] = Q[r] * inv_n - (S[r] * S[r]) * inv_nn;             // inference_mem (model.py:458-461)
        else { const f32x4 m = S[r] * inv_n; S[r] = Q[r] * inv_n - m * m; }         // inference / GRU (model.py:330-332)
        if (negate) S[r] = -S[r];
    }
    if (pix_ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (dlb + i < dl1) {
                *reinterpret_cast<float4*>(dst + (size_t)i * plane_stride) = make_float4(S[0][i], S[1][i], S[2][i], S[3][i]);
                *reinterpret_cast<float4*>(dst + (size_t)i * plane_stride + 16) = make_float4(S[4][i], S[5][i], S[6][i], S[7][i]);
            }
    }
}

// ---- direct kernel: grid x = ceil(H*W / 64) (a wave = 16 consecutive pixels), y = plane chunks (a multiple of 4) -------
__global__ void __launch_bounds__(256, 2)
cost_volume_mfma_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ transforms,
                        int n_src, int depth_total, int d_begin, int d_count, int planes_per_block, int H, int W,
                        int variant, int negate, float* __restrict__ cost) {
    constexpr int C = 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = lane >> 2, t = lane & 3;
    const long long HW = (long long)H * W;
    const long long pix_raw = ((long long)xcd_swizzle(blockIdx.x, gridDim.x) * 4 + wave) * 16 + blk;
    if (pix_raw - blk >= HW) return;                                   // the whole wave is past the image (wave-uniform)
    const bool pix_ok = pix_raw < HW;
    const long long pix = pix_ok ? pix_raw : HW - 1;
    const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    const float xf = (float)x, yf = (float)y;
    const int dl0 = blockIdx.y * planes_per_block, dl1 = min(dl0 + planes_per_block, d_count);

    // reference view: this lane's channels 16h + 4t + r of the pixel
    const float4 r0 = ld4g(ref + (size_t)pix * C + 4 * t), r1 = ld4g(ref + (size_t)pix * C + 16 + 4 * t);
    const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const unsigned row_bytes = (unsigned)W * C * 4u, img_bytes = (unsigned)H * row_bytes;
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((unsigned)n_src * img_bytes), 0x00020000);
    const float n = (float)(n_src + 1);
    const float inv_n = 1.0f / n, inv_nn = 1.0f / (n * n);             // as cost_volume_sweep_kernel
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    for (int dlb = dl0; dlb < dl1; dlb += 4) {
        const int my_plane = d_begin + min(dlb + t, dl1 - 1);           // this lane's plane of the group (tail: repeats)
        const float* tr = transforms + (size_t)my_plane * 8;
        float4 ta = ld4g(tr), tb = ld4g(tr + 4);                        // view 0; the next view's are fetched a view ahead
        f32x4 S[8], Q[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {                                   // the reference feature in all 4 planes
            S[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, rr[r], zero, 0, 0, 0);
            Q[r] = S[r] * S[r];
        }
#pragma unroll 1
        for (int v = 0; v < n_src; ++v) {
            const Sample s = make_sample(ta, tb, xf, yf, H, W);
            {
                const float* tn = tr + (size_t)min(v + 1, n_src - 1) * depth_total * 8;
                ta = ld4g(tn); tb = ld4g(tn + 4);
            }
            view_direct(srsrc, s, (unsigned)v * img_bytes + (unsigned)t * 16u - row_bytes - 128u, row_bytes, t, H, W, S, Q);
        }
        finish_group(S, Q, inv_n, inv_nn, variant, negate, pix_ok, dlb, dl1, cost + ((size_t)dlb * HW + pix) * C + 4 * t,
                     (size_t)HW * C);
    }
}

// ---- LDS-staged kernel: grid x = tiles of 4 x 16 pixels, y = runs of 8 planes ----------------------------------------------
constexpr int CVM_TH = 4, CVM_TW = 16, CVM_LP = 8;

template <int NSRC, int CAP>
__global__ void __launch_bounds__(256, 2)
cost_volume_mfma_lds_kernel(const float* __restrict__ ref, const float* __restrict__ src, const float* __restrict__ transforms,
                            int depth_total, int d_begin, int d_count, int H, int W, int variant, int negate, int tiles_x,
                            float* __restrict__ cost) {
    constexpr int C = 32;
    extern __shared__ __attribute__((aligned(16))) float4 cvm_smem[];
    float4* box = cvm_smem;                                            // [NSRC][CAP texels][8 float4]
    float4* ttab = cvm_smem + NSRC * CAP * 8;                          // [NSRC][CVM_LP][2]: the run's transforms
    int4* boxp = reinterpret_cast<int4*>(ttab + NSRC * CVM_LP * 2);    // [NSRC]: box origin and size

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = lane >> 2, t = lane & 3;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x_raw = tx * CVM_TW + blk, y_raw = ty * CVM_TH + wave;
    const bool pix_ok = x_raw < W && y_raw < H;
    const int x = min(x_raw, W - 1), y = min(y_raw, H - 1);
    const float xf = (float)x, yf = (float)y;
    const long long HW = (long long)H * W;
    const long long pix = (long long)y * W + x;
    const int dl0 = blockIdx.y * CVM_LP, dl1 = min(dl0 + CVM_LP, d_count);

    const float4 r0 = ld4g(ref + (size_t)pix * C + 4 * t), r1 = ld4g(ref + (size_t)pix * C + 16 + 4 * t);
    const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    const unsigned row_bytes = (unsigned)W * C * 4u, img_bytes = (unsigned)H * row_bytes;
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((unsigned)NSRC * img_bytes), 0x00020000);
    const float n = (float)(NSRC + 1);
    const float inv_n = 1.0f / n, inv_nn = 1.0f / (n * n);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    int run = CVM_LP;                                                  // planes per staging; halved once if the box is too big
    for (int p0 = dl0; p0 < dl1;) {
        const int p1 = min(p0 + run, dl1);
        // ---- box of every view over planes [p0, p1), computed by every wave alike: lane = (view, corner) ----------------
        int bx0[NSRC], by0[NSRC], bw[NSRC], bh[NSRC];
        bool fits = true;
        {
            const int v = min(lane >> 3, NSRC - 1), k = lane & 7;
            const float cxf = (float)((k & 1) ? min(tx * CVM_TW + CVM_TW - 1, W - 1) : tx * CVM_TW);
            const float cyf = (float)((k & 2) ? min(ty * CVM_TH + CVM_TH - 1, H - 1) : ty * CVM_TH);
            const float* tc = transforms + ((size_t)v * depth_total + d_begin + ((k & 4) ? p1 - 1 : p0)) * 8;
            const float4 ca = ld4g(tc), cb = ld4g(tc + 4);
            const float proj = cb.z * cxf + cb.w * cyf + 1.0f;
            const float inv = __builtin_amdgcn_rcpf(proj);
            const float sx = (ca.x * cxf + ca.y * cyf + ca.z) * inv, sy = (ca.w * cxf + cb.x * cyf + cb.y) * inv;
            int lox = (int)floorf(sx), loy = (int)floorf(sy), hix = lox, hiy = loy;
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                lox = min(lox, __shfl_xor(lox, o, 64)); loy = min(loy, __shfl_xor(loy, o, 64));
                hix = max(hix, __shfl_xor(hix, o, 64)); hiy = max(hiy, __shfl_xor(hiy, o, 64));
            }
#pragma unroll
            for (int vv = 0; vv < NSRC; ++vv) {
                // clamped to the border ring like the windows; one texel of margin on the high side (the windows of
                // neighbouring pixels share the wave's shape)
                const int lx = min(max(__builtin_amdgcn_readlane(lox, vv * 8), -1), W - 1);
                const int ly = min(max(__builtin_amdgcn_readlane(loy, vv * 8), -1), H - 1);
                const int hx = min(max(__builtin_amdgcn_readlane(hix, vv * 8), -1), W - 1);
                const int hy = min(max(__builtin_amdgcn_readlane(hiy, vv * 8), -1), H - 1);
                bx0[vv] = lx; by0[vv] = ly; bw[vv] = hx - lx + 3; bh[vv] = hy - ly + 3;
                fits = fits && bw[vv] * bh[vv] <= CAP;
            }
        }
        if (!fits && p1 - p0 > 4) { run = 4; continue; }              // block-uniform: every wave computed the same boxes
        __syncthreads();                                               // the previous run's reads of the LDS are done
        if (fits) {
            constexpr int NIT = (CAP * 8 + 255) / 256;
            u32x4_t st[NSRC][NIT];
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const int npos = bw[v] * bh[v];
                const int m = (65536 + bw[v] - 1) / bw[v];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int pos = (tid + 256 * it) >> 3;
                    const int r = (pos * m) >> 16, cc = pos - r * bw[v];
                    const int gy = by0[v] + r, gx = bx0[v] + cc;
                    const bool ok = pos < npos && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    st[v][it] = __builtin_amdgcn_raw_buffer_load_b128(
                        srsrc, ok ? (int)((((unsigned)(v * H + gy) * W + gx) << 7) + (tid & 7) * 16) : (int)0x80000000u, 0, 0);
                }
            }
            if (tid < NSRC * CVM_LP * 2) {                             // transforms of the run: [view][plane][2 float4]
                const int v = tid / (CVM_LP * 2), k = (tid >> 1) & (CVM_LP - 1), half = tid & 1;
                ttab[tid] = ld4g(transforms + ((size_t)v * depth_total + d_begin + min(p0 + k, p1 - 1)) * 8 + 4 * half);
            }
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                if (tid == v) boxp[v] = make_int4(bx0[v], by0[v], bw[v], bh[v]);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = tid + 256 * it;
                    if (i < CAP * 8)                                   // the whole slab: texels past the box are zeros
                        box[v * CAP * 8 + i] = make_float4(__uint_as_float(st[v][it][0]), __uint_as_float(st[v][it][1]),
                                                           __uint_as_float(st[v][it][2]), __uint_as_float(st[v][it][3]));
                }
            }
        }
        __syncthreads();
        // ---- the groups of 4 planes of the run -----------------------------------------------------------------------------
        for (int dlb = p0; dlb < p1; dlb += 4) {
            const int kplane = min(dlb + t, p1 - 1);                    // this lane's plane of the group (tail: repeats)
            f32x4 S[8], Q[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                S[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, rr[r], zero, 0, 0, 0);
                Q[r] = S[r] * S[r];
            }
#pragma unroll 1
            for (int v = 0; v < NSRC; ++v) {
                float4 ta, tb;
                if (fits) { ta = ttab[(v * CVM_LP + kplane - p0) * 2]; tb = ttab[(v * CVM_LP + kplane - p0) * 2 + 1]; }
                else { const float* tr = transforms + ((size_t)v * depth_total + d_begin + kplane) * 8; ta = ld4g(tr); tb = ld4g(tr + 4); }
                const Sample s = make_sample(ta, tb, xf, yf, H, W);
                bool staged = fits && !s.wide;
                int rx = 0, ry = 0, pitch = 0;
                if (staged) {                                           // do this wave's windows lie inside the staged box?
                    const int4 bp = boxp[v];
                    rx = s.ox - bp.x; ry = s.oy - bp.y; pitch = bp.z * 128;
                    const int wxu = 2 + (s.x3 ? 1 : 0) + (s.x4 ? 1 : 0), wyu = 2 + (s.y3 ? 1 : 0);
                    const bool outside = rx < 0 || ry < 0 || rx + wxu > bp.z || ry + wyu > bp.w;
                    staged = __builtin_amdgcn_ballot_w64(outside) == 0;
                }
                if (staged) {
                    const char* lbase = reinterpret_cast<const char*>(box) + ((v * CAP + ry * (pitch >> 7) + rx) << 7) + t * 16;
                    const float sxo = s.sx - (float)s.ox, syo = s.sy - (float)s.oy;
                    if (!s.y3) {
                        if (!s.x3) view_lds<2, 2>(lbase, pitch, sxo, syo, S, Q);
                        else if (!s.x4) view_lds<3, 2>(lbase, pitch, sxo, syo, S, Q);
                        else view_lds<4, 2>(lbase, pitch, sxo, syo, S, Q);
                    } else {
                        if (!s.x3) view_lds<2, 3>(lbase, pitch, sxo, syo, S, Q);
                        else view_lds<3, 3>(lbase, pitch, sxo, syo, S, Q);
                    }
                } else {
                    view_direct(srsrc, s, (unsigned)v * img_bytes + (unsigned)t * 16u - row_bytes - 128u, row_bytes, t, H, W, S, Q);
                }
            }
            finish_group(S, Q, inv_n, inv_nn, variant, negate, pix_ok, dlb, dl1, cost + ((size_t)dlb * HW + pix) * C + 4 * t,
                         (size_t)HW * C);
        }
        p0 = p1;
    }
}

template <int NSRC>
int launch_mfma_lds(const float* ref, const float* src, const float* transforms, int depth_total, int d_begin, int d_count,
                    int H, int W, int variant, int negate, float* cost, hipStream_t st) {
    constexpr int CAP = NSRC <= 2 ? 304 : (NSRC == 3 ? 200 : 152);    // texels per view: two workgroups per CU (160 KB LDS)
    const int tiles_x = mvs_cdiv(W, CVM_TW), tiles_y = mvs_cdiv(H, CVM_TH);
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)mvs_cdiv(d_count, CVM_LP));
    const size_t smem = (size_t)(NSRC * CAP * 8 + NSRC * CVM_LP * 2) * sizeof(float4) + NSRC * sizeof(int4);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)cost_volume_mfma_lds_kernel<NSRC, CAP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    cost_volume_mfma_lds_kernel<NSRC, CAP><<<grid, 256, smem, st>>>(ref, src, transforms, depth_total, d_begin, d_count, H, W,
                                                                    variant, negate, tiles_x, cost);
    return (int)hipGetLastError();
}

}  // namespace

// LAB VARIANT entry point (not in libmvsnet_hip.so; `python -m mvsnet_amd.build --lab`, tests/test_gpu_lab.py).  C = 32, zero
// fill, all source maps together below 2 GiB, H, W < 65535.  direct != 0 forces the direct kernel (no LDS staging).
extern "C" int mvs_lab_cost_volume_mfma_f32(const float* ref, const float* src, const float* transforms, int view_num,
                                            int depth_total, int d_begin, int d_count, int H, int W, int C, int variant,
                                            int negate, int direct, float* cost, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && cost);
    MVS_CHECK_ARG(view_num >= 2 && depth_total >= 1 && d_begin >= 0 && d_count >= 1 && d_begin + d_count <= depth_total && H > 0 && W > 0);
    if (C != 32 || H >= 65535 || W >= 65535 || (long long)(view_num - 1) * H * W * C * 4 >= (1LL << 31)) return MVS_E_SHAPE;
    const int n_src = view_num - 1;
    hipStream_t st = mvs_stream(stream);
    if (!direct) {
        switch (n_src) {
            case 1: return launch_mfma_lds<1>(ref, src, transforms, depth_total, d_begin, d_count, H, W, variant, negate, cost, st);
            case 2: return launch_mfma_lds<2>(ref, src, transforms, depth_total, d_begin, d_count, H, W, variant, negate, cost, st);
            case 3: return launch_mfma_lds<3>(ref, src, transforms, depth_total, d_begin, d_count, H, W, variant, negate, cost, st);
            case 4: return launch_mfma_lds<4>(ref, src, transforms, depth_total, d_begin, d_count, H, W, variant, negate, cost, st);
            default: break;
        }
    }
    int ppb = 16;
    if (ppb > (d_count + 3) / 4 * 4) ppb = (d_count + 3) / 4 * 4;
    const long long hw = (long long)H * W;
    dim3 grid((unsigned)mvs_cdiv(hw, 64), (unsigned)mvs_cdiv(d_count, ppb));
    cost_volume_mfma_kernel<<<grid, 256, 0, st>>>(ref, src, transforms, n_src, depth_total, d_begin, d_count, ppb, H, W,
                                                  variant, negate, cost);
    return (int)hipGetLastError();
}
