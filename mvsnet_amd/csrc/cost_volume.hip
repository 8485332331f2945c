// Fused projective bilinear warp + cross-view variance -> (D,H,W,C) cost volume (R2 + R3).
// Reference behaviour: tf.contrib.image.transform BILINEAR with zero fill per tap
// (mvsnet/homography_warping.py:251-252) inside the cost loops of mvsnet/model.py:422-463
// (inference_mem), :315-334 (inference) and :680-693 (GRU body).
//
// Roofline: HBM write of the volume (D*H*W*C*4 bytes) + one read of the N feature maps.
// Layout: channel-last; one lane owns 4 consecutive channels (16 B) of one voxel, so the C/4
// lanes of a voxel read each bilinear tap as one contiguous C*4-byte segment and the block's
// stores are fully coalesced 16-B-per-lane rows of the volume.  Feature maps (N*H*W*C*4 bytes,
// 13 MB at the metric config) stay resident in L2 / Infinity Cache across the D planes.
#include "common.h"
#include <climits>
#include <cstdlib>


namespace {


#include "cost_volume_common.h"

template <int BORDER>
__global__ void __launch_bounds__(256)
cost_volume_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                   const float* __restrict__ transforms, int n_src, int depth_total, int d_begin,
                   int H, int W, int C, int variant, int negate, float* __restrict__ cost) {
    const int cq = C >> 2;
    const long long total = (long long)H * W * cq;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int dl = blockIdx.y;            // plane index inside the output
    const int d = d_begin + dl;           // plane index inside the transform table
    int c = (int)(idx % cq) * 4;
    long long pix = idx / cq;
    int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    float xf = (float)x, yf = (float)y;

    float4 r = ld4(ref + (size_t)pix * C + c);
    float4 S = r;
    float4 Q = make_float4(r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w);
    const size_t img_elems = (size_t)H * W * C;
    for (int v = 0; v < n_src; ++v) {
        const float* t = transforms + ((size_t)v * depth_total + d) * 8;   // block-uniform
        float4 w = warp_sample<BORDER>(src + v * img_elems, t, xf, yf, H, W, C, c);
        S.x += w.x; S.y += w.y; S.z += w.z; S.w += w.w;
        Q.x += w.x * w.x; Q.y += w.y * w.y; Q.z += w.z * w.z; Q.w += w.w * w.w;
    }
    const float n = (float)(n_src + 1);
    float4 o;
    if (variant == 0) {          // inference_mem: Q/N - S*S/(N*N)      (model.py:458-461)
        const float nn = n * n;
        o.x = Q.x / n - (S.x * S.x) / nn; o.y = Q.y / n - (S.y * S.y) / nn;
        o.z = Q.z / n - (S.z * S.z) / nn; o.w = Q.w / n - (S.w * S.w) / nn;
    } else {                     // inference / GRU: Q/N - (S/N)^2      (model.py:330-332)
        float ax = S.x / n, ay = S.y / n, az = S.z / n, aw = S.w / n;
        o.x = Q.x / n - ax * ax; o.y = Q.y / n - ay * ay;
        o.z = Q.z / n - az * az; o.w = Q.w / n - aw * aw;
    }
    if (negate) { o.x = -o.x; o.y = -o.y; o.z = -o.z; o.w = -o.w; }
    *reinterpret_cast<float4*>(cost + ((size_t)dl * H * W + pix) * C + c) = o;
}

// Depth-sweep variant (the 3D-CNN path): one lane = (pixel, 4 channels) sweeping DC consecutive
// planes.  Along depth a pixel's sample point slides along its epipolar line by a fraction of a
// pixel per plane (that is how plane-sweep depth intervals are chosen), so its 2x2 tap
// neighbourhood is kept in registers and re-fetched only when floor(sx) or floor(sy) changes:
// tap traffic to L1/L2 drops from 16 x 16 B per voxel-lane to a few, the kernel becomes bound by
// the coalesced HBM write of the volume.  Zero fill per tap exactly as warp_sample<0>.
// VN = 2 * variant + negate is a template parameter: as kernel arguments both forms of the variance and both signs were
// computed for every voxel and selected (12 packed + 8 select instructions per plane where 6 packed do, in a kernel that is
// bound by vector issue) -- *r5*.
template <int NSRC, int Q, int VN>       // Q float4 per lane: 4*Q channels per lane, C/(4*Q) lanes per pixel
__global__ void __launch_bounds__(256)
cost_volume_sweep_kernel(const float* __restrict__ ref, const float* __restrict__ src,
                         const float* __restrict__ transforms, int depth_total, int d_begin,
                         int d_count, int planes_per_block, int H, int W, int C, int ty_log2,
                         float* __restrict__ cost) {
    constexpr int variant = VN >> 1;
    constexpr bool negate = (VN & 1) != 0;
    const int lg = C / (4 * Q);                           // lanes per pixel: power of two (host-checked)
    // A wave's 64/lg pixels are a TY x TX tile, the workgroup's waves sit side by side along x (*r5*).  What a wave pays for
    // vector memory is the number of load INSTRUCTIONS (16 clocks per 16-byte-per-lane load on the CU's texture path whether
    // all lanes, one lane or none is active: tools/exec0_vmem_probe.hip), i.e. the number of (plane, view) in which ANY of its
    // pixels' blocks moved.  Pixels across the epipolar direction cross tap boundaries on the same plane; pixels along it do
    // not.  At the metric workload (horizontal baselines) a pixel's block moves on 23 % of the (plane, view); a wave of 8 pixels
    // along x has a move on 44 %, 4 rows x 2 columns on 26 %, a column of 8 on 23 %: 174 / 154-159 / 149-151 us, 945 / 964 / 973 depth maps/s.
    const int ppw = 64 / lg;                                 // pixels per wave
    if (ty_log2 < 0) {
        // The shape is voted here, the same in every workgroup: the sample point of the image centre moves by about (mx, my)
        // source pixels over the whole sweep, summed over the views (numerators only: a direction is all that is needed, and every
        // wave pays for these instructions); mostly along x -> a column of pixels, mostly along y -> a row, otherwise the
        // squarer shape that is longer across the stronger direction.
        float mx = 0.0f, my = 0.0f;
        const float cx = 0.5f * (float)W, cy = 0.5f * (float)H;
#pragma unroll
        for (int v = 0; v < NSRC; ++v) {
            const float* t0 = transforms + (size_t)v * depth_total * 8;
            const float* t1 = t0 + (size_t)(depth_total - 1) * 8;
            mx += fabsf((t1[0] - t0[0]) * cx + (t1[1] - t0[1]) * cy + (t1[2] - t0[2]));
            my += fabsf((t1[3] - t0[3]) * cx + (t1[4] - t0[4]) * cy + (t1[5] - t0[5]));
        }
        ty_log2 = mx >= 2.0f * my ? 3 : my >= 2.0f * mx ? 0 : mx >= my ? 2 : 1;
        while ((1 << ty_log2) > ppw) --ty_log2;
    }
    const int ty_mask = (1 << ty_log2) - 1;
    const int tiles_x = (W + (int)(blockDim.x >> 6) * (ppw >> ty_log2) - 1) / ((int)(blockDim.x >> 6) * (ppw >> ty_log2));
    const int blk = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_y = blk / tiles_x, tile_x = blk - tile_y * tiles_x;      // (the grid covers the shape with the most tiles)
    const int pixl = (threadIdx.x & 63) / lg;                // pixel slot inside the wave: rows first
    const int y = (tile_y << ty_log2) + (pixl & ty_mask);
    const int x = (tile_x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (ppw >> ty_log2) + (pixl >> ty_log2);
    if (y >= H || x >= W) return;
    const int dl0 = blockIdx.y * planes_per_block;
    const int dl1 = min(dl0 + planes_per_block, d_count);
    const int sub = threadIdx.x & (lg - 1);
    const int c = sub * 4 * Q;                            // first channel of this lane
    const long long pix = (long long)y * W + x;
    const float xf = (float)x, yf = (float)y;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x2 rr[2 * Q], rq[2 * Q];                           // reference feature and its square, channel pairs
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const float4 r = ld4(ref + (size_t)pix * C + c + 4 * k);
        rr[2 * k] = (f32x2){r.x, r.y}; rr[2 * k + 1] = (f32x2){r.z, r.w};
        rq[2 * k] = rr[2 * k] * rr[2 * k]; rq[2 * k + 1] = rr[2 * k + 1] * rr[2 * k + 1];
    }
    // The 2x2 block a view reads is ONE table entry and ONE address (*r5*): its top-left tap is clamped so that the whole block
    // lies inside the image (rows rb, rb+1 in [0,H), columns cb, cb+1 in [0,W)), the other three taps are the same voffset with
    // the pixel / row / row+pixel stride in the instruction's scalar offset, and where the clamp moved the block (sample point in
    // the one-pixel band around the image) the separable weights move with it -- the tap that fell outside carries weight 0 as
    // before, the products and the order of the four multiply-adds that carry a non-zero weight are unchanged.  Per plane and
    // view that is 1 compare + 1 add + 1 move where four separately clamped offsets cost 2 + 4 + 2, and 5 table dwords for 8.
    int cur[NSRC];                                        // cached block identity (byte offset of its top-left tap)
    float4 t00[NSRC][Q], t01[NSRC][Q], t10[NSRC][Q], t11[NSRC][Q];
    const auto srsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, NSRC * H * W * C * 4, 0x00020000);
    const int pix_bytes = C * 4, row_bytes = W * C * 4, img_bytes = H * W * C * 4, lane_bytes = c * 4;
    const int diag_bytes = row_bytes + pix_bytes;
#pragma unroll
    for (int v = 0; v < NSRC; ++v) {
        cur[v] = -1;
#pragma unroll
        for (int k = 0; k < Q; ++k) t00[v][k] = t01[v][k] = t10[v][k] = t11[v][k] = z4;
    }
    // One IEEE division each, hoisted out of the sweep: the per-plane scalings become multiplies
    // (differs from Q/N, S*S/(N*N) by at most 1 ulp; the single-pass E[x^2]-E[x]^2 form is kept).
    const float n = (float)(NSRC + 1);
    const float inv_n = 1.0f / n, inv_nn = 1.0f / (n * n);

    // Plane-vectorised bookkeeping.  The kernel is VALU-bound and everything except the blend itself
    // is identical for the lg lanes of a pixel, so lane `sub` does ALL the per-view bookkeeping of
    // plane (batch + sub): projective map, floor, the clamped block offset, zero-fill-masked bilinear
    // weights (already multiplied out to one weight per tap).  The 5 numbers per (plane, pixel, view)
    // go through a wave-private LDS table: while sweeping plane p every lane of the pixel reads them
    // back with broadcast ds_read_b128 (one per view for the weights, one per four views for the offsets)
    // and only pays for the loads and the blend.
    extern __shared__ __attribute__((aligned(16))) float4 book_all[];
    constexpr int NOF = (NSRC + 3) / 4, ENT = NSRC + NOF;    // float4 per (plane, pixel): NSRC weight quads, then the offsets
    float4* book = book_all + (size_t)(threadIdx.x >> 6) * (64 * ENT);      // [lg planes][ppw pixels][ENT]

    float* dstp = cost + ((size_t)dl0 * H * W + pix) * C + c;          // this lane's 16 bytes of plane dl0; walks plane by plane
    const size_t plane_elems = (size_t)H * W * C;
    for (int dlb = dl0; dlb < dl1; dlb += lg) {
        {
            const int dmy = d_begin + min(dlb + sub, dl1 - 1);
            float4* mine = book + ((size_t)sub * ppw + pixl) * ENT;
            int ob[NOF * 4];
#pragma unroll
            for (int v = NSRC; v < NOF * 4; ++v) ob[v] = 0;
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float* t = transforms + ((size_t)v * depth_total + dmy) * 8;
                const float4 ta = ld4(t), tb = ld4(t + 4);
                float proj = tb.z * xf + tb.w * yf + 1.0f;
                // v_rcp_f32 (1 ulp, exact for proj = 1) and a multiply where the reference divides: the sample point can differ
                // from the quotient in its last bit, which moves a blend weight by the same amount (oracle parity 1e-5 covers it)
                float inv = __builtin_amdgcn_rcpf(proj);
                float sx = (ta.x * xf + ta.y * yf + ta.z) * inv;
                float sy = (ta.w * xf + tb.x * yf + tb.y) * inv;
                float x0 = floorf(sx), y0 = floorf(sy);
                int ix0 = (int)x0, iy0 = (int)y0;                // v_cvt saturates, NaN -> 0
                // The block that is read: top-left tap clamped to [0, W-2] x [0, H-2] (host-checked: H, W >= 2).  Per-tap zero fill is
                // folded into the separable weights -- a tap is dropped iff its row or its column is outside the image, exactly as
                // reading 0 for it (w * finite = 0) -- and the weights move with the block: unmoved (ix0 = cb) -> the block's columns
                // carry (1 - fx, fx); block moved right (ix0 = -1) -> its first column is the sample's second tap; moved left
                // (ix0 = W-1) -> its second column is the sample's first tap; further out both taps are outside.
                const int cb = min(max(ix0, 0), W - 2), rb = min(max(iy0, 0), H - 2);
                const int dx = ix0 - cb, dy = iy0 - rb;                          // (no overflow: cb, rb >= 0, saturated ix0 / iy0 included)
                const float fx1 = (x0 + 1.0f) - sx, fx0 = sx - x0, fy1 = (y0 + 1.0f) - sy, fy0 = sy - y0;
                const float ax = dx == 0 ? fx1 : (dx == -1 ? fx0 : 0.0f), bx = dx == 0 ? fx0 : (dx == 1 ? fx1 : 0.0f);
                const float ay = dy == 0 ? fy1 : (dy == -1 ? fy0 : 0.0f), by = dy == 0 ? fy0 : (dy == 1 ? fy1 : 0.0f);
                // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): clamped indices and the strides are < 2^24 (host-checked)
                ob[v] = v * img_bytes + (int)__umul24(rb, row_bytes) + (int)__umul24(cb, pix_bytes);
                mine[v] = make_float4(ay * ax, ay * bx, by * ax, by * bx);
            }
#pragma unroll
            for (int k = 0; k < NOF; ++k)
                mine[NSRC + k] = make_float4(__int_as_float(ob[4 * k]), __int_as_float(ob[4 * k + 1]), __int_as_float(ob[4 * k + 2]),
                                             __int_as_float(ob[4 * k + 3]));
        }
        const int np = min(lg, dl1 - dlb);
        // one plane: refill the register tap cache where the block moved, blend, reduce, store
        auto plane = [&](const float4 (&ofq)[NOF], const float4 (&wts)[NSRC]) __attribute__((always_inline)) {
            // phase A: all views' loads are issued before the first one is consumed.
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float4 q = ofq[v >> 2];
                const int o = __float_as_int((v & 3) == 0 ? q.x : (v & 3) == 1 ? q.y : (v & 3) == 2 ? q.z : q.w);
                if (o != cur[v]) {
                    const int vo = o + lane_bytes;
#pragma unroll
                    for (int k = 0; k < Q; ++k) {
                        t00[v][k] = ldbs(srsrc, vo + 16 * k, 0); t01[v][k] = ldbs(srsrc, vo + 16 * k, pix_bytes);
                        t10[v][k] = ldbs(srsrc, vo + 16 * k, row_bytes); t11[v][k] = ldbs(srsrc, vo + 16 * k, diag_bytes);
                    }
                    cur[v] = o;
                }
            }
            // phase B: bilinear blend + running sums, two channels per packed instruction
            f32x2 S[2 * Q], Qs[2 * Q];
#pragma unroll
            for (int k = 0; k < 2 * Q; ++k) { S[k] = rr[k]; Qs[k] = rq[k]; }
#pragma unroll
            for (int v = 0; v < NSRC; ++v) {
                const float w00 = wts[v].x, w01 = wts[v].y, w10 = wts[v].z, w11 = wts[v].w;
#pragma unroll
                for (int k = 0; k < Q; ++k) {
                    f32x2 a0 = (f32x2){t00[v][k].x, t00[v][k].y}, a1 = (f32x2){t00[v][k].z, t00[v][k].w};
                    f32x2 b0 = (f32x2){t01[v][k].x, t01[v][k].y}, b1 = (f32x2){t01[v][k].z, t01[v][k].w};
                    f32x2 c0 = (f32x2){t10[v][k].x, t10[v][k].y}, c1 = (f32x2){t10[v][k].z, t10[v][k].w};
                    f32x2 e0 = (f32x2){t11[v][k].x, t11[v][k].y}, e1 = (f32x2){t11[v][k].z, t11[v][k].w};
                    f32x2 w0 = w00 * a0 + w01 * b0 + w10 * c0 + w11 * e0;
                    f32x2 w1 = w00 * a1 + w01 * b1 + w10 * c1 + w11 * e1;
                    S[2 * k] += w0; S[2 * k + 1] += w1;
                    Qs[2 * k] += w0 * w0; Qs[2 * k + 1] += w1 * w1;
                }
            }
            float* dst = dstp;
            dstp += plane_elems;
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                f32x2 o0, o1;
                // the negated forms are the same fused multiply-adds with the signs of the operands exchanged: same bits, sign flipped
                if (variant == 0) {
                    const f32x2 s0 = (S[2 * k] * S[2 * k]) * inv_nn, s1 = (S[2 * k + 1] * S[2 * k + 1]) * inv_nn;
                    if (negate) { o0 = s0 - Qs[2 * k] * inv_n; o1 = s1 - Qs[2 * k + 1] * inv_n; }
                    else { o0 = Qs[2 * k] * inv_n - s0; o1 = Qs[2 * k + 1] * inv_n - s1; }
                } else {
                    const f32x2 m0 = S[2 * k] * inv_n, m1 = S[2 * k + 1] * inv_n;
                    const f32x2 s0 = m0 * m0, s1 = m1 * m1;
                    if (negate) { o0 = s0 - Qs[2 * k] * inv_n; o1 = s1 - Qs[2 * k + 1] * inv_n; }
                    else { o0 = Qs[2 * k] * inv_n - s0; o1 = Qs[2 * k + 1] * inv_n - s1; }
                }
                // non-temporal: the 503 MB volume streams out and must not push the 13 MB of feature maps, which every plane
                // re-reads, out of the L2 (199.7 -> 185.6 us inside a depth map, 246 -> 228 us alone)
                __builtin_nontemporal_store((f32x4_nt){o0[0], o0[1], o1[0], o1[1]}, reinterpret_cast<f32x4_nt*>(dst + 4 * k));
            }
        };
        auto fetch = [&](int p, float4 (&ofq)[NOF], float4 (&wts)[NSRC]) __attribute__((always_inline)) {
            const float4* bk = book + ((size_t)min(p, lg - 1) * ppw + pixl) * ENT;
#pragma unroll
            for (int v = 0; v < NSRC; ++v) wts[v] = bk[v];
#pragma unroll
            for (int k = 0; k < NOF; ++k) ofq[k] = bk[NSRC + k];
        };
        // (reading plane p+1's entries during plane p costs a second register set and an occupancy
        // step: measured slower)
        for (int p = 0; p < np; ++p) {
            float4 ofq[NOF], wts[NSRC];
            fetch(p, ofq, wts);
            __builtin_amdgcn_sched_barrier(0);
            plane(ofq, wts);
        }
    }
}

template <int NSRC>
void launch_sweep(const float* ref, const float* src, const float* transforms, int depth_total,
                  int d_begin, int d_count, int H, int W, int C, int variant, int negate,
                  float* cost, hipStream_t st, int threads = 256) {
    // Planes per workgroup: whole rounds of the chip's 1024 workgroup slots (four 4-wave workgroups per CU at 117 VGPRs) -- a
    // workgroup costs its planes + ~1.5 planes of start-up (reference features, vote, first table block).  At the metric workload
    // 16 / 24 / 32 / 48 planes are 7.5 / 5.0 / 3.75 / 2.5 rounds: 967 / 971 / 970 / 970 depth maps/s (8, 12: 955, 952; 96: 958).
    int ppb = d_count;
    if (d_count > 16) {
        const int lg0 = C / 4, ppw0 = 64 / lg0, nw0 = threads / 64;
        const long long wgs_per_chunk = (long long)mvs_cdiv(W, nw0 * ppw0) * H;      // about the same for every wave tile shape
        double best = 1e30;
        for (int cand = 16; cand <= 48; cand += 8) {
            const long long wgs = wgs_per_chunk * mvs_cdiv(d_count, cand);
            const double cost = (double)((wgs + 1023) / 1024) * (cand + 1.5);
            if (cost < best) { best = cost; ppb = cand; }
        }
    }
    // Q = 2 (8 channels per lane) halves the per-lane bookkeeping per channel but needs 236 VGPRs
    // (2 waves/SIMD instead of 3): measured 0.258 ms vs 0.243 ms at the metric config, so only Q = 1 is instantiated.
    const int lg = C / 4, ppw = 64 / lg, nw = threads / 64;
    // wave tile: rows x columns of pixels, voted in the kernel from the transforms (ty_log2 = -1); the test hook MVS_HOOK_CV_TILE_ROWS_LOG2 = 0..3 forces
    // a shape (tests, measurements).  The grid covers the shape that needs the most workgroups; the others return early.
    int ty_log2 = mvs_hook(MVS_HOOK_CV_TILE_ROWS_LOG2);
    if (ty_log2 > 3) ty_log2 = 3;
    while (ty_log2 >= 0 && (1 << ty_log2) > ppw) --ty_log2;
    int blocks = 0;
    for (int t = 0; t <= 3 && (1 << t) <= ppw; ++t) {
        if (ty_log2 >= 0 && t != ty_log2) continue;
        const int b = mvs_cdiv(W, nw * (ppw >> t)) * mvs_cdiv(H, 1 << t);
        if (b > blocks) blocks = b;
    }
    dim3 grid(blocks, mvs_cdiv(d_count, ppb));
    const size_t smem = (size_t)(threads / 64) * 64 * (NSRC + (NSRC + 3) / 4) * sizeof(float4);     // bookkeeping table: 1 KB per wave and view + 1 KB per four views
#define MVS_SWEEP_VN(VN) case VN: cost_volume_sweep_kernel<NSRC, 1, VN><<<grid, threads, smem, st>>>(ref, src, transforms, depth_total, d_begin, d_count, ppb, H, W, C, ty_log2, cost); break;
    switch ((variant == 0 ? 0 : 2) + (negate ? 1 : 0)) {
        MVS_SWEEP_VN(0) MVS_SWEEP_VN(1) MVS_SWEEP_VN(2) MVS_SWEEP_VN(3)
    }
#undef MVS_SWEEP_VN
}

template <int BORDER>
__global__ void __launch_bounds__(256)
warp_kernel(const float* __restrict__ img, const float* __restrict__ t, int H, int W, int C,
            float* __restrict__ out) {
    const int cq = C >> 2;
    const long long total = (long long)H * W * cq;
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int c = (int)(idx % cq) * 4;
    long long pix = idx / cq;
    int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    float4 w = warp_sample<BORDER>(img, t, (float)x, (float)y, H, W, C, c);
    *reinterpret_cast<float4*>(out + (size_t)pix * C + c) = w;
}

}  // namespace

// `threads` per workgroup of the depth sweep (64-thread multiples up to 256).  The recurrent sweep's producer stream asks for 128:
// such a workgroup needs 16 KB of LDS and fits on a CU beside a 141-144 KB workgroup of the fused ConvGRU launches (gru.hip).
int mvs_cost_volume_threads_f32(const float* ref, const float* src, const float* transforms,
                                int view_num, int depth_total, int d_begin, int d_count,
                                int H, int W, int C, int variant, int negate, int border,
                                float* cost, int threads, void* stream) {
    MVS_CHECK_ARG(ref && src && transforms && cost);
    MVS_CHECK_ARG(view_num >= 2 && depth_total >= 1 && d_begin >= 0 && d_count >= 1 &&
                  d_begin + d_count <= depth_total && H > 0 && W > 0 && C > 0);
    // whole waves only, within __launch_bounds__(256): the LDS bookkeeping table is sized per wave
    MVS_CHECK_ARG(threads == 64 || threads == 128 || threads == 192 || threads == 256);
    if (C % 4 != 0) return MVS_E_SHAPE;
    long long total = (long long)H * W * (C / 4);
    dim3 grid(mvs_cdiv(total, 256), d_count);
    const int cq_ = C / 4;
    const bool cq_pow2 = cq_ <= 16 && (cq_ & (cq_ - 1)) == 0;     // lanes of a pixel stay inside one wave
    // 32-bit byte offsets into the source maps: all views together must stay below 2 GiB
    const bool off32 = (long long)(view_num - 1) * H * W * C * 4 < (1LL << 31);
    // (Two other forms of this kernel were built, are exact, and measured no faster: the LDS-staged sweep and the MFMA-blend
    // sweeps.  They live under csrc/lab/ with their own entry points and tests -- DESIGN 4.1 -- not in this library.)
    const bool u24 = H < (1 << 24) && W < (1 << 24) && (long long)W * C * 4 < (1 << 24);     // the sweep's 24-bit offset multiplies
    if (border == 0 && view_num <= 8 && cq_pow2 && off32 && u24 && H >= 2 && W >= 2) {      // depth sweep with register tap reuse
        hipStream_t st = mvs_stream(stream);
#define MVS_SWEEP(NS) case NS: launch_sweep<NS>(ref, src, transforms, depth_total, d_begin, d_count, H, W, C, variant, negate, cost, st, threads); break;
        switch (view_num - 1) {
            MVS_SWEEP(1) MVS_SWEEP(2) MVS_SWEEP(3) MVS_SWEEP(4) MVS_SWEEP(5) MVS_SWEEP(6) MVS_SWEEP(7)
        }
#undef MVS_SWEEP
        MVS_LAUNCH_RET();
    }
    if (border == 0)
        cost_volume_kernel<0><<<grid, 256, 0, mvs_stream(stream)>>>(
            ref, src, transforms, view_num - 1, depth_total, d_begin, H, W, C, variant, negate, cost);
    else
        cost_volume_kernel<1><<<grid, 256, 0, mvs_stream(stream)>>>(
            ref, src, transforms, view_num - 1, depth_total, d_begin, H, W, C, variant, negate, cost);
    MVS_LAUNCH_RET();
}

extern "C" int mvs_cost_volume_f32(const float* ref, const float* src, const float* transforms,
                                   int view_num, int depth_total, int d_begin, int d_count,
                                   int H, int W, int C, int variant, int negate, int border,
                                   float* cost, void* stream) {
    return mvs_cost_volume_threads_f32(ref, src, transforms, view_num, depth_total, d_begin, d_count, H, W, C, variant, negate, border,
                                       cost, 256, stream);
}

extern "C" int mvs_warp_f32(const float* image, const float* transform8, int H, int W, int C,
                            int border, float* out, void* stream) {
    MVS_CHECK_ARG(image && transform8 && out && H > 0 && W > 0 && C > 0);
    if (C % 4 != 0) return MVS_E_SHAPE;
    long long total = (long long)H * W * (C / 4);
    if (border == 0)
        warp_kernel<0><<<mvs_cdiv(total, 256), 256, 0, mvs_stream(stream)>>>(image, transform8, H, W, C, out);
    else
        warp_kernel<1><<<mvs_cdiv(total, 256), 256, 0, mvs_stream(stream)>>>(image, transform8, H, W, C, out);
    MVS_LAUNCH_RET();
}
