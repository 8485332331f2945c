/* mvsnet_hip.h -- C ABI of libmvsnet_hip.so (MI355X / gfx950 plane-sweep depth inference).
 *
 * The reference (ubiquity6/MVSNet) has NO plugin / operator / FFI interface: its seam is a set of
 * Python functions building one TensorFlow graph.  This header is therefore the new
 * drop-in boundary: one entry point per reference function on the hot path, cited below.  The
 * Python host side (mvsnet_amd/model.py, same names and argument meaning as mvsnet/model.py) binds
 * these through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller unless marked [host];
 *   - tensors are float32, channel-last, contiguous: images (H,W,C), volumes (D,H,W,C);
 *   - functions enqueue kernels on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); they never allocate, never free, never synchronise, never create streams or events -- so a
 *     whole pass captures into a hipGraph.  The exceptions are named where they are declared and are all set-up /
 *     tear-down / diagnostic calls, never compute calls: mvs_gru_prepare and mvs_gru_release (create / destroy the
 *     recurrent sweep's side streams and events, synchronise), and the profiling read-outs mvs_profile_dominant_ms /
 *     mvs_profile_layers_ms (wait for their timing events);
 *   - return value: 0 on success, a positive hipError_t on a HIP failure, a negative
 *     MVS_E_* code on an argument error.  Nothing throws.
 */
#ifndef MVSNET_HIP_H_
#define MVSNET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_ABI_VERSION 1

#define MVS_E_BADARG   (-1)   /* null pointer / non-positive size                         */
#define MVS_E_SHAPE    (-2)   /* shape not supported by this build (see function comment) */
#define MVS_E_WORKSPACE (-3)  /* workspace too small                                      */
#define MVS_E_NOT_PREPARED (-4) /* mvs_gru_prepare on a stream under hipGraph capture (it synchronises);
                                   mvs_gru_stream_layout on a stream without a set */
#define MVS_E_NO_SLOT  (-5)   /* mvs_gru_prepare: all 16 stream sets of the process are in use (mvs_gru_release frees one).
                                   Not fatal for the sweep: without a set it runs on the caller's stream alone */

/* Regulariser implementation selector (mvs_set_conv_impl): the MFMA path is the product;
 * the scalar path is a slow, shape-generic HIP cross-check (never a CPU fallback). */
#define MVS_CONV_IMPL_AUTO   0
#define MVS_CONV_IMPL_SCALAR 1
#define MVS_CONV_IMPL_MFMA   2
/* opt-in: stride-1 MFMA layers evaluate fp32 products as three bf16 MFMAs (x = hi + lo split,
 * fp32 accumulate, ~1.5e-5 relative error per product); only through mvs_regnet_us0_prepared_f32 */
#define MVS_CONV_IMPL_BF16X3 3

int mvs_abi_version(void);
/* Static description of a return code (never NULL). */
const char* mvs_error_string(int code);
int mvs_set_conv_impl(int impl);
int mvs_get_conv_impl(void);

/* Test / measurement hooks.  The library reads NOTHING from the environment: every switch that lets a test reach a schedule the
 * launchers would not pick by themselves is one of these process-wide integers (atomic; read at launch time, so set them while
 * no call of the library is in flight).  Results never depend on them except where stated; 0 (or the stated default) = the
 * product behaviour.  mvs_set_test_hook returns MVS_E_BADARG for an unknown id or a value outside the hook's range and leaves
 * the hook unchanged; mvs_get_test_hook returns the current value (or MVS_E_BADARG). */
#define MVS_HOOK_CV_TILE_ROWS_LOG2    0  /* warp + variance: force the wave tile to 2^v rows (v = 0..3); -1 (default) = voted from the transforms.  Same bits */
#define MVS_HOOK_CONV_NO_SPAN         1  /* 1: 3dconv0_1 + 1_0 launch walks whole depth chunks instead of SPAN ranges.  Same bits */
#define MVS_HOOK_CONV_NO_FUSE2        2  /* 1: 3dconv1_1 and 2_0 run as two launches instead of the fused one.  Same bits */
#define MVS_HOOK_S2_PLANES            3  /* >= 1: output planes per workgroup of the stride-2 plane-march kernel; 0 = the launcher's choice.  Same bits */
#define MVS_HOOK_GRU_ONE_STREAM       4  /* 1: the wavefront formulations of the recurrent sweep stay on the caller's stream.  Same bits */
#define MVS_HOOK_GRU_PRODUCER_THREADS 5  /* threads per workgroup of the wavefront's producer launches: 64, 128 (default), 192 or 256.  Same bits */
#define MVS_HOOK_UNET_PERSISTENT       6  /* 1 (default): tower layers with a persistent instance (csrc/unet2d_p.hip) take it; 0: the one-tile-per-workgroup kernels.  Sums differ in the last float64 bits */
#define MVS_HOOK_UNET_GRID            7  /* >= 1: persistent workgroups per launch (per cout group; capped at the tile count) -- tests reach multi-tile ranges at small sizes, measurements vary the residency; 0 (default) = the launcher's choice */
#define MVS_HOOK_FUSE2_PLANES          8  /* even, >= 2: planes per workgroup of the fused 3dconv1_1 + 2_0 launch (measurement); 0 (default) = the launcher's choice.  BatchNorm sums arrive in another order: last bits */
#define MVS_HOOK_REGNET_SIDE_BRANCH    9  /* 1: RegNetUS0's 3dconv1_1 on a side stream of the caller's stream set (mvs_gru_prepare) beside 3dconv2_0 and the low-resolution chain (measurement; needs a set, no capture); 0 (default): fused with 3dconv2_0 in line */
#define MVS_HOOK_COUNT                10
int mvs_set_test_hook(int id, int value);
int mvs_get_test_hook(int id);

/* ---------------------------------------------------------------------------------------------
 * R1/R1' + R2 prep: plane-induced homographies and their tf.contrib.image.transform 8-vectors.
 * Replaces get_homographies (mvsnet/homography_warping.py:10-58), get_homographies_inv_depth
 * (:60-106) and the coefficient algebra of tf_transform_homography (:216-250).
 *   cams          (view_num,2,4,4)  reference layout, view 0 = reference view
 *   inverse_depth 0: depth_d = depth_start + d*depth_interval          (depth_end ignored)
 *                 1: depth_d = 1/linspace(1/depth_start, 1/depth_end, D) (depth_interval ignored)
 *   homographies  (view_num-1, depth_num, 3, 3)   may be NULL
 *   transforms    (view_num-1, depth_num, 8)      a0 a1 a2 b0 b1 b2 c0 c1, normalised by c2'
 */
int mvs_homography_transforms_f32(const float* cams, int view_num, int depth_num,
                                  float depth_start, float depth_interval, float depth_end,
                                  int inverse_depth, float* homographies, float* transforms,
                                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * R2 + R3: fused projective bilinear warp (zero fill per tap) + cross-view variance.
 * Replaces tf_transform_homography (mvsnet/homography_warping.py:251-252) and the cost loop of
 * inference_mem (mvsnet/model.py:422-463) / inference (:315-334) / the GRU body (:680-693).
 *   ref        (H,W,C)            reference-view feature map
 *   src        (view_num-1,H,W,C) source-view feature maps
 *   transforms (view_num-1,depth_total,8)
 *   d_begin,d_count   planes [d_begin, d_begin+d_count) are produced (d_count = depth_total for the
 *                     3D-CNN path, 1 per step for the recurrent path)
 *   variant    0: cost = Q/N - S*S/(N*N)   (inference_mem, model.py:458-461)
 *              1: cost = Q/N - (S/N)^2     (inference / GRU, model.py:330-332)
 *   negate     1: writes -cost (the GRU consumes -cost, model.py:698)
 *   border     0: zero fill per tap (reference behaviour)   1: clamp taps to the border
 *              (the reference's dead homography_warping path, homography_warping.py:146-149)
 *   cost       (d_count,H,W,C)
 * C must be a multiple of 4.
 */
int mvs_cost_volume_f32(const float* ref, const float* src, const float* transforms,
                        int view_num, int depth_total, int d_begin, int d_count,
                        int H, int W, int C, int variant, int negate, int border,
                        float* cost, void* stream);

/* Stand-alone warp of one feature map by one transform (tf_transform_homography,
 * mvsnet/homography_warping.py:211-253); used by the parity tests.  image/out (H,W,C). */
int mvs_warp_f32(const float* image, const float* transform8, int H, int W, int C, int border,
                 float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * R4: 3x3x3 convolution / transposed convolution with TensorFlow SAME padding, no bias, and the
 * batch-statistics BatchNorm of the PRODUCER fused into the consumer's load:
 *      in(v,c) = act(x(v,c)*x_scale[c] + x_shift[c]) [ + act(x2(v,c)*x2_scale[c] + x2_shift[c]) ]
 * with act = ReLU when a scale pointer is given, identity (scale=1, shift=0) when it is NULL.
 * Replaces Network.conv / conv_bn / deconv / deconv_bn / batch_normalization / add
 * (mvsnet/cnn_wrapper/network.py:171-215,278-348,457-459,492-509).
 *   x, x2     (D,H,W,Cin)        x2 may be NULL (no skip connection)
 *   w         conv:   (3,3,3,Cin,Cout)  TensorFlow conv3d kernel layout
 *             deconv: (3,3,3,Cout,Cin)  TensorFlow conv3d_transpose kernel layout
 *   stride    conv: 1 or 2 (out = ceil(n/stride)); deconv: 2 (out = 2n)
 *   y         raw (pre-BN) output, (Do,Ho,Wo,Cout)
 *   stats     NULL, or (2,Cout) float64 accumulators [sum, sum of squares] over all output voxels;
 *             must be zeroed by the caller before the launch (mvs_zero_f64)
 */
int mvs_conv3d_f32(const float* x, const float* x_scale, const float* x_shift,
                   const float* x2, const float* x2_scale, const float* x2_shift,
                   const float* w, int D, int H, int W, int Cin, int Cout, int stride,
                   float* y, double* stats, void* stream);
int mvs_deconv3d_f32(const float* x, const float* x_scale, const float* x_shift,
                     const float* x2, const float* x2_scale, const float* x2_shift,
                     const float* w, int D, int H, int W, int Cin, int Cout,
                     float* y, double* stats, void* stream);

/* The two consumers of the cost volume in ONE pass over it: y1 = conv3d(x, w1, stride 1) and
 * y2 = conv3d(x, w2, stride 2), i.e. 3dconv0_1 and 3dconv1_0 of RegNetUS0
 * (mvsnet/cnn_wrapper/mvsnetworks.py:130-134), each with the semantics of mvs_conv3d_f32 (raw input,
 * no producer BatchNorm).  Only the shape of that layer pair is implemented: Cin = 32, Cout1 = 8,
 * Cout2 = 16, even D, H, W; anything else returns MVS_E_SHAPE (call mvs_conv3d_f32 twice instead).
 *   x (D,H,W,32)   w1 (3,3,3,32,8)   w2 (3,3,3,32,16)   y1 (D,H,W,8)   y2 (D/2,H/2,W/2,16)
 *   stats1 / stats2: NULL or zeroed (2,8) / (2,16) float64 accumulators as in mvs_conv3d_f32 */
int mvs_conv3d_pair_f32(const float* x, const float* w1, const float* w2, int D, int H, int W,
                        int Cin, int Cout1, int Cout2, float* y1, double* stats1,
                        float* y2, double* stats2, void* stream);

/* BatchNorm (training-mode statistics, biased variance; network.py:496-506) folded to an affine:
 *   mean = sum/count; var = sumsq/count - mean^2; scale = gamma/sqrt(var+eps); shift = beta-mean*scale
 * stats (2,C) float64 as produced above; scale/shift (C). */
int mvs_bn_finalize_f32(const double* stats, int C, double count, const float* gamma,
                        const float* beta, float eps, float* scale, float* shift, void* stream);
int mvs_zero_f64(double* p, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * R5: the whole RegNetUS0 3D U-Net (mvsnet/cnn_wrapper/mvsnetworks.py:122-158).
 *   cost      (D,H,W,Cin) with D,H,W divisible by 8
 *   weights   [host] array of 11 device pointers in the order
 *             1_0, 2_0, 3_0, 0_1, 1_1, 2_1, 3_1, 4_0, 5_0, 6_0, 6_2   (TensorFlow layouts)
 *   gammas, betas  [host] arrays of 10 device pointers (same order, no entry for 6_2)
 *   base      base_filter (8 for network_mode 'normal'); Cin of the volume = cin
 *   workspace device scratch of at least mvs_regnet_workspace_bytes() bytes
 *   reg       (D,H,W) filtered cost volume (the squeezed 3dconv6_2 output)
 */
size_t mvs_regnet_workspace_bytes(int D, int H, int W, int cin, int base);
/* Optional one-off weight pre-layout: re-orders the 10 MFMA layers' kernels into the order the
 * kernels keep them in LDS, so that a workgroup uploads its weights with coalesced 16-byte lanes.
 * `prepared` holds mvs_regnet_prepared_floats() floats; pass it to mvs_regnet_us0_prepared_f32
 * together with the original weights (still used by layers outside the MFMA tiling). */
size_t mvs_regnet_prepared_floats(int cin, int base);
int mvs_regnet_prepare_f32(const float* const* weights, int cin, int base, float* prepared,
                           void* stream);
int mvs_regnet_us0_prepared_f32(const float* cost, int D, int H, int W, int cin, int base,
                                const float* const* weights, const float* prepared,
                                const float* const* gammas, const float* const* betas, float eps,
                                void* workspace, size_t workspace_bytes, float* reg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * the 2D convolutions of the UNetDS2GN feature extractor (mvsnet/cnn_wrapper/
 * mvsnetworks.py:53-115; Network.conv_gn / deconv_gn / conv, network.py:171-276,350-409), batched
 * over the V views of a cluster (GroupNorm is per view).  As in the 3D stack a layer stores its RAW
 * output plus float64 group sums, and the consumer applies the producer's GroupNorm (eps 1e-5,
 * 8 channels per group, biased variance) while loading:
 *      in(v,p,c) = act(x(v,p,c) * s(v,c) + t(v,c)),  s, t from stats (V, C/8, S, 2) [sum, sumsq], gamma, beta
 * where S = mvs_gn_stat_slots() partial accumulators per (view, group) that consumers add up (they keep
 * the float64 atomics of ~10^4 workgroups from serialising on one address).
 * (stats == NULL: identity, e.g. the image; relu != 0 for conv_gn producers, 0 for deconv_gn).
 *   mvs_conv2d_gn_f32     k x k (3 or 5) SAME convolution, stride 1 or 2, no bias, of the channel
 *                         concatenation of one or two sources; weights pre-laid-out by
 *                         mvs_conv2d_prepare_f32 (from the TensorFlow (k,k,Cin,Cout) layout, same
 *                         cin1/cin2/cout); channel counts multiples of 4 (image: pad 3 -> 4), Cout of 8
 *   mvs_deconv2d_gn_f32   3x3 stride-2 SAME transposed convolution (out = 2H x 2W): MFMA path when
 *                         `prepared` (mvs_deconv2d_prepare_f32, Cin a multiple of 16) is given, else a
 *                         VALU gather on `w` in the TensorFlow (3,3,Cout,Cin) layout
 *   stats_out             NULL or zeroed (V, Cout/8, S, 2) float64 accumulators of the raw output
 */
int mvs_gn_stat_slots(void);
size_t mvs_conv2d_prepared_floats(int ks, int cin1, int cin2, int cout);
int mvs_conv2d_prepare_f32(const float* w, int ks, int cin1, int cin2, int cout, float* prepared, void* stream);
/* Prepared weights of the stride-1 convolution that computes a layer's INPUT gradient, from the layer's forward kernel
 * w (k,k,cin_fwd,cout_fwd): consumer and buffer size as mvs_conv2d_prepare_f32(ks, cout_fwd, 0, cin_fwd). */
int mvs_conv2d_prepare_dgrad_f32(const float* w, int ks, int cin_fwd, int cout_fwd, float* prepared, void* stream);
int mvs_conv2d_gn_f32(const float* x1, const double* stats1, const float* gamma1, const float* beta1, int c1, int relu1,
                      const float* x2, const double* stats2, const float* gamma2, const float* beta2, int c2, int relu2,
                      const float* prepared, int V, int H, int W, int cout, int ks, int stride,
                      float* y, double* stats_out, void* stream);
/* n weight preparations in one launch (the training towers: every layer's forward and input-gradient kernel once per step).
 * Job i: kind 0 = mvs_conv2d_prepare_f32(w, ks, c1, c2, cout) with cin_src channels actually in `w` (the image layer: c1 = 4,
 * cin_src = 3, the fourth laid out as zeros; otherwise c1 + c2); kind 1 = mvs_conv2d_prepare_dgrad_f32(w, ks, cin_fwd = c1,
 * cout_fwd = cout); kind 2 = mvs_deconv2d_prepare_f32(w, cin = c1, cout); into prepared[i], sized as for the single calls.
 * All array arguments are host arrays of n entries. */
int mvs_unet_prepare_many_f32(int n, const int* kind, const float* const* w, const int* ks, const int* c1, const int* c2,
                              const int* cin_src, const int* cout, float* const* prepared, void* stream);
size_t mvs_deconv2d_prepared_floats(int cin, int cout);
int mvs_deconv2d_prepare_f32(const float* w, int cin, int cout, float* prepared, void* stream);
int mvs_deconv2d_gn_f32(const float* x, const double* stats, const float* gamma, const float* beta, int cin, int relu,
                        const float* w, const float* prepared, int V, int H, int W, int cout, float* y,
                        double* stats_out, void* stream);

/* The towers' input side: per-image, per-channel standardisation of decoded uint8 images on the device, replacing the host
 * call mvs_data_generation/utils.py:33-38 center_image (applied per image by cluster_generator before the graph sees it):
 *   images     (V, H, W, 3) uint8, 4-byte aligned, H*W a multiple of 4
 *   out4       (V, H, W, 4) float32, 16-byte aligned: (x - mean_c) / (sqrt(var_c) + 1e-8) per image and channel, channel 3 = 0
 *              (the layout mvs_conv2d_gn_f32 reads the image in: 3 channels padded to 4)
 *   workspace  mvs_center_images_workspace_bytes(V) bytes (zeroed by the call): exact uint64 sums and sums of squares
 * mean and biased variance come from exact integer totals in float64; the output expression is evaluated in float32 as the
 * reference's.  MVS_E_SHAPE for sizes / alignments outside the above. */
size_t mvs_center_images_workspace_bytes(int V);
int mvs_center_images_u8_f32(const uint8_t* images, int V, int H, int W, float* out4, void* workspace, void* stream);

/* Live timing of the dominant kernel for bench.py's `roofline` object: while enabled, every
 * mvs_regnet_us0_*_f32 call brackets its first launch (the fused 3dconv0_1 + 3dconv1_0 pass over the
 * cost volume, conv3d_c8_kernel) with HIP events on the caller's stream (up to 64 calls).
 * mvs_profile_dominant_ms waits for those events, returns their average duration in milliseconds and
 * the number of samples, and clears them.  Do not enable during hipGraph capture. */
int mvs_profile_dominant(int enable);
int mvs_profile_dominant_ms(double* avg_ms, int* count);
/* Per-layer timing inside a depth map (bench.py's `roofline_kernels` rows of the low-resolution layers): while
 * enabled, every mvs_regnet_us0_*_f32 call brackets EACH of its layer launches with HIP events on the stream that
 * launch goes to (the branch layers' side stream included), up to 32 calls.  mvs_profile_layers_ms waits for the
 * events and returns, for the 11 layers in weight order (3dconv1_0 2_0 3_0 0_1 1_1 2_1 3_1 4_0 5_0 6_0 6_2), the
 * average duration in milliseconds (the fused 3dconv0_1 + 3dconv1_0 pass reports its time under 3dconv0_1 and 0 under
 * 3dconv1_0) and the number of calls sampled.  Every event costs a few microseconds of device idle time: use it
 * outside timed regions; not during hipGraph capture. */
/* The same for the three stages of mvs_depth_from_features_f32 -- [0] homographies + warp + variance, [1] the RegNetUS0 stack
 * (its 11 launches), [2] softmax / soft-argmin / probability -- so that bench.py's stage rows describe the timed path itself (one
 * library call per depth map) and sum to its duration, plus the ~1-2 us of device idle time each event record costs. */
int mvs_profile_stages(int enable);
int mvs_profile_stages_ms(double* avg_ms3, int* count);
int mvs_profile_layers(int enable);
int mvs_profile_layers_ms(double* avg_ms11, int* count);
/* Launch plan of the 1/8-resolution chain.  With prepared weights and the metric's channel widths 3dconv2_1 (read only by
 * the decoder's 3dconv5_0, mvsnetworks.py:139,148-150) has no launch of its own: its blocks ride as filler workgroups behind
 * the blocks of 3dconv3_0, 3dconv3_1 and 3dconv4_0 (240 workgroups each, half of their time fixed cost with the matrix pipe
 * idle).  [host] permille3 receives the share of 3dconv2_1's blocks in each of those three launches, in 1/1000
 * (compile-time constants); mvs_profile_layers_ms then reports 0 under 3dconv2_1.  Without prepared weights, or for other
 * channel widths / volumes of 2 GB and more, the four layers are launched apart whatever this call says. */
int mvs_regnet_filler_shares(int* permille3);
/* The whole 3D-CNN path after the feature towers in ONE call (inference_mem, model.py:408-502): plane homographies ->
 * fused warp + variance cost volume -> RegNetUS0 -> softmax / soft-argmin / probability map; 14 launches, none of them for
 * bookkeeping (the BatchNorm sums are cleared by the homography launch).
 *   features (view_num,H,W,C): the reference view first; cams (view_num,2,4,4); depth samples as in
 *   mvs_homography_transforms_f32; variant as in mvs_cost_volume_f32;
 *   caller-owned scratch: transforms (view_num-1,D,8), cost (D,H,W,C), workspace (mvs_regnet_workspace_bytes), reg (D,H,W);
 *   outputs depth, prob (H,W).  prepared: mvs_regnet_prepare_f32's buffer or NULL. */
int mvs_depth_from_features_f32(const float* features, const float* cams, int view_num, int depth_num,
                                int H, int W, int C, int base, float depth_start, float depth_interval,
                                float depth_end, int inverse_depth, int variant,
                                const float* const* weights, const float* prepared,
                                const float* const* gammas, const float* const* betas, float eps,
                                float* transforms, float* cost, void* workspace, size_t workspace_bytes,
                                float* reg, float* depth, float* prob, void* stream);
/* RegNetUS0 on a batch (FLAGS.batch_size > 1, model.py:466-469): the reference's BatchNorm layers normalise with
 * the statistics of the whole batch (B,D,H,W) (network.py:496-506), so samples are coupled through every layer's sums.
 *   cost (B,D,H,W,cin) contiguous; reg (B,D,H,W); workspace >= B * mvs_regnet_workspace_bytes(D,H,W,cin,base);
 *   prepared: the buffer of mvs_regnet_prepare_f32 or NULL. */
int mvs_regnet_us0_batch_f32(const float* cost, int batch, int D, int H, int W, int cin, int base,
                             const float* const* weights, const float* prepared,
                             const float* const* gammas, const float* const* betas, float eps,
                             void* workspace, size_t workspace_bytes, float* reg, void* stream);
int mvs_regnet_us0_f32(const float* cost, int D, int H, int W, int cin, int base,
                       const float* const* weights, const float* const* gammas,
                       const float* const* betas, float eps, void* workspace,
                       size_t workspace_bytes, float* reg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * R6 + R7: softmax(-reg) over depth, soft-argmin depth, 4-bucket probability map.
 * Replaces mvsnet/model.py:471-498 and get_probability_map_slice (:45-144).
 *   reg (D,H,W); depth, prob (H,W); depth samples as in mvs_homography_transforms_f32 with
 *   depth_end = depth_start + (D-1)*depth_interval.
 */
int mvs_softargmin_prob_f32(const float* reg, int D, int H, int W, float depth_start,
                            float depth_interval, int inverse_depth, float* depth, float* prob,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * R8: 3x3 SAME 2D convolution over the channel concatenation [xa | xb] with bias, used by the
 * ConvGRU cell (mvsnet/convgru.py:89-93,107-111) and prob_conv (mvsnet/model.py:701-702).
 *   xa (H,W,Ca), xb (H,W,Cb) (xb may be NULL with Cb = 0); w (3,3,Ca+Cb,Cout); bias (Cout)
 *   y  (H,W,Cout)
 *   stats  NULL or (groups,2) float64 [sum,sumsq] over (H,W, Cout/groups channels) per group,
 *          zeroed by the caller: the LayerNorm moments of convgru.py:30-31 (groups = 2 for the
 *          gate convolution: reset | update; 1 for the output convolution).
 */
int mvs_conv2d_cat_f32(const float* xa, int Ca, const float* xb, int Cb, const float* w,
                       const float* bias, int H, int W, int Cout, float* y, double* stats,
                       int groups, void* stream);

/* ConvGRU element-wise stages (mvsnet/convgru.py:97-102,114-120); LayerNorm = tf.contrib.layers
 * layer_norm: moments over (H,W,F), eps 1e-12, per-channel gamma/beta.
 *   gates: g (H,W,2F) raw gate conv; stats (2,2) f64 -> r*h (H,W,F) and u (H,W,F)
 *   blend: c (H,W,F) raw output conv; stats (1,2) f64; h updated in place:
 *          h = u*h + (1-u)*tanh(LN(c))
 */
int mvs_gru_gates_f32(const float* g, const double* stats, const float* reset_gamma,
                      const float* reset_beta, const float* update_gamma, const float* update_beta,
                      const float* h, int H, int W, int F, float* rh, float* u, void* stream);
int mvs_gru_blend_f32(const float* c, const double* stats, const float* out_gamma,
                      const float* out_beta, const float* u, int H, int W, int F, float* h,
                      void* stream);

/* R9: winner-take-all update (mvsnet/model.py:703-731): prob = exp(reg); strict '<' keeps the first
 * maximum.  All maps (H,W).  mvs_wta_finish: prob_out = max_prob / (exp_sum + 1e-7) (:749-751). */
int mvs_wta_update_f32(const float* reg, float depth_value, int H, int W, float* max_prob,
                       float* depth_image, float* exp_sum, void* stream);
int mvs_wta_finish_f32(const float* max_prob, const float* exp_sum, int H, int W, float* prob_out,
                       void* stream);

/* R9 composed: the whole recurrent sweep of inference_winner_take_all (mvsnet/model.py:601-751)
 * after the feature towers.
 *   params [host] array of 32 device pointers: for cell i in 1..3:
 *            gates_w, gates_b, reset_gamma, reset_beta, update_gamma, update_beta,
 *            out_w, out_b, out_gamma, out_beta   (10 each), then prob_w, prob_b
 *   depth_values [host] depth_num floats (depth of plane d, model.py:706-715)
 *   filters (f1,f2,f3): ConvGRU filter counts (16,4,2 for 'normal')
 * The sweep is a wavefront over (plane, cell) on library-owned side streams forked from / joined to `stream`: the set
 * mvs_gru_prepare(stream) created for this caller stream.  The sweep itself creates nothing and never synchronises:
 *   - default formulation at the reference's shape (mvs_gru_set_formulation 0 / 3): the fused sweep, two launches per plane on
 *     `stream` alone, with or without a set, eagerly or under hipGraph capture -- the same bits every time;
 *   - wavefront formulations (1 / 2) on a prepared stream: four streams; on a stream without a set: the same sweep on `stream`
 *     alone (same winning planes, ~1.7x the time at 400 x 300), one note on stderr per process;
 *   - wavefront formulations under hipGraph capture: always the one-stream form -- hip::Stream::EndCapture of the HIP runtime the
 *     PyTorch wheel bundles (7.0.70002) recurses without bound when captured streams wait on each other in both directions (root
 *     cause, native backtrace and a stand-alone reproducer: profiles/r05_capture_wavefront_root_cause.txt,
 *     tools/capture_wavefront_repro.hip; ROCm 7.2's own runtime does not have the defect);
 *   - every exit after the fork, error returns included, first makes `stream` wait for the side streams.
 * The workspace holds a batch of 16 cost slices, two batches of the hoisted x-part of cell 1 and 16-plane state rings:
 * ~1.0 GB at 400 x 300, C = 32 (mvs_gru_workspace_bytes).
 */
size_t mvs_gru_workspace_bytes(int H, int W, int C, int f1, int f2, int f3);
int mvs_gru_wta_f32(const float* ref, const float* src, const float* transforms, int view_num,
                    int depth_num, int H, int W, int C, int f1, int f2, int f3,
                    const float* const* params, const float* depth_values, void* workspace,
                    size_t workspace_bytes, float* depth_out, float* prob_out, void* stream);

/* The same sweep for `views` (1..8) INDEPENDENT reference views of one size in the same launches -- how config 3 shards
 * (mvsnet/inference.py:105-119 loops over ~135 reference views per GPU): every kernel of the wavefront takes a view index
 * from its grid, so the ~7 launches per plane, their latency floors and the cross-stream waits are shared by `views` depth
 * maps.  Per-view results equal the single-view call's (same tiles, same arithmetic; the LayerNorm sums are float per tile
 * and double across tiles, so they do not depend on how tiles are dealt to workgroups).
 *   ref / src / transforms [host] arrays of `views` device pointers (as in mvs_gru_wta_f32, one entry per view)
 *   depth_values [host] views x depth_num floats
 *   workspace    views x mvs_gru_workspace_bytes(...) bytes (one block per view)
 *   depth_out / prob_out (views, H, W)
 * Shape limits (both entry points): f1, f2, f3 <= 64, MVS_E_SHAPE otherwise (the blend kernels keep 2 x F LayerNorm affines per
 * cell in LDS and the staging code is written for at most 64 state channels; the reference uses 16 / 4 / 2, the 'fat' variant
 * 32 / 8 / 4); views in 1..8 (MVS_E_BADARG); C as mvs_cost_volume_f32 accepts it.  The fused two-launch sweep covers C = 32 with 16 / 4 / 2 while a view's
 * workspace block (mvs_gru_workspace_bytes, ~10.2 KB per pixel) stays below 2 GiB -- its kernels address the block with 32-bit
 * byte offsets -- i.e. feature maps up to ~210 k pixels; larger maps (e.g. 576 x 384) and every other accepted shape run the
 * wavefront kernels.  mvs_gru_fused_route is that predicate (1 = fused), decided before anything is enqueued.
 */
int mvs_gru_fused_route(int C, int f1, int f2, int f3, size_t view_block_bytes);
int mvs_gru_wta_batch_f32(const float* const* ref, const float* const* src, const float* const* transforms,
                          int views, int view_num, int depth_num, int H, int W, int C, int f1, int f2,
                          int f3, const float* const* params, const float* depth_values, void* workspace,
                          size_t workspace_bytes, float* depth_out, float* prob_out, void* stream);

/* Set-up of the recurrent sweep for caller stream `stream` on the current device: three side streams, ~30 events, and ONE
 * ~10-30 ms calibration that finds the compute pipe the caller's hardware queue lives on (csrc/gru.hip) so that the
 * side streams avoid it.  THE call of this header that creates resources and synchronises (`stream` and the new streams):
 * call it once per caller stream at start-up (DepthPlan(..., "GRU") does), never inside a latency-critical region or under
 * hipGraph capture (returns MVS_E_NOT_PREPARED there).  Idempotent.  An inconclusive calibration is reported once on stderr and
 * the set is still usable.  At most 16 sets per process (MVS_E_NO_SLOT beyond; mvs_gru_release frees one).  Thread-safe.
 * Optional since round 5: without a set the sweep runs as the fused two-launches-per-plane pipeline on the caller's stream alone
 * (csrc/gru_fused.hip), which needs none; the round-4 four-stream wavefront (mvs_gru_set_formulation 1 / 2) still does. */
int mvs_gru_prepare(void* stream);
/* Tear-down: waits for the set's side streams, destroys them and their events (MVS_E_BADARG: no set for this stream).  Call it
 * before destroying `stream` -- a later stream may receive the same handle on another hardware queue. */
int mvs_gru_release(void* stream);

/* Formulation of the recurrent sweep at the reference's shape (32 feature channels, filters 16 / 4 / 2):
 *   0 (default) = 3 = the FUSED sweep (csrc/gru_fused.hip): two launches per plane on the caller's stream alone carry all three
 *       cells, prob_conv and the winner-take-all update; no stream set needed, captures into a hipGraph;
 *   1 = the round-4 wavefront over a stream set (mvs_gru_prepare) with cell 1's x-part hoisted into batched producer launches,
 *   2 = the same wavefront with full 48-channel per-plane kernels.
 * 1 and 2 give the same bits as each other; the fused sweep runs the same multiply-add chains for cell 1 and differs from them
 * in the last bits (LayerNorm partial sums, the small cells' four partial chains).  Within a formulation a batch of views gives
 * the single view's bits.  Other shapes take the wavefront / generic routes whatever is set.  A tuning / test switch: process-wide
 * atomic, read once at the start of each sweep -- a sweep keeps the formulation it started with. */
int mvs_gru_set_formulation(int form);

/* Diagnostic (csrc/gru_fused.hip): per-workgroup time stamps of the fused two-launches-per-plane sweep.  `buffer` = caller-owned
 * device memory of (1 + 8 * capacity) int64, zeroed by the caller; every workgroup of every fused launch appends one record
 * [launch << 32 | phase << 16 | workgroup, entry, first tile staged, first tile done, loop done, exit (100 MHz ticks), tiles,
 * HW_ID].  null switches it off (the default).  Not thread-safe; tools/gru_fused_trace.py. */
int mvs_gru_fused_trace(void* buffer, int capacity);

/* Diagnostic: the side-stream layout of the set mvs_gru_prepare made for `stream` (MVS_E_NOT_PREPARED without one):
 * *pipe_of_caller = which of the four candidate pipes the caller's hardware queue was measured on (-1: none stood out),
 * probe_us[8] = the calibration chain times of the eight candidate streams (4 high-, 4 low-priority). */
int mvs_gru_stream_layout(void* stream, int* pipe_of_caller, float* probe_us);

/* ---------------------------------------------------------------------------------------------
 * backward passes of the plane-sweep path for training.  The reference has no such
 * functions: TensorFlow differentiates the graph of `inference` (mvsnet/model.py:257-372) inside
 * opt.compute_gradients (mvsnet/train.py:428-429).  Input gradients of the 3D convolutions need no
 * entry point of their own: a stride-2 convolution's input gradient is mvs_deconv3d_f32 with the SAME
 * kernel array, a transposed convolution's is mvs_conv3d_f32 (stride 2) with the same array, and a
 * stride-1 convolution's is mvs_conv3d_f32 with the kernel flipped along kd,kh,kw and Cin/Cout swapped.
 *
 * mvs_softargmin_bwd_f32   g_reg(d) = -g_depth * P_d * (z_d - depth) - g_prob * P_d * (m_d - prob), P = softmax(-reg),
 *                          m_d = number of the four probability buckets equal to plane d (model.py:343-366, 45-144;
 *                          the bucket indices carry no gradient); g_depth or g_prob may be NULL
 * mvs_bn_relu_f32          out = act(y*scale+shift) [+ act(y2*scale2+shift2)], act = ReLU when the scale
 *                          is given: the normalised layer input the forward kernels form on load
 * mvs_bn_bwd_reduce_f32    BatchNorm(batch statistics)+ReLU backward, pass 1: sums (S,2,C) float64 (zeroed
 *                          by the caller; S = mvs_bn_bwd_sum_slots() partial rows that pass 2 folds, so that
 *                          the float64 atomics of ~1000 workgroups do not serialise) += [sum gz, sum gz*xhat], gz = (g1 [+ g2]) * [y*scale+shift > 0],
 *                          xhat from `stats` (the forward's (2,C) sums) / count / eps (network.py:492-509)
 * mvs_bn_bwd_apply_f32     pass 2: g_y = gamma/std * (gz - mean(gz) - xhat*mean(gz*xhat));
 *                          g_gamma = sum gz*xhat, g_beta = sum gz (either may be NULL)
 * mvs_conv3d_wgrad_f32     dW(tap,cb,cs) = sum_o big(stride*o + tap - pad, cb) * small(o, cs); big (D,H,W,Cbig),
 *                          small (D/s,H/s,W/s,Csmall); convolution: big = layer input, small = output
 *                          gradient -> dW in the conv3d layout; transposed convolution (stride 2): big =
 *                          output gradient, small = layer input -> dW in the conv3d_transpose layout.
 *                          Deterministic (per-workgroup partials in `workspace`, fixed-order float64 sum).
 *                          Built for the channel pairs of RegNetUS0 'normal' (else MVS_E_SHAPE).
 * mvs_cost_volume_bwd_f32  gradient of mvs_cost_volume_f32 (either variant) w.r.t. the feature maps:
 *                          g_ref (H,W,C) and g_src (view_num-1,H,W,C), zeroed by the caller, are
 *                          accumulated with float atomics; g2 may be NULL (second consumer of the volume)
 * mvs_rmsprop_step_f32     tf.train.RMSPropOptimizer update over a flat parameter buffer (train.py:259):
 *                          ms += (g^2-ms)(1-decay); mom = momentum*mom + lr*g/sqrt(ms+eps); w -= mom,
 *                          g = grad * grad_scale (1/world_size after a sum all-reduce)
 */
int mvs_softargmin_bwd_f32(const float* reg, const float* g_depth, const float* g_prob, int D, int H, int W,
                           float depth_start, float depth_interval, int inverse_depth,
                           float* g_reg, void* stream);
int mvs_bn_relu_f32(const float* y, const float* scale, const float* shift, const float* y2,
                    const float* scale2, const float* shift2, size_t voxels, int C, float* out,
                    void* stream);
int mvs_bn_bwd_sum_slots(void);
int mvs_bn_bwd_reduce_f32(const float* y, const double* stats, double count, float eps,
                          const float* scale, const float* shift, const float* g1,
                          const float* g2, size_t voxels, int C, double* sums, void* stream);
int mvs_bn_bwd_apply_f32(const float* y, const double* stats, double count, float eps,
                         const float* scale, const float* shift, const float* gamma,
                         const float* g1, const float* g2, const double* sums, size_t voxels,
                         int C, float* g_y, float* g_gamma, float* g_beta, void* stream);
size_t mvs_conv3d_wgrad_workspace_bytes(int D, int H, int W, int Cbig, int Csmall, int stride);
int mvs_conv3d_wgrad_f32(const float* big, const float* small, int D, int H, int W, int Cbig,
                         int Csmall, int stride, void* workspace, size_t workspace_bytes,
                         float* dw, void* stream);
int mvs_cost_volume_bwd_f32(const float* ref, const float* src, const float* transforms,
                            int view_num, int depth_num, int H, int W, int C, const float* g1,
                            const float* g2, float* g_ref, float* g_src, void* stream);
/* Atomic-free, bit-reproducible variant of mvs_cost_volume_bwd_f32 (C = 32 or 16): stores the per-view
 * warped-sample gradients in `workspace` (mvs_cost_volume_bwd_workspace_bytes: (N-1)*D*H*W*C floats plus
 * chunk rows) and gathers them in the source frame through the inverse plane homographies.  g_ref / g_src
 * are overwritten (no zeroing needed).  Planes that minify a source view more than ~8x fall outside its
 * candidate search; use the scatter version for such geometry. */
size_t mvs_cost_volume_bwd_workspace_bytes(int view_num, int depth_num, int H, int W, int C);
int mvs_cost_volume_bwd_gather_f32(const float* ref, const float* src, const float* transforms,
                                   int view_num, int depth_num, int H, int W, int C, const float* g1,
                                   const float* g2, void* workspace, size_t workspace_bytes,
                                   float* g_ref, float* g_src, void* stream);
int mvs_rmsprop_step_f32(float* w, const float* g, float* ms, float* mom, size_t n, float lr,
                         float decay, float momentum, float eps, float grad_scale, void* stream);
/* GroupNorm (+ReLU) of the 2D towers for TRAINING (Network.conv_gn / deconv_gn, network.py:217-276, 350-409:
 * groups of 8 channels, biased variance; inference fuses GroupNorm into mvs_conv2d_gn_f32 instead).
 *   x, y, g, dx   (V, hw, C) channel-last, C a multiple of 8
 *   stats         (V, 2, C) float64 per-channel [sum, sum of squares] of x, zeroed by the caller before
 *                 mvs_gn_stats_f32 (group moments are folded from a group's 8 channel sums where needed)
 *   sums          mvs_gn_bwd_sums_doubles(V, C) float64 = mvs_gn_bwd_sum_slots() copies of (V, 2, C) [sum gz, sum gz*xhat],
 *                 zeroed by the caller before mvs_gn_bwd_reduce_f32: the workgroups spread their float64 atomics over the
 *                 copies (atomics on one address are performed one after the other), mvs_gn_bwd_apply_f32 adds them up;
 *                 g_beta(c) = sum over slots and views of sums(s,v,0,c), g_gamma(c) likewise of sums(s,v,1,c)
 *   relu          1: y = ReLU(gamma*xhat+beta) (conv_gn), 0: no activation (deconv_gn) */
int mvs_gn_stats_f32(const float* x, int V, size_t hw, int C, double* stats, void* stream);
/* The same statistics from the sums the tower convolutions already wrote: slots (V, C/8, nslot, 2) float64 partial [sum, sumsq]
 * per 8-channel group (mvs_conv2d_gn_f32 / mvs_deconv2d_gn_f32, nslot = mvs_gn_stat_slots()) -> stats (V, 2, C), every channel
 * carrying an eighth of its group's totals (the group moments the kernels below fold from 8 channel sums are then the forward's). */
int mvs_gn_slots_to_channel_sums_f64(const double* slots, int V, int C, int nslot, double* stats, void* stream);
/* ... for n layers in one launch: layer i's slots start slot_off[i] float64 behind `slots`, its (V, 2, C[i]) statistics
 * stat_off[i] behind `stats` (host arrays of n entries). */
int mvs_gn_slots_to_channel_sums_many_f64(int n, const double* slots, const long long* slot_off, const int* C, int V, int nslot,
                                          double* stats, const long long* stat_off, void* stream);
int mvs_gn_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                     int relu, int V, size_t hw, int C, float* y, void* stream);
int mvs_gn_bwd_sum_slots(void);
size_t mvs_gn_bwd_sums_doubles(int V, int C);
int mvs_gn_bwd_reduce_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                          int relu, const float* g, int V, size_t hw, int C, double* sums, void* stream);
int mvs_gn_bwd_apply_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                         int relu, const float* g, const double* sums, int V, size_t hw, int C, float* dx,
                         void* stream);
/* ... with g_beta / g_gamma ADDED to totals (2, C) float64 by the launch's first workgroup (the parameter gradients without a
 * reduction launch per layer); C <= 128. */
int mvs_gn_bwd_apply_tot_f32(const float* x, const double* stats, const float* gamma, const float* beta, float eps,
                             int relu, const float* g, const double* sums, double* totals, int V, size_t hw, int C, float* dx,
                             void* stream);
/* The other two optimisers of setup_optimizer (train.py:248-271): tf.train.MomentumOptimizer
 * (accum = momentum*accum + g; w -= lr*accum) and tf.train.AdamOptimizer (lr_t = lr*sqrt(1-beta2^t)/
 * (1-beta1^t) formed by the caller; w -= lr_t*m/(sqrt(v)+eps)). */
int mvs_momentum_step_f32(float* w, const float* g, float* accum, size_t n, float lr, float momentum,
                          float grad_scale, void* stream);
int mvs_adam_step_f32(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float beta1,
                      float beta2, float eps, float grad_scale, void* stream);

/* Many small tensors in one launch (the end of the towers' backward, replacing a permute-copy and an autograd accumulation
 * launch per parameter -- average_gradients' inputs, train.py:155-187, land in the flat gradient buffer directly):
 *   mvs_transpose_add_many_f32   job i: dst_i (KK, keep, B) += src_i (B, A, KK) with the axes reversed, rows a >= keep dropped
 *                                (ATen's (Cout, Cin, k, k) weight gradient into TensorFlow's (k, k, Cin, Cout) variable);
 *                                dims = n x 4 host ints [B, A, KK, keep], src / dst host arrays of n device pointers
 *   mvs_add_f64_many_f32         job i: dst_i[0..counts_i) += (float)src_i[..]  (GroupNorm gamma / beta gradients accumulated
 *                                in float64 by mvs_gn_bwd_apply_tot_f32)
 * Any n (the jobs travel in the kernel arguments, 48 / 96 per launch). */
int mvs_transpose_add_many_f32(int n, const float* const* src, float* const* dst, const int* dims, void* stream);
int mvs_add_f64_many_f32(int n, const double* const* src, float* const* dst, const int* counts, void* stream);
/*   mvs_add_many_f32             job i: dst_i[0..counts_i) += src_i[..] (the regulariser's parameter gradients, already in the
 *                                variables' layouts, into the flat gradient buffer) */
int mvs_add_many_f32(int n, const float* const* src, float* const* dst, const long long* counts, void* stream);

/* Training of the recurrent regulariser (inference_prob_recurrent, mvsnet/model.py:505-599; ConvGRUCell,
 * mvsnet/convgru.py:82-122): the plane-sequential part of back-propagation through time of ONE cell over all D
 * planes (csrc/gru_train.hip).  px (D,H,W,3F) holds the x parts of the cell's two convolutions with their biases,
 * channels [reset | update | candidate]; wgh (3,3,F,2F) / woh (3,3,F,F) are the h parts of the two kernels;
 * ln (6,F) = reset gamma, reset beta, update gamma, update beta, candidate gamma, candidate beta.
 * Forward keeps g (D,H,W,2F), c (D,H,W,F) (raw convolutions), rh (D,H,W,F) = r*h, h (D+1,H,W,F) with h[0] the initial
 * state set by the caller, and the LayerNorm moments stats (D, forward_slots, 6) float64, zeroed by the caller.
 * Backward takes gh (D,H,W,F), the gradient reaching every state from outside the recurrence, the flipped /
 * transposed kernels wgh_t (3,3,2F,F), woh_t (3,3,F,F), and returns gpx (D,H,W,3F), the gradient w.r.t. px (from
 * which the host forms every input, weight and bias gradient with batched convolutions), and
 * part (D,3,backward_slots,2,F) float64 (zeroed by the caller): per plane and LayerNorm the sums over pixels of
 * dz and dz*xhat, i.e. the gradients of beta and gamma once summed over planes and slots.
 * scratch: 6*H*W*F floats.  dh_in (H,W,F): gradient w.r.t. the state LEAVING the last plane that arrives from
 * planes above this call's range (null: none, and then the caller zeroes `scratch`); dh_out (H,W,F), optional:
 * receives the gradient w.r.t. h[0], the state entering plane 0 -- so a sweep can be run in chunks of planes (forward
 * in ascending chunks with h carried through the (D+1)-plane buffer, backward in descending chunks with dh_out of
 * one call as dh_in of the next), which is what lets the three cells run as a wavefront on three streams.
 * F in {16, 8, 4, 2, 1}. */
int mvs_gru_train_slots(int* forward_slots, int* backward_slots);
int mvs_gru_train_cell_fwd_f32(const float* px, const float* wgh, const float* woh, const float* ln, int D, int H,
                               int W, int F, float* g, float* c, float* rh, float* h, double* stats, void* stream);
int mvs_gru_train_cell_bwd_f32(const float* gh, const float* g, const float* c, const float* h, const double* stats,
                               const float* wgh_t, const float* woh_t, const float* ln, int D, int H, int W, int F,
                               float* gpx, double* part, float* scratch, const float* dh_in, float* dh_out,
                               void* stream);

/* Weight gradient of a 3x3 SAME stride-1 2D convolution over a batch of planes (csrc/conv2d_wgrad.hip; in the
 * reference: TensorFlow's Conv2DBackpropFilter of the tf.layers.conv2d calls of mvsnet/convgru.py:92,110 behind
 * opt.compute_gradients, train.py:428-429):  dw (3,3,Cin,Cout) = sum over n,y,x of x[n,y+kh-1,x+kw-1,ci] * g[n,y,x,co].
 * x (N,H,W,Cin); g (N,H,W,g_stride) of which channels [g_off, g_off+Cout) are used (g_stride, g_off multiples of 4).
 * Deterministic (per-workgroup partials in `workspace`, fixed-order float64 sum).  Built for (Cin, Cout) in
 * {16,32} x {16,32,48}; anything else returns MVS_E_SHAPE. */
size_t mvs_conv2d_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int mvs_conv2d_wgrad_f32(const float* x, const float* g, int g_stride, int g_off, int N, int H, int W, int Cin,
                         int Cout, void* workspace, size_t workspace_bytes, float* dw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MVSNET_HIP_H_ */
