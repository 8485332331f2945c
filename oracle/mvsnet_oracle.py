"""CPU restatement (numpy) of MVSNet's plane-sweep depth-inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mvsnet_amd/`` may import this module; it is
the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  The product path is the HIP library and fails loudly without it.

PARITY UNPINNED BY THE REFERENCE: ubiquity6/MVSNet ships no tests, golden vectors,
fixtures or checkpoints, and its arithmetic lives in TensorFlow 1.12 kernels that are not
in the repository and not installable here (``import tensorflow`` -> ModuleNotFoundError;
python2-only code).  Every TF-op semantic restated below is therefore pinned by the
hand-computed known-answer tests in ``tests/test_oracle_kat.py``,
not by the reference's own outputs.

Every function cites the reference file:line (relative to the reference checkout) whose
behaviour it restates.  ``dtype`` selects the arithmetic type: ``np.float32`` follows the
reference's fp32 operation order as closely as numpy allows; ``np.float64`` is the strict
high-precision form used to bound fp32 rounding noise.

Layouts are the reference's: channel-last, images (H, W, C), volumes (D, H, W, C),
cams (2, 4, 4) with cams[0] = extrinsic, cams[1][:3,:3] = K, cams[1][3] = (depth_min,
depth_interval, depth_num, depth_max).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# R1 / R1'  plane-induced homographies
# --------------------------------------------------------------------------------------


def _depth_samples(depth_num, depth_start, depth_interval, dtype):
    """mvsnet/homography_warping.py:26-30 : depth_d = start + d * interval."""
    d = np.arange(int(depth_num), dtype=dtype)
    return d * dtype(depth_interval) + dtype(depth_start)


def _inv_depth_samples(depth_num, depth_start, depth_end, dtype):
    """mvsnet/homography_warping.py:74-77 : 1 / linspace(1/start, 1/end, D).

    tf.lin_space(a, b, n)[i] = a + i * (b - a) / (n - 1)  (SURVEY 8c item 6).
    """
    a = dtype(1.0) / dtype(depth_start)
    b = dtype(1.0) / dtype(depth_end)
    n = int(depth_num)
    i = np.arange(n, dtype=dtype)
    step = (b - a) / dtype(max(n - 1, 1))
    inv = a + i * step
    return dtype(1.0) / inv


def _homographies_for_depths(left_cam, right_cam, depth, dtype):
    """mvsnet/homography_warping.py:33-56.

    H[d] = K_r R_r (I - (c_r - c_l) n_l^T / depth_d) R_l^T K_l^-1,
    n_l^T = third row of R_l, c = -R^T t.  "left" is the reference view.
    """
    left_cam = np.asarray(left_cam, dtype=dtype)
    right_cam = np.asarray(right_cam, dtype=dtype)
    R_left = left_cam[0, :3, :3]
    R_right = right_cam[0, :3, :3]
    t_left = left_cam[0, :3, 3:4]
    t_right = right_cam[0, :3, 3:4]
    K_left = left_cam[1, :3, :3]
    K_right = right_cam[1, :3, :3]

    K_left_inv = np.linalg.inv(K_left.astype(np.float64)).astype(dtype)  # :33
    R_left_trans = R_left.T                                               # :34
    R_right_trans = R_right.T                                             # :35
    fronto_direction = R_left[2:3, :]                                     # :37  (1,3)
    c_left = -(R_left_trans @ t_left)                                     # :39
    c_right = -(R_right_trans @ t_right)                                  # :40
    c_relative = c_right - c_left                                         # :41
    temp_vec = c_relative @ fronto_direction                              # :45  (3,3)
    middle_mat1 = R_left_trans @ K_left_inv                               # :51
    eye = np.eye(3, dtype=dtype)
    out = np.empty((depth.shape[0], 3, 3), dtype=dtype)
    for i, dep in enumerate(depth):
        middle_mat0 = eye - temp_vec / dtype(dep)                         # :50
        middle_mat2 = middle_mat0 @ middle_mat1                           # :52
        out[i] = K_right @ (R_right @ middle_mat2)                        # :54-56
    return out


def get_homographies(left_cam, right_cam, depth_num, depth_start, depth_interval,
                     dtype=np.float32):
    """mvsnet/homography_warping.py:10-58 -> (D, 3, 3)."""
    depth = _depth_samples(depth_num, depth_start, depth_interval, dtype)
    return _homographies_for_depths(left_cam, right_cam, depth, dtype)


def get_homographies_inv_depth(left_cam, right_cam, depth_num, depth_start, depth_end,
                               dtype=np.float32):
    """mvsnet/homography_warping.py:60-106 -> (D, 3, 3); batch 1 only, as the reference."""
    depth = _inv_depth_samples(depth_num, depth_start, depth_end, dtype)
    return _homographies_for_depths(left_cam, right_cam, depth, dtype)


# --------------------------------------------------------------------------------------
# R2  pixel-centre conversion + tf.contrib.image.transform(BILINEAR)
# --------------------------------------------------------------------------------------


def homography_to_transform8(homography, dtype=np.float32):
    """mvsnet/homography_warping.py:216-250 : (…,3,3) -> (…,8).

    Converts an image-coordinate homography (x_img = x_pix + 0.5) to the 8-vector that
    tf.contrib.image.transform consumes in pixel coordinates, normalised by c2'.
    """
    h = np.asarray(homography, dtype=dtype).reshape(-1, 9)
    a0, a1, a2, b0, b1, b2, c0, c1, c2 = [h[:, i] for i in range(9)]
    two, four = dtype(2), dtype(4)
    a_0 = a0 - c0 / two                                               # :226
    a_1 = a1 - c1 / two                                               # :227
    a_2 = (a0 + a1) / two + a2 - (c0 + c1) / four - c2 / two          # :228
    b_0 = b0 - c0 / two                                               # :229
    b_1 = b1 - c1 / two                                               # :230
    b_2 = (b0 + b1) / two + b2 - (c0 + c1) / four - c2 / two          # :231
    c_0 = c0                                                          # :232
    c_1 = c1                                                          # :233
    c_2 = c2 + (c0 + c1) / two                                        # :234
    lin = np.stack([a_0, a_1, a_2, b_0, b_1, b_2, c_0, c_1], axis=1)
    lin = lin / c_2[:, None]                                          # :248-250
    return lin.reshape(np.asarray(homography).shape[:-2] + (8,)).astype(dtype)


def image_projective_transform_bilinear(image, t8, dtype=np.float32):
    """tf.contrib.image.transform(image, t8, 'BILINEAR') of TensorFlow 1.12
    (call site mvsnet/homography_warping.py:251-252; kernel = ImageProjectiveTransform,
    third-party, restated from its published algorithm).

    For every integer output pixel (x, y):
        p  = c0*x + c1*y + 1
        sx = (a0*x + a1*y + a2) / p ;  sy = (b0*x + b1*y + b2) / p
        x0 = floor(sx), x1 = x0+1, y0 = floor(sy), y1 = y0+1
        out = (y1-sy)*[(x1-sx)*I(y0,x0) + (sx-x0)*I(y0,x1)]
            + (sy-y0)*[(x1-sx)*I(y1,x0) + (sx-x0)*I(y1,x1)]
    where each tap I(.,.) individually reads 0 when its index is outside the image.
    image: (H, W, C); t8: (8,).
    """
    img = np.asarray(image, dtype=dtype)
    H, W, C = img.shape
    t = np.asarray(t8, dtype=dtype)
    xs = np.arange(W, dtype=dtype)[None, :]
    ys = np.arange(H, dtype=dtype)[:, None]
    proj = t[6] * xs + t[7] * ys + dtype(1)
    sx = (t[0] * xs + t[1] * ys + t[2]) / proj
    sy = (t[3] * xs + t[4] * ys + t[5]) / proj
    x0f = np.floor(sx)
    y0f = np.floor(sy)
    x1f = x0f + dtype(1)
    y1f = y0f + dtype(1)

    def read(yf, xf):
        with np.errstate(invalid="ignore"):
            ok = (yf >= 0) & (yf < H) & (xf >= 0) & (xf < W)   # NaN/inf -> False
        yi = np.where(ok, yf, 0).astype(np.int64)
        xi = np.where(ok, xf, 0).astype(np.int64)
        v = img[yi, xi, :]
        return np.where(ok[..., None], v, dtype(0))

    wx1 = (x1f - sx)[..., None]
    wx0 = (sx - x0f)[..., None]
    wy1 = (y1f - sy)[..., None]
    wy0 = (sy - y0f)[..., None]
    with np.errstate(invalid="ignore"):
        v_floor = wx1 * read(y0f, x0f) + wx0 * read(y0f, x1f)
        v_ceil = wx1 * read(y1f, x0f) + wx0 * read(y1f, x1f)
        out = wy1 * v_floor + wy0 * v_ceil
    return out.astype(dtype)


def tf_transform_homography(image, homography, dtype=np.float32):
    """mvsnet/homography_warping.py:211-253 (the ACTIVE warp; zero-fill per tap)."""
    return image_projective_transform_bilinear(
        image, homography_to_transform8(homography, dtype), dtype)


def homography_warping_clamp(image, homography, dtype=np.float32):
    """mvsnet/homography_warping.py:108-210 (DEAD code in the reference, every call site
    is commented out: model.py:325,444,579,686).  Same geometry in image coordinates but
    tap indices are clamped to the border (:146-149) instead of zero-filled.  Kept only
    as the optional border='clamp' mode."""
    img = np.asarray(image, dtype=dtype)
    Hh, W, C = img.shape
    Hm = np.asarray(homography, dtype=dtype)
    xs = (np.arange(W, dtype=dtype) + dtype(0.5))[None, :] * np.ones((Hh, 1), dtype)
    ys = (np.arange(Hh, dtype=dtype) + dtype(0.5))[:, None] * np.ones((1, W), dtype)
    ax = Hm[0, 0] * xs + Hm[0, 1] * ys + Hm[0, 2]
    ay = Hm[1, 0] * xs + Hm[1, 1] * ys + Hm[1, 2]
    dv = Hm[2, 0] * xs + Hm[2, 1] * ys + Hm[2, 2]
    dv = dv + (dv == 0).astype(dtype) * dtype(1e-7)                   # :197
    x = ax / dv - dtype(0.5)                                          # :138
    y = ay / dv - dtype(0.5)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1
    x0 = np.clip(x0, 0, W - 1); x1 = np.clip(x1, 0, W - 1)            # :146-149
    y0 = np.clip(y0, 0, Hh - 1); y1 = np.clip(y1, 0, Hh - 1)
    x0f, x1f, y0f, y1f = [a.astype(dtype) for a in (x0, x1, y0, y1)]
    a = ((y1f - y) * (x1f - x))[..., None]
    b = ((y1f - y) * (x - x0f))[..., None]
    c = ((y - y0f) * (x1f - x))[..., None]
    d = ((y - y0f) * (x - x0f))[..., None]
    return (a * img[y0, x0] + b * img[y0, x1] + c * img[y1, x0] + d * img[y1, x1]).astype(dtype)


# --------------------------------------------------------------------------------------
# R3  variance cost volume
# --------------------------------------------------------------------------------------


def variance_cost_mem(ref_feature, warped_list, view_num, dtype=np.float32):
    """Per-depth cost of inference_mem, mvsnet/model.py:436-462:
        S = F_ref + sum W_v ;  Q = F_ref^2 + sum W_v^2
        cost = Q / N - S^2 / (N*N)        (N = FLAGS.view_num, INCLUDING the reference)
    Single-pass E[x^2]-E[x]^2 is kept on purpose."""
    ref = np.asarray(ref_feature, dtype=dtype)
    S = ref.copy()
    Q = ref * ref
    for w in warped_list:
        w = np.asarray(w, dtype=dtype)
        S = S + w                                                     # :447
        Q = Q + w * w                                                 # :448-449
    n = dtype(view_num)
    ave = (S * S) / dtype(view_num * view_num)                        # :458-459
    return (Q / n - ave).astype(dtype)                                # :460-461


def variance_cost_eager(ref_feature, warped_list, view_num, dtype=np.float32):
    """Per-depth cost of inference()/GRU twins, mvsnet/model.py:319-332,680-693:
        cost = Q/N - (S/N)^2."""
    ref = np.asarray(ref_feature, dtype=dtype)
    S = ref.copy()
    Q = ref * ref
    for w in warped_list:
        w = np.asarray(w, dtype=dtype)
        S = S + w
        Q = Q + w * w
    n = dtype(view_num)
    S = S / n
    Q = Q / n
    return (Q - S * S).astype(dtype)


def cost_volume(ref_feature, src_features, homographies, view_num=None, variant="mem",
                dtype=np.float32):
    """mvsnet/model.py:422-463 (variant='mem') / :315-334 (variant='eager').

    ref_feature (H,W,C); src_features (N-1,H,W,C); homographies (N-1,D,3,3)
    -> (D,H,W,C)."""
    src = np.asarray(src_features, dtype=dtype)
    Hs = np.asarray(homographies, dtype=dtype)
    n_src, D = Hs.shape[0], Hs.shape[1]
    if view_num is None:
        view_num = n_src + 1
    fn = variance_cost_mem if variant == "mem" else variance_cost_eager
    out = np.empty((D,) + np.asarray(ref_feature).shape, dtype=dtype)
    for d in range(D):
        warped = [tf_transform_homography(src[v], Hs[v, d], dtype) for v in range(n_src)]
        out[d] = fn(ref_feature, warped, view_num, dtype)
    return out


# --------------------------------------------------------------------------------------
# R4 / R5  conv3d, conv3d_transpose, batch-norm (batch statistics), RegNetUS0
# --------------------------------------------------------------------------------------


def same_pad(n, k, s):
    """TensorFlow 'SAME' padding: out = ceil(n/s); total = max((out-1)*s + k - n, 0);
    before = total // 2, after = total - before."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    before = total // 2
    return out, before, total - before


def convnd_same(x, w, stride, dtype=np.float32):
    """tf.layers.conv2d / conv3d, padding='SAME', no bias, cross-correlation
    (mvsnet/cnn_wrapper/network.py:203-210).  x: (*spatial, Cin) channel-last;
    w: (*k, Cin, Cout) (TF kernel layout).  Same stride on every spatial axis."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    nd = x.ndim - 1
    ks = w.shape[:nd]
    cin, cout = w.shape[nd], w.shape[nd + 1]
    assert x.shape[-1] == cin
    outs, pads = [], []
    for a in range(nd):
        o, pb, pa = same_pad(x.shape[a], ks[a], stride)
        outs.append(o)
        pads.append((pb, pa))
    xp = np.pad(x, pads + [(0, 0)])
    acc = np.zeros(tuple(outs) + (cout,), dtype=dtype)
    for kidx in np.ndindex(*ks):
        sl = tuple(slice(kidx[a], kidx[a] + (outs[a] - 1) * stride + 1, stride)
                   for a in range(nd))
        acc += xp[sl] @ w[kidx]
    return acc


def conv3d_same(x, w, stride, dtype=np.float32):
    """tf.layers.conv3d SAME (network.py:210).  x (D,H,W,Cin), w (3,3,3,Cin,Cout)."""
    return convnd_same(x, w, stride, dtype)


def conv2d_same(x, w, stride=1, bias=None, dtype=np.float32):
    """tf.layers.conv2d SAME (network.py:205; convgru.py:92,110; model.py:701)."""
    y = convnd_same(x, w, stride, dtype)
    if bias is not None:
        y = y + np.asarray(bias, dtype=dtype)
    return y


def convnd_transpose_same(x, w, stride=2, dtype=np.float32):
    """tf.layers.conv2d_transpose / conv3d_transpose, padding='SAME', no bias
    (network.py:325-327).  out spatial = in * stride.  Defined as the gradient of the
    SAME forward conv: full[o = i*stride + k - pad_before] += x[i] * w[k][co][ci], cropped
    to [0, n*stride)  (SURVEY 8c item 3).  w: (*k, Cout, Cin) (TF transpose layout)."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    nd = x.ndim - 1
    ks = w.shape[:nd]
    cout, cin = w.shape[nd], w.shape[nd + 1]
    assert x.shape[-1] == cin
    outs = [x.shape[a] * stride for a in range(nd)]
    pbs = [same_pad(outs[a], ks[a], stride)[1] for a in range(nd)]
    full = [(x.shape[a] - 1) * stride + ks[a] for a in range(nd)]
    acc = np.zeros(tuple(full) + (cout,), dtype=dtype)
    for kidx in np.ndindex(*ks):
        sl = tuple(slice(kidx[a], kidx[a] + (x.shape[a] - 1) * stride + 1, stride)
                   for a in range(nd))
        acc[sl] += x @ w[kidx].T
    crop = tuple(slice(pbs[a], pbs[a] + outs[a]) for a in range(nd))
    return np.ascontiguousarray(acc[crop])


def conv3d_transpose_same(x, w, stride=2, dtype=np.float32):
    """tf.layers.conv3d_transpose SAME stride 2 (network.py:327); w (3,3,3,Cout,Cin)."""
    return convnd_transpose_same(x, w, stride, dtype)


def batch_norm_train(x, gamma, beta, eps=1e-5, relu=True, dtype=np.float32):
    """tf.layers.batch_normalization(training=True, fused=True) + ReLU
    (network.py:492-509; the inference graph runs BN in training mode: SURVEY 3.1 note).
    Per-channel mean and BIASED variance over every non-channel axis, statistics taken in
    float64 then applied in ``dtype``."""
    x = np.asarray(x, dtype=dtype)
    axes = tuple(range(x.ndim - 1))
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=axes)
    var = x64.var(axis=axes)
    inv = (np.asarray(gamma, np.float64) / np.sqrt(var + eps))
    y = (x * inv.astype(dtype) + (np.asarray(beta, np.float64) - mean * inv).astype(dtype))
    if relu:
        y = np.maximum(y, dtype(0))
    return y.astype(dtype)


REGNET_LAYERS = (
    # name, kind, cin_mult, cout_mult, stride   (multiples of base_filter = 8 / divisor)
    ("3dconv1_0", "conv", 4, 2, 2),
    ("3dconv2_0", "conv", 2, 4, 2),
    ("3dconv3_0", "conv", 4, 8, 2),
    ("3dconv0_1", "conv", 4, 1, 1),
    ("3dconv1_1", "conv", 2, 2, 1),
    ("3dconv2_1", "conv", 4, 4, 1),
    ("3dconv3_1", "conv", 8, 8, 1),
    ("3dconv4_0", "deconv", 8, 4, 2),
    ("3dconv5_0", "deconv", 4, 2, 2),
    ("3dconv6_0", "deconv", 2, 1, 2),
    ("3dconv6_2", "conv", 1, 0, 1),   # cout = 1, no BN / ReLU / bias
)


def regnet_us0(cost, params, dtype=np.float32, eps=1e-5):
    """RegNetUS0, mvsnet/cnn_wrapper/mvsnetworks.py:122-158.  cost (D,H,W,Cin) ->
    (D,H,W).  params[name] = {'w': kernel, 'gamma': .., 'beta': ..}."""
    def cb(x, name, stride):
        p = params[name]
        return batch_norm_train(conv3d_same(x, p["w"], stride, dtype),
                                p["gamma"], p["beta"], eps, True, dtype)

    def db(x, name):
        p = params[name]
        return batch_norm_train(conv3d_transpose_same(x, p["w"], 2, dtype),
                                p["gamma"], p["beta"], eps, True, dtype)

    x = np.asarray(cost, dtype=dtype)
    c1_0 = cb(x, "3dconv1_0", 2)                # :130-131
    c2_0 = cb(c1_0, "3dconv2_0", 2)             # :132
    c3_0 = cb(c2_0, "3dconv3_0", 2)             # :133
    c0_1 = cb(x, "3dconv0_1", 1)                # :135-136
    c1_1 = cb(c1_0, "3dconv1_1", 1)             # :138-139
    c2_1 = cb(c2_0, "3dconv2_1", 1)             # :141-142
    c3_1 = cb(c3_0, "3dconv3_1", 1)             # :144-145
    c4_0 = db(c3_1, "3dconv4_0")                # :146
    c4_1 = c4_0 + c2_1                          # :148-149
    c5_0 = db(c4_1, "3dconv5_0")                # :150
    c5_1 = c5_0 + c1_1                          # :152-153
    c6_0 = db(c5_1, "3dconv6_0")                # :154
    c6_1 = c6_0 + c0_1                          # :156-157
    out = conv3d_same(c6_1, params["3dconv6_2"]["w"], 1, dtype)   # :158
    return out[..., 0]


def regnet_us0_batch(costs, params, dtype=np.float32, eps=1e-5):
    """RegNetUS0 on a batch (FLAGS.batch_size > 1, model.py:466-469): convolutions per sample, every BatchNorm layer's
    mean / biased variance over (B,D,H,W) (tf.layers.batch_normalization reduces over all but the channel axis,
    network.py:496-506).  costs (B,D,H,W,Cin) -> (B,D,H,W)."""
    B = len(costs)

    def bn(ys, name):
        z = batch_norm_train(np.stack(ys), params[name]["gamma"], params[name]["beta"], eps, True, dtype)
        return [z[b] for b in range(B)]
    cb = lambda xs, name, stride: bn([conv3d_same(x, params[name]["w"], stride, dtype) for x in xs], name)
    db = lambda xs, name: bn([conv3d_transpose_same(x, params[name]["w"], 2, dtype) for x in xs], name)
    add = lambda a, b: [u + v for u, v in zip(a, b)]
    x = [np.asarray(c, dtype=dtype) for c in costs]
    c1_0 = cb(x, "3dconv1_0", 2); c2_0 = cb(c1_0, "3dconv2_0", 2); c3_0 = cb(c2_0, "3dconv3_0", 2)
    c0_1 = cb(x, "3dconv0_1", 1); c1_1 = cb(c1_0, "3dconv1_1", 1); c2_1 = cb(c2_0, "3dconv2_1", 1)
    c3_1 = cb(c3_0, "3dconv3_1", 1)
    c4 = add(db(c3_1, "3dconv4_0"), c2_1)
    c5 = add(db(c4, "3dconv5_0"), c1_1)
    c6 = add(db(c5, "3dconv6_0"), c0_1)
    return np.stack([conv3d_same(v, params["3dconv6_2"]["w"], 1, dtype)[..., 0] for v in c6])


# --------------------------------------------------------------------------------------
# R6 / R7  softmax over depth, soft-argmin, 4-bucket probability map
# --------------------------------------------------------------------------------------


def depth_values(depth_num, depth_start, depth_interval, inverse_depth=False,
                 dtype=np.float32):
    """mvsnet/model.py:378-379,480-488: linspace(start, end, D) with
    end = start + (D-1)*interval, or 1/linspace(1/start, 1/end, D)."""
    D = int(depth_num)
    start = dtype(depth_start)
    end = start + (dtype(D) - dtype(1)) * dtype(depth_interval)       # :378-379
    if inverse_depth:
        return _inv_depth_samples(D, start, end, dtype)               # :481-485
    i = np.arange(D, dtype=dtype)
    return (start + i * ((end - start) / dtype(max(D - 1, 1)))).astype(dtype)   # :487-488


def softmax_neg(reg, dtype=np.float32):
    """tf.nn.softmax(-reg, axis=depth), mvsnet/model.py:474-475.  reg (D,H,W)."""
    z = -np.asarray(reg, dtype=dtype)
    z = z - z.max(axis=0, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=0, keepdims=True)).astype(dtype)


def soft_argmin(prob, depth_num, depth_start, depth_interval, inverse_depth=False,
                dtype=np.float32):
    """mvsnet/model.py:477-495: depth = sum_d P_d * z_d -> (H,W)."""
    z = depth_values(depth_num, depth_start, depth_interval, inverse_depth, dtype)
    return (np.asarray(prob, dtype=dtype) * z[:, None, None]).sum(axis=0).astype(dtype)


def probability_map(prob, depth_map, depth_start, depth_interval, inverse_depth=False,
                    num_buckets=4, dtype=np.float32):
    """get_probability_map_slice, mvsnet/model.py:45-144.  prob (D,H,W), depth (H,W)."""
    P = np.asarray(prob, dtype=dtype)
    D = P.shape[0]
    dm = np.asarray(depth_map, dtype=dtype)
    start = dtype(depth_start)
    interval = dtype(depth_interval)
    if inverse_depth:
        end = start + (dtype(D) - dtype(1)) * interval                # :84-85
        inv_s = dtype(1) / start
        inv_e = dtype(1) / end
        inv_int = (inv_s - inv_e) / (dtype(D) - dtype(1))             # :92-93
        idx = (dtype(1) / dm - inv_e) / inv_int                       # :94-95
        l0 = D - np.ceil(idx).astype(np.int64) - 1                    # :98
        l0 = np.clip(l0, 0, D - 1)
        r0 = D - np.floor(idx).astype(np.int64) - 1                   # :100-101
        r0 = np.clip(r0, 0, D - 1)
        l1 = np.clip(l0 - 1, 0, D - 1)                                # :104-105
        r1 = np.clip(r0 + 1, 0, D - 1)                                # :106-107
    else:
        idx = (dm - start) / interval                                 # :111-112
        l0 = np.clip(np.floor(idx).astype(np.int64), 0, D - 1)        # :113-114
        l1 = np.clip(l0 - 1, 0, D - 1)                                # :115-116
        r0 = np.clip(np.ceil(idx).astype(np.int64), 0, D - 1)         # :117-118
        r1 = np.clip(r0 + 1, 0, D - 1)                                # :119-120
    yy, xx = np.meshgrid(np.arange(P.shape[1]), np.arange(P.shape[2]), indexing="ij")
    out = P[l0, yy, xx] + P[r0, yy, xx]                               # :128-130
    if num_buckets == 4:
        out = out + (P[l1, yy, xx] + P[r1, yy, xx])                   # :138-140
    return out.astype(dtype)


def softargmin_and_prob(reg, depth_num, depth_start, depth_interval, inverse_depth=False,
                        dtype=np.float32):
    """mvsnet/model.py:471-498 composed: reg (D,H,W) -> depth (H,W), prob (H,W)."""
    P = softmax_neg(reg, dtype)
    depth = soft_argmin(P, depth_num, depth_start, depth_interval, inverse_depth, dtype)
    prob = probability_map(P, depth, depth_start, depth_interval, inverse_depth, 4, dtype)
    return depth, prob


# --------------------------------------------------------------------------------------
# R8 / R9  ConvGRU cell, winner-take-all sweep
# --------------------------------------------------------------------------------------


def layer_norm(x, gamma, beta, eps=1e-12, dtype=np.float32):
    """tf.contrib.layers.layer_norm on one sample (convgru.py:30-31): moments over
    (H,W,C), eps 1e-12, per-channel gamma/beta."""
    x = np.asarray(x, dtype=dtype)
    x64 = x.astype(np.float64)
    mean = x64.mean()
    var = x64.var()
    inv = np.asarray(gamma, np.float64) / np.sqrt(var + eps)
    return (x * inv.astype(dtype) + (np.asarray(beta, np.float64) - mean * inv).astype(dtype)).astype(dtype)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def conv_gru_cell(x, h, p, dtype=np.float32):
    """ConvGRUCell.__call__, mvsnet/convgru.py:82-122.  x (H,W,Cin), h (H,W,F).
    p: gates_w (3,3,Cin+F,2F), gates_b (2F), reset_gamma/beta (F), update_gamma/beta (F),
       out_w (3,3,Cin+F,F), out_b (F), out_gamma/beta (F)."""
    x = np.asarray(x, dtype=dtype)
    h = np.asarray(h, dtype=dtype)
    F = h.shape[-1]
    inputs = np.concatenate([x, h], axis=-1)                          # :89
    conv = conv2d_same(inputs, p["gates_w"], 1, p["gates_b"], dtype)  # :92-93
    r, u = conv[..., :F], conv[..., F:]                               # :94
    r = layer_norm(r, p["reset_gamma"], p["reset_beta"], dtype=dtype)    # :97
    u = layer_norm(u, p["update_gamma"], p["update_beta"], dtype=dtype)  # :98
    r = _sigmoid(r).astype(dtype)                                     # :101
    u = _sigmoid(u).astype(dtype)                                     # :102
    inputs = np.concatenate([x, r * h], axis=-1)                      # :107
    conv = conv2d_same(inputs, p["out_w"], 1, p["out_b"], dtype)      # :110-111
    conv = layer_norm(conv, p["out_gamma"], p["out_beta"], dtype=dtype)  # :114
    y = np.tanh(conv).astype(dtype)                                   # :117
    out = u * h + (dtype(1) - u) * y                                  # :120
    return out.astype(dtype)


def winner_take_all(ref_feature, src_features, homographies, depths, gru_params,
                    view_num=None, dtype=np.float32):
    """Loop body + tail of inference_winner_take_all, mvsnet/model.py:676-751.

    depths[d] is the depth value of plane d (model.py:706-715).  gru_params =
    {'gru1','gru2','gru3': cell params, 'prob_w' (3,3,F3,1), 'prob_b' (1,)}.
    Returns depth_image (H,W), prob (H,W)."""
    ref = np.asarray(ref_feature, dtype=dtype)
    src = np.asarray(src_features, dtype=dtype)
    Hs = np.asarray(homographies, dtype=dtype)
    n_src, D = Hs.shape[:2]
    if view_num is None:
        view_num = n_src + 1
    Hh, W, _ = ref.shape
    f1 = gru_params["gru1"]["out_b"].shape[0]
    f2 = gru_params["gru2"]["out_b"].shape[0]
    f3 = gru_params["gru3"]["out_b"].shape[0]
    s1 = np.zeros((Hh, W, f1), dtype); s2 = np.zeros((Hh, W, f2), dtype)
    s3 = np.zeros((Hh, W, f3), dtype)                                 # :649-654
    exp_sum = np.zeros((Hh, W), dtype)
    depth_image = np.zeros((Hh, W), dtype)
    max_prob = np.zeros((Hh, W), dtype)                               # :663-673
    for d in range(D):
        warped = [tf_transform_homography(src[v], Hs[v, d], dtype) for v in range(n_src)]
        cost = variance_cost_eager(ref, warped, view_num, dtype)      # :680-693
        s1 = conv_gru_cell(-cost, s1, gru_params["gru1"], dtype)      # :698
        s2 = conv_gru_cell(s1, s2, gru_params["gru2"], dtype)         # :699
        s3 = conv_gru_cell(s2, s3, gru_params["gru3"], dtype)         # :700
        reg = conv2d_same(s3, gru_params["prob_w"], 1, gru_params["prob_b"], dtype)[..., 0]  # :701-702
        prob = np.exp(reg).astype(dtype)                              # :703
        upd = max_prob < prob                                         # :721-722 (strict <)
        max_prob = np.where(upd, prob, max_prob)
        depth_image = np.where(upd, dtype(depths[d]), depth_image)
        exp_sum = exp_sum + prob                                      # :731
    return depth_image.astype(dtype), (max_prob / (exp_sum + dtype(1e-7))).astype(dtype)  # :749-751


def wta_depths(depth_num, depth_start, depth_end, inverse_depth=False, dtype=np.float32):
    """Depth value per plane inside the WTA loop, mvsnet/model.py:605-607,706-715."""
    D = int(depth_num)
    d_idx = np.arange(D, dtype=dtype)
    start, end = dtype(depth_start), dtype(depth_end)
    if inverse_depth:
        inv_s = dtype(1) / start
        inv_e = dtype(1) / end
        inv_interval = (inv_s - inv_e) / (dtype(D) - dtype(1))        # :710-711
        return (dtype(1) / (inv_s - d_idx * inv_interval)).astype(dtype)   # :712-713
    interval = (end - start) / (dtype(D) - dtype(1))                  # :606-607
    return (start + d_idx * interval).astype(dtype)                   # :715


# --------------------------------------------------------------------------------------
# R11  2D feature extractor UNetDS2GN (boundary input producer; PyTorch in the product)
# --------------------------------------------------------------------------------------


def group_norm_nhwc(x, gamma, beta, group_channel=8, eps=1e-5, dtype=np.float32):
    """conv_gn / deconv_gn normalisation, network.py:239-273: G = max(1, C // 8) groups
    (python-2 integer division at :247), moments over (C//G, H, W), eps 1e-5."""
    x = np.asarray(x, dtype=dtype)
    Hh, W, C = x.shape
    G = max(1, C // group_channel)
    xg = x.astype(np.float64).reshape(Hh, W, G, C // G)
    mean = xg.mean(axis=(0, 1, 3), keepdims=True)
    var = xg.var(axis=(0, 1, 3), keepdims=True)
    y = ((xg - mean) / np.sqrt(var + eps)).reshape(Hh, W, C)
    return (y * np.asarray(gamma, np.float64) + np.asarray(beta, np.float64)).astype(dtype)


UNET_LAYERS = (
    # name, kind, source(s), kernel, cout_mult, stride ; kinds: cg conv_gn(+relu), dg deconv_gn (no relu), c plain conv
    ("2dconv1_0", "cg", ("data",), 3, 2, 2), ("2dconv2_0", "cg", ("2dconv1_0",), 3, 4, 2),
    ("2dconv3_0", "cg", ("2dconv2_0",), 3, 8, 2), ("2dconv4_0", "cg", ("2dconv3_0",), 3, 16, 2),
    ("2dconv0_1", "cg", ("data",), 3, 1, 1), ("2dconv0_2", "cg", ("2dconv0_1",), 3, 1, 1),
    ("2dconv1_1", "cg", ("2dconv1_0",), 3, 2, 1), ("2dconv1_2", "cg", ("2dconv1_1",), 3, 2, 1),
    ("2dconv2_1", "cg", ("2dconv2_0",), 3, 4, 1), ("2dconv2_2", "cg", ("2dconv2_1",), 3, 4, 1),
    ("2dconv3_1", "cg", ("2dconv3_0",), 3, 8, 1), ("2dconv3_2", "cg", ("2dconv3_1",), 3, 8, 1),
    ("2dconv4_1", "cg", ("2dconv4_0",), 3, 16, 1), ("2dconv4_2", "cg", ("2dconv4_1",), 3, 16, 1),
    ("2dconv5_0", "dg", ("2dconv4_2",), 3, 8, 2),
    ("2dconv5_1", "cg", ("2dconv5_0", "2dconv3_2"), 3, 8, 1), ("2dconv5_2", "cg", ("2dconv5_1",), 3, 8, 1),
    ("2dconv6_0", "dg", ("2dconv5_2",), 3, 4, 2),
    ("2dconv6_1", "cg", ("2dconv6_0", "2dconv2_2"), 3, 4, 1), ("2dconv6_2", "cg", ("2dconv6_1",), 3, 4, 1),
    ("2dconv7_0", "dg", ("2dconv6_2",), 3, 2, 2),
    ("2dconv7_1", "cg", ("2dconv7_0", "2dconv1_2"), 3, 2, 1), ("2dconv7_2", "cg", ("2dconv7_1",), 3, 2, 1),
    ("2dconv8_0", "dg", ("2dconv7_2",), 3, 1, 2),
    ("2dconv8_1", "cg", ("2dconv8_0", "2dconv0_2"), 3, 1, 1), ("2dconv8_2", "cg", ("2dconv8_1",), 3, 1, 1),
    ("conv9_0", "cg", ("2dconv8_2",), 5, 2, 2), ("conv9_1", "cg", ("conv9_0",), 3, 2, 1),
    ("conv9_2", "cg", ("conv9_1",), 3, 2, 1),
    ("conv10_0", "cg", ("conv9_2",), 5, 4, 2), ("conv10_1", "cg", ("conv10_0",), 3, 4, 1),
    ("conv10_2", "c", ("conv10_1",), 3, 4, 1),
)


def unet_ds2gn(image, params, dtype=np.float32):
    """UNetDS2GN, mvsnet/cnn_wrapper/mvsnetworks.py:53-115.  image (H,W,3) ->
    (H/4, W/4, 4*base).  params[name] = {'w', 'gamma', 'beta'} ('w' only for conv10_2)."""
    layers = {"data": np.asarray(image, dtype=dtype)}
    for name, kind, srcs, k, _mult, stride in UNET_LAYERS:
        x = layers[srcs[0]] if len(srcs) == 1 else np.concatenate([layers[s] for s in srcs], -1)
        p = params[name]
        if kind == "dg":
            y = convnd_transpose_same(x, p["w"], stride, dtype)
            y = group_norm_nhwc(y, p["gamma"], p["beta"], dtype=dtype)    # no ReLU (network.py:357)
        else:
            y = convnd_same(x, p["w"], stride, dtype)
            if kind == "cg":
                y = np.maximum(group_norm_nhwc(y, p["gamma"], p["beta"], dtype=dtype), dtype(0))
        layers[name] = y
    return layers["conv10_2"]


def standardise_image(image, dtype=np.float32):
    """The towers' input (mvs_data_generation/utils.py:33-38, applied per image by the cluster generators): every channel of ONE
    decoded image minus its mean over the image, over (its biased standard deviation + 1e-8).  dtype float32 follows the
    reference to the letter (numpy's float32 reductions -- over the two leading axes numpy keeps RUNNING float32 sums, so at
    640 x 512 and above the reference's own moments are only good to ~1e-3 relative and depend on the numpy build); float64
    gives the exact moments the HIP path forms from integer sums (the two agree to ~2e-6 on a 64 x 80 image)."""
    x = np.asarray(image).astype(dtype)
    spread = np.sqrt(np.var(x, axis=(0, 1), keepdims=True))
    centre = np.mean(x, axis=(0, 1), keepdims=True)
    return (x - centre) / (spread + dtype(0.00000001))


# --------------------------------------------------------------------------------------
# 8f row f3  depth refinement (model.py:753-811, mvsnetworks.py:178-193,261-324)
# --------------------------------------------------------------------------------------

# (name, kind 'c'|'d', sources, out-channel multiple of the base filter (0 = one channel), stride, relu)
REFINE_ORIGINAL = tuple(("refine_conv%d" % i, "c", ("concat_image",) if i == 0 else ("refine_conv%d" % (i - 1),),
                         1 if i < 3 else 0, 1, i < 3) for i in range(4))


def _refine_unet_table():
    n = lambda s: "2dconv%s_refine" % s
    t = []
    prev = "concat_image"
    for lvl, mult in ((1, 2), (2, 4), (3, 8), (4, 16)):            # strided encoder chain (:272-276)
        t.append((n("%d_0" % lvl), "c", (prev,), mult, 2, True)); prev = n("%d_0" % lvl)
    for lvl, mult, src in ((0, 1, "concat_image"), (1, 2, n("1_0")), (2, 4, n("2_0")), (3, 8, n("3_0")), (4, 16, n("4_0"))):
        t.append((n("%d_1" % lvl), "c", (src,), mult, 1, True))    # same-resolution pairs (:278-296)
        t.append((n("%d_2" % lvl), "c", (n("%d_1" % lvl),), mult, 1, True))
    for up, skip, mult in ((5, 3, 8), (6, 2, 4), (7, 1, 2), (8, 0, 1)):   # decoder (:297-320)
        below = n("4_2") if up == 5 else n("%d_2" % (up - 1))
        t.append((n("%d_0" % up), "d", (below,), mult, 2, True))
        t.append((n("%d_1" % up), "c", (n("%d_0" % up), n("%d_2" % skip)), mult, 1, True))
        t.append((n("%d_2" % up), "c", (n("%d_1" % up),), mult, 1, True))
    t.append((n("8_3"), "c", (n("8_2"),), 4, 1, True))              # (:322-324)
    t.append((n("8_4"), "c", (n("8_3"),), 0, 1, False))
    return tuple(t)


REFINE_UNET = _refine_unet_table()


def resize_bilinear_tf1(x, out_h, out_w, dtype=np.float32):
    """tf.image.resize_bilinear(x, [out_h, out_w]) with TF 1.x defaults (align_corners=False):
    src = dst * (in / out), lower index floor(src), upper index min(lower + 1, in - 1)
    (model.py:768-781).  x (H,W,C)."""
    x = np.asarray(x, dtype=dtype)
    h, w = x.shape[:2]

    def axis(n_in, n_out):
        src = (np.arange(n_out, dtype=dtype) * dtype(n_in / float(n_out))).astype(dtype)
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        return i0, np.minimum(i0 + 1, n_in - 1), (src - i0.astype(dtype)).astype(dtype)

    y0, y1, fy = axis(h, out_h)
    x0, x1, fx = axis(w, out_w)
    fx = fx[None, :, None]; fy = fy[:, None, None]
    top = x[y0][:, x0] * (1 - fx) + x[y0][:, x1] * fx
    bot = x[y1][:, x0] * (1 - fx) + x[y1][:, x1] * fx
    return (top * (1 - fy) + bot * fy).astype(dtype)


def refine_net(color_image, depth_image, params, network_type="original", dtype=np.float32):
    """RefineNetConv / RefineUNetConv: Network.conv / deconv with their defaults -- bias, ReLU except
    on the last layer, SAME (network.py:171-215,300-329).  Inputs (H,W,3), (H,W,1|2) -> (H,W,1)."""
    table = REFINE_ORIGINAL if network_type == "original" else REFINE_UNET
    layers = {"concat_image": np.concatenate([np.asarray(color_image, dtype), np.asarray(depth_image, dtype)], -1)}
    for name, kind, srcs, _mult, stride, relu in table:
        x = layers[srcs[0]] if len(srcs) == 1 else np.concatenate([layers[s] for s in srcs], -1)
        p = params[name]
        y = convnd_transpose_same(x, p["w"], stride, dtype) if kind == "d" else convnd_same(x, p["w"], stride, dtype)
        y = y + np.asarray(p["b"], dtype)
        layers[name] = np.maximum(y, dtype(0)) if relu else y
    return layers[table[-1][0]]


def depth_refine(init_depth_map, image, prob_map, depth_num, depth_start, depth_interval, params,
                 network_type="original", upsample_depth=False, refine_with_confidence=False,
                 residual_refinement=True, dtype=np.float32, stereo_image=None):
    """model.py:753-811 for one sample: init_depth_map, prob_map (h,w,1); image (H,W,3); stereo_image (H,W,3) is the
    optional stereo partner concatenated after the confidence (:777-789)."""
    d = np.asarray(init_depth_map, dtype)
    scale = dtype((depth_start + (float(depth_num) - 1.0) * depth_interval) - depth_start)
    norm = (d - dtype(depth_start)) / scale
    image = np.asarray(image, dtype); prob_map = np.asarray(prob_map, dtype)
    if upsample_depth:
        H, W = image.shape[:2]
        norm, d = resize_bilinear_tf1(norm, H, W, dtype), resize_bilinear_tf1(d, H, W, dtype)
        if refine_with_confidence:
            prob_map = resize_bilinear_tf1(prob_map, H, W, dtype)
    else:
        image = resize_bilinear_tf1(image, d.shape[0], d.shape[1], dtype)
        if stereo_image is not None:
            stereo_image = resize_bilinear_tf1(np.asarray(stereo_image, dtype), d.shape[0], d.shape[1], dtype)
    data = np.concatenate([norm, prob_map], -1) if refine_with_confidence else norm
    if stereo_image is not None:
        data = np.concatenate([data, np.asarray(stereo_image, dtype)], -1)
    residual = refine_net(image, data, params, network_type, dtype) * scale
    return (residual + d if residual_refinement else residual), residual


# --------------------------------------------------------------------------------------
# benchmark metrics of mvsnet/test.py (loss.py:15-28,134-220), forward only
# --------------------------------------------------------------------------------------

def regression_metrics(estimated, gt, depth_start, depth_end, grad_loss=True):
    """mvsnet_regression_loss with loss_type='original' (loss.py:189-220) for (B,H,W,1) arrays in
    float64: (loss, less_one, less_three, debug).  interval = (end - start) / 191; invalid GT = 0.
    gradient_loss (loss.py:134-158) is written on 2-D maps but is handed the 4-D batch, so its
    "vertical" term differences the BATCH axis and its "horizontal" term the image rows."""
    y, p = np.asarray(gt, np.float64), np.asarray(estimated, np.float64)
    interval = (np.asarray(depth_end, np.float64).reshape(-1) - np.asarray(depth_start, np.float64).reshape(-1)) / 191.0
    m = (y != 0).astype(np.float64)
    denom = np.abs(m.sum(axis=(1, 2, 3))) + 1e-6
    loss = (((m * (y - p)).__abs__().sum(axis=(1, 2, 3)) / interval) / denom).sum()
    debug = None
    if grad_loss:
        diff = y - p
        v = np.abs((diff[0:-2] - diff[2:]) * (m[0:-2] * m[2:]))
        h = np.abs((diff[:, 0:-2] - diff[:, 2:]) * (m[:, 0:-2] * m[:, 2:]))
        debug = (np.log(1.0 + v).sum() + np.log(1.0 + h).sum()) / m.sum()
        loss = loss + 0.5 * debug
    rel = np.abs(y - p) / interval.reshape(-1, 1, 1, 1)
    total = np.abs(m.sum()) + 1e-6
    return loss, (m * (rel <= 1.0)).sum() / total, (m * (rel <= 3.0)).sum() / total, debug


# --------------------------------------------------------------------------------------
# R10  composition
# --------------------------------------------------------------------------------------


def inference_mem_from_features(features, cams, depth_num, depth_start, depth_interval,
                                regnet_params, inverse_depth=False, dtype=np.float32):
    """mvsnet/model.py:374-502 after the feature towers.  features (N,H,W,C), cams
    (N,2,4,4) -> depth (H,W), prob (H,W)."""
    feats = np.asarray(features, dtype=dtype)
    N = feats.shape[0]
    D = int(depth_num)
    end = dtype(depth_start) + (dtype(D) - dtype(1)) * dtype(depth_interval)
    Hs = []
    for v in range(1, N):
        if inverse_depth:
            Hs.append(get_homographies_inv_depth(cams[0], cams[v], D, depth_start, end, dtype))
        else:
            Hs.append(get_homographies(cams[0], cams[v], D, depth_start, depth_interval, dtype))
    cost = cost_volume(feats[0], feats[1:], np.stack(Hs), N, "mem", dtype)
    reg = regnet_us0(cost, regnet_params, dtype)
    return softargmin_and_prob(reg, D, depth_start, depth_interval, inverse_depth, dtype)


def inference_winner_take_all_from_features(features, cams, depth_num, depth_start, depth_end,
                                            gru_params, inverse_depth=False, dtype=np.float32):
    """mvsnet/model.py:601-751 after the feature towers."""
    feats = np.asarray(features, dtype=dtype)
    N = feats.shape[0]
    D = int(depth_num)
    Hs = []
    for v in range(1, N):
        if inverse_depth:
            Hs.append(get_homographies_inv_depth(cams[0], cams[v], D, depth_start, depth_end, dtype))
        else:
            interval = (dtype(depth_end) - dtype(depth_start)) / (dtype(D) - dtype(1))
            Hs.append(get_homographies(cams[0], cams[v], D, depth_start, interval, dtype))
    depths = wta_depths(D, depth_start, depth_end, inverse_depth, dtype)
    return winner_take_all(feats[0], feats[1:], np.stack(Hs), depths, gru_params, N, dtype)


# --------------------------------------------------------------------------------------
# Output formats (byte contract), mvsnet/preprocess.py:294-356
# --------------------------------------------------------------------------------------


def pfm_bytes(image):
    """Bytes write_pfm would emit (preprocess.py:327-356) for a float32 little-endian
    greyscale/colour image: header 'Pf\\n' / 'PF\\n', '%d %d\\n' % (W, H), '%f\\n' % -1.0,
    rows bottom-to-top."""
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    image = np.flipud(image)
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    head = ("PF\n" if color else "Pf\n") + "%d %d\n" % (image.shape[1], image.shape[0]) + "%f\n" % -1.0
    return head.encode("ascii") + np.ascontiguousarray(image).astype("<f4").tobytes()
