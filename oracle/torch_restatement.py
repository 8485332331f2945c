"""Multi-threaded CPU restatement (torch-CPU) of the features->depth path.

TEST / MEASUREMENT INFRASTRUCTURE: never imported by ``mvsnet_amd``.  Two uses:
  * fp32 (default): the timed ``cpu_baseline`` of bench.py ("kind": "port");
  * fp64 (``dtype=torch.float64``): the generator of the FULL-SIZE golden fixtures
    (tests/golden/make_golden.py --full: workloads M, c1, c2, c3), because the strict numpy oracle
    (mvsnet_oracle.py, python loops over planes and taps) needs hours at those sizes.  In fp64
    this module is the same function as ``mvsnet_oracle`` with ``dtype=np.float64`` — homographies
    come from the numpy oracle itself — and tests/test_cpu_restatement.py holds the two together to
    ~1e-9 at the sizes the numpy oracle finishes in seconds.
Parity unpinned by the reference (see mvsnet_oracle.py header).

The TensorFlow reference cannot run here (python2 + TF 1.12 absent), so this is labelled
"CPU restatement", not "TensorFlow reference" (BASELINE.md section 3).  Citations as in
mvsnet_oracle.py: warp = homography_warping.py:211-253, variance = model.py:436-462 (3D-CNN) and
:680-693 (recurrent), RegNetUS0 = mvsnetworks.py:122-158 + network.py:278-348,492-509,
soft-argmin = model.py:471-498, ConvGRU = convgru.py:82-122, winner-take-all = model.py:676-751.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mvsnet_oracle as O

_NP = {torch.float32: np.float32, torch.float64: np.float64}
# torch's CPU convolutions have no blocked fp64 kernel: they unfold the input into a
# (27*Cin) x voxels matrix first.  Convolve in slabs along depth so that matrix stays below this.
_COL_BYTES = 2 << 30


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, _NP[dtype])))


def _warp_planes(src, t8):
    """src (C,H,W) tensor; t8 (D,8) -> (D,C,H,W): tf.contrib.image.transform BILINEAR, zero fill
    per tap == grid_sample(bilinear, zeros, align_corners=True) at the same sample points."""
    C, H, W = src.shape
    D = t8.shape[0]
    dt = src.dtype
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    t = t8.view(D, 8, 1, 1)
    proj = t[:, 6] * xs + t[:, 7] * ys + 1.0
    sx = (t[:, 0] * xs + t[:, 1] * ys + t[:, 2]) / proj
    sy = (t[:, 3] * xs + t[:, 4] * ys + t[:, 5]) / proj
    gx = sx * (2.0 / max(W - 1, 1)) - 1.0
    gy = sy * (2.0 / max(H - 1, 1)) - 1.0
    grid = torch.stack([gx, gy], dim=-1)                     # (D,H,W,2)
    return F.grid_sample(src[None].expand(D, C, H, W), grid, mode="bilinear", padding_mode="zeros",
                         align_corners=True)


def cost_volume(features, transforms, view_num, dtype=torch.float32, variant="mem", planes=None):
    """features (N,H,W,C) numpy, transforms (N-1,D,8) numpy -> (C,D,H,W) tensor.
    variant 'mem': Q/N - S*S/(N*N) (model.py:458-461); 'eager': Q/N - (S/N)^2 (model.py:330-332,
    :690-693).  `planes` = (d0, d1) restricts the output to that plane range."""
    f = _t(features, dtype).permute(0, 3, 1, 2).contiguous()
    t8 = _t(transforms, dtype)
    d0, d1 = (0, t8.shape[1]) if planes is None else planes
    n = float(view_num)
    out = []
    step = 16                                                  # planes per pass: bounds the temporaries
    for a in range(d0, d1, step):
        b = min(a + step, d1)
        ref = f[0][None]
        S = ref.expand(b - a, -1, -1, -1).clone()
        Q = (ref * ref).expand(b - a, -1, -1, -1).clone()
        for v in range(1, f.shape[0]):
            w = _warp_planes(f[v], t8[v - 1, a:b])
            S += w
            Q += w * w
        if variant == "mem":
            out.append(Q / n - (S * S) / (n * n))
        else:
            S = S / n
            out.append(Q / n - S * S)
    return torch.cat(out, 0).permute(1, 0, 2, 3).contiguous()  # (C,D,H,W)


def _wt(w, dtype):
    return _t(w, dtype).permute(4, 3, 0, 1, 2).contiguous()      # TF (kd,kh,kw,a,b) -> torch (b,a,kd,kh,kw)


def _conv(x, w, stride):
    """tf.layers.conv3d(padding='SAME', use_bias=False) on x (1,Cin,D,H,W); explicit TF padding
    (stride 2, even size: 0 before / 1 after -- mvsnet_oracle.same_pad)."""
    wt = _wt(w, x.dtype)
    pads = []
    for n in reversed(x.shape[-3:]):                          # F.pad order: W, H, D
        _, pb, pa = O.same_pad(int(n), 3, stride)
        pads += [pb, pa]
    xp = F.pad(x, pads)
    Dout = (xp.shape[2] - 3) // stride + 1
    plane_cols = 27 * x.shape[1] * ((xp.shape[3] - 3) // stride + 1) * ((xp.shape[4] - 3) // stride + 1) * x.element_size()
    if x.dtype == torch.float32 or plane_cols * Dout <= _COL_BYTES:
        return F.conv3d(xp, wt, stride=stride)
    step = max(1, _COL_BYTES // plane_cols)
    return torch.cat([F.conv3d(xp[:, :, o * stride: (min(o + step, Dout) - 1) * stride + 3], wt, stride=stride)
                      for o in range(0, Dout, step)], 2)


def _deconv(x, w):
    """tf.layers.conv3d_transpose(stride 2, 'SAME'): the full transposed output cropped at the end."""
    wt = _wt(w, x.dtype)
    y = F.conv_transpose3d(x, wt, stride=2)
    D, H, W = x.shape[-3:]
    return y[..., : 2 * D, : 2 * H, : 2 * W]


def _bn_relu(x, p, eps=1e-5):
    g, b = _t(p["gamma"], x.dtype), _t(p["beta"], x.dtype)
    return F.relu(F.batch_norm(x, None, None, g, b, training=True, eps=eps))


def regnet_us0(cost, params):
    """cost (C,D,H,W) tensor -> (D,H,W) tensor (mvsnetworks.py:122-158)."""
    x = cost[None]
    cb = lambda t, n, s: _bn_relu(_conv(t, params[n]["w"], s), params[n])
    db = lambda t, n: _bn_relu(_deconv(t, params[n]["w"]), params[n])
    c1_0 = cb(x, "3dconv1_0", 2); c2_0 = cb(c1_0, "3dconv2_0", 2); c3_0 = cb(c2_0, "3dconv3_0", 2)
    c0_1 = cb(x, "3dconv0_1", 1); c1_1 = cb(c1_0, "3dconv1_1", 1); c2_1 = cb(c2_0, "3dconv2_1", 1)
    c3_1 = cb(c3_0, "3dconv3_1", 1)
    del x, c1_0, c2_0, c3_0
    c4 = db(c3_1, "3dconv4_0") + c2_1
    c5 = db(c4, "3dconv5_0") + c1_1
    c6 = db(c5, "3dconv6_0") + c0_1
    return _conv(c6, params["3dconv6_2"]["w"], 1)[0, 0]


def softargmin_prob(reg, depth_start, depth_interval):
    D = reg.shape[0]
    P = torch.softmax(-reg, dim=0)
    z = _t(O.depth_values(D, depth_start, depth_interval, False, _NP[reg.dtype]), reg.dtype)
    depth = (P * z[:, None, None]).sum(0)
    idx = (depth - float(depth_start)) / float(depth_interval)
    l0 = idx.floor().long().clamp(0, D - 1); r0 = idx.ceil().long().clamp(0, D - 1)
    l1 = (l0 - 1).clamp(0, D - 1); r1 = (r0 + 1).clamp(0, D - 1)
    g = lambda i: torch.gather(P, 0, i[None])[0]
    return depth, g(l0) + g(r0) + g(l1) + g(r1)


def _transforms(cams, depth_num, depth_start, depth_interval, npdt, inverse_depth=False):
    """fp32: the reference's fp32 homography algebra; fp64: the strict oracle's.  inverse_depth: R1'
    (homography_warping.py:60-106) with end = start + (D-1)*interval as model.py:378-379,439-441 pass it."""
    N = cams.shape[0]
    if inverse_depth:
        end = npdt(depth_start) + (npdt(depth_num) - npdt(1)) * npdt(depth_interval)
        Hs = np.stack([O.get_homographies_inv_depth(cams[0], cams[v], depth_num, depth_start, end, npdt)
                       for v in range(1, N)])
    else:
        Hs = np.stack([O.get_homographies(cams[0], cams[v], depth_num, depth_start, depth_interval, npdt)
                       for v in range(1, N)])
    return O.homography_to_transform8(Hs, npdt)


@torch.no_grad()
def inference_mem_from_features(features, cams, depth_num, depth_start, depth_interval, regnet_params,
                                dtype=torch.float32, inverse_depth=False):
    """Same contract as mvsnet_oracle.inference_mem_from_features, all host cores.  With inverse_depth the
    soft-argmin / four-bucket tail (model.py:480-485,83-107) is the strict numpy oracle's own (vectorised, seconds).
    Returns numpy depth (H,W), prob (H,W) in `dtype`."""
    N = features.shape[0]
    T = _transforms(cams, depth_num, depth_start, depth_interval, _NP[dtype], inverse_depth)
    cost = cost_volume(features, T, N, dtype)
    reg = regnet_us0(cost, regnet_params)
    if inverse_depth:
        return O.softargmin_and_prob(reg.numpy(), int(depth_num), depth_start, depth_interval, True, _NP[dtype])
    depth, prob = softargmin_prob(reg, depth_start, depth_interval)
    return depth.numpy(), prob.numpy()


# ---- recurrent regulariser (R-MVSNet) ------------------------------------------------------------

def _conv2d_same(x, w, b):
    """x (1,C,H,W), w TF layout (3,3,Cin,Cout): tf.layers.conv2d(padding='same'), with bias."""
    return F.conv2d(x, w, b, padding=1)


def _layer_norm(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm on one sample (convgru.py:30-31): moments over (C,H,W) in
    float64 as mvsnet_oracle.layer_norm takes them, eps 1e-12."""
    x64 = x.double()
    mean = x64.mean()
    var = ((x64 - mean) ** 2).mean()
    inv = gamma.double() / torch.sqrt(var + eps)
    return x * inv.to(x.dtype).view(1, -1, 1, 1) + (beta.double() - mean * inv).to(x.dtype).view(1, -1, 1, 1)


def _cell_params(p, dtype):
    q = {k: _t(v, dtype) for k, v in p.items()}
    q["gates_w"] = q["gates_w"].permute(3, 2, 0, 1).contiguous()
    q["out_w"] = q["out_w"].permute(3, 2, 0, 1).contiguous()
    return q


def conv_gru_cell(x, h, p):
    """ConvGRUCell.__call__ (convgru.py:82-122); p from _cell_params."""
    Fn = h.shape[1]
    g = _conv2d_same(torch.cat([x, h], 1), p["gates_w"], p["gates_b"])
    r = torch.sigmoid(_layer_norm(g[:, :Fn], p["reset_gamma"], p["reset_beta"]))
    u = torch.sigmoid(_layer_norm(g[:, Fn:], p["update_gamma"], p["update_beta"]))
    c = _conv2d_same(torch.cat([x, r * h], 1), p["out_w"], p["out_b"])
    y = torch.tanh(_layer_norm(c, p["out_gamma"], p["out_beta"]))
    return u * h + (1.0 - u) * y


@torch.no_grad()
def inference_winner_take_all_from_features(features, cams, depth_num, depth_start, depth_end, gru_params,
                                            dtype=torch.float32, progress=None, inverse_depth=False):
    """Same contract as mvsnet_oracle.inference_winner_take_all_from_features (inverse_depth: planes uniform in
    1/depth, homography_warping.py:60-106, model.py:706-713).
    Returns numpy depth (H,W), prob (H,W) and the winning plane index (H,W) int32."""
    npdt = _NP[dtype]
    N, Hh, W, _ = features.shape
    D = int(depth_num)
    if inverse_depth:
        Hs = np.stack([O.get_homographies_inv_depth(cams[0], cams[v], D, depth_start, depth_end, npdt) for v in range(1, N)])
        T = O.homography_to_transform8(Hs, npdt)
    else:
        interval = (npdt(depth_end) - npdt(depth_start)) / (npdt(D) - npdt(1))      # model.py:605-607
        T = _transforms(cams, D, depth_start, interval, npdt)
    depths = O.wta_depths(D, depth_start, depth_end, bool(inverse_depth), npdt)
    cells = [_cell_params(gru_params[k], dtype) for k in ("gru1", "gru2", "gru3")]
    pw = _t(gru_params["prob_w"], dtype).permute(3, 2, 0, 1).contiguous()
    pb = _t(gru_params["prob_b"], dtype)
    s = [torch.zeros((1, int(c["out_b"].shape[0]), Hh, W), dtype=dtype) for c in cells]   # model.py:649-654
    exp_sum = torch.zeros((Hh, W), dtype=dtype)
    max_prob = torch.zeros((Hh, W), dtype=dtype)
    depth_image = torch.zeros((Hh, W), dtype=dtype)
    index = torch.full((Hh, W), -1, dtype=torch.int32)
    batch = 16
    for d0 in range(0, D, batch):
        cost = cost_volume(features, T, N, dtype, "eager", (d0, min(d0 + batch, D)))      # (C,b,H,W)
        for j in range(cost.shape[1]):
            d = d0 + j
            s[0] = conv_gru_cell(-cost[:, j][None], s[0], cells[0])                       # model.py:698
            s[1] = conv_gru_cell(s[0], s[1], cells[1])
            s[2] = conv_gru_cell(s[1], s[2], cells[2])
            prob = torch.exp(_conv2d_same(s[2], pw, pb)[0, 0])                            # :701-703
            upd = max_prob < prob                                                         # :721-722 (strict <)
            max_prob = torch.where(upd, prob, max_prob)
            depth_image = torch.where(upd, torch.full_like(prob, float(depths[d])), depth_image)
            index = torch.where(upd, torch.full_like(index, d), index)
            exp_sum = exp_sum + prob                                                      # :731
        if progress:
            progress(min(d0 + batch, D), D)
    return depth_image.numpy(), (max_prob / (exp_sum + 1e-7)).numpy(), index.numpy()      # :749-751
