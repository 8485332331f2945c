"""Gradient checker for the training path: the features -> depth composition of
mvsnet/model.py:257-372 (`inference`: eager variance :315-334, RegNetUS0 mvsnetworks.py:122-158 with
batch-statistics BatchNorm network.py:492-509, soft-argmin model.py:343-366) written with differentiable
torch-CPU float64 ops, so that torch autograd plays the role TensorFlow's autodiff has in the reference
(train.py:428-429).  TEST INFRASTRUCTURE ONLY: never imported by ``mvsnet_amd``.  Parity unpinned by the
reference (no tests / fixtures there; see mvsnet_oracle.py header); the forward value of this file is
checked against the strict numpy restatement in tests/test_cpu_restatement.py.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mvsnet_oracle as O


def warp_all_planes(src, t8):
    """src (C,H,W), t8 (D,8) -> (D,C,H,W); tf.contrib.image.transform BILINEAR with zero fill per tap
    (homography_warping.py:251-252) == grid_sample(bilinear, zeros, align_corners=True)."""
    C, H, W = src.shape
    D = t8.shape[0]
    dt = src.dtype
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    t = t8.view(D, 8, 1, 1)
    proj = t[:, 6] * xs + t[:, 7] * ys + 1.0
    sx = (t[:, 0] * xs + t[:, 1] * ys + t[:, 2]) / proj
    sy = (t[:, 3] * xs + t[:, 4] * ys + t[:, 5]) / proj
    grid = torch.stack([sx * (2.0 / max(W - 1, 1)) - 1.0, sy * (2.0 / max(H - 1, 1)) - 1.0], dim=-1)
    return F.grid_sample(src[None].expand(D, C, H, W), grid, mode="bilinear", padding_mode="zeros",
                         align_corners=True)


def cost_volume(features, t8):
    """features (N,H,W,C) tensor, t8 (N-1,D,8) tensor -> (C,D,H,W): cost = Q/N - (S/N)^2 (model.py:330-332)."""
    f = features.permute(0, 3, 1, 2)
    n = float(f.shape[0])
    S, Q = f[0][None], (f[0] * f[0])[None]
    for v in range(1, f.shape[0]):
        w = warp_all_planes(f[v], t8[v - 1])
        S = S + w
        Q = Q + w * w
    return (Q / n - (S / n) ** 2).permute(1, 0, 2, 3)


def _pad_same(x, stride):
    pads = []
    for n in reversed(x.shape[-3:]):
        _, pb, pa = O.same_pad(int(n), 3, stride)
        pads += [pb, pa]
    return F.pad(x, pads)


def _conv(x, w, stride):
    return F.conv3d(_pad_same(x, stride), w.permute(4, 3, 0, 1, 2), stride=stride)


def _deconv(x, w):
    y = F.conv_transpose3d(x, w.permute(4, 3, 0, 1, 2), stride=2)
    D, H, W = x.shape[-3:]
    return y[..., : 2 * D, : 2 * H, : 2 * W]


def _bn_relu(x, p, eps=1e-5):
    return F.relu(F.batch_norm(x, None, None, p["gamma"], p["beta"], training=True, eps=eps))


def regnet_us0(cost, p):
    """cost (C,D,H,W) -> (D,H,W), or a batch (B,C,D,H,W) -> (B,D,H,W) whose BatchNorm statistics run over the whole
    batch (what cross-replica BatchNorm computes with one sample per replica); p[name] = {'w','gamma','beta'} tensors
    in the TensorFlow layouts."""
    batched = cost.dim() == 5
    x = cost if batched else cost[None]
    cb = lambda t, n, s: _bn_relu(_conv(t, p[n]["w"], s), p[n])
    db = lambda t, n: _bn_relu(_deconv(t, p[n]["w"]), p[n])
    c1_0 = cb(x, "3dconv1_0", 2); c2_0 = cb(c1_0, "3dconv2_0", 2); c3_0 = cb(c2_0, "3dconv3_0", 2)
    c0_1 = cb(x, "3dconv0_1", 1); c1_1 = cb(c1_0, "3dconv1_1", 1); c2_1 = cb(c2_0, "3dconv2_1", 1)
    c3_1 = cb(c3_0, "3dconv3_1", 1)
    c4 = db(c3_1, "3dconv4_0") + c2_1
    c5 = db(c4, "3dconv5_0") + c1_1
    c6 = db(c5, "3dconv6_0") + c0_1
    out = _conv(c6, p["3dconv6_2"]["w"], 1)[:, 0]
    return out if batched else out[0]


def soft_argmin(reg, depth_start, depth_interval):
    D = reg.shape[0]
    P = torch.softmax(-reg, dim=0)
    z = torch.from_numpy(O.depth_values(D, depth_start, depth_interval, False, np.float64)).to(reg.dtype)
    return (P * z[:, None, None]).sum(0)


def depth_from_features(features, t8, depth_start, depth_interval, p):
    """features (N,H,W,C), t8 (N-1,D,8) -> depth (H,W), differentiable in features and p."""
    return soft_argmin(regnet_us0(cost_volume(features, t8), p), depth_start, depth_interval)


# ---- recurrent regulariser (model.py:505-599) and classification loss (loss.py:223-267) ------------------------

def _conv2d_same(x, w, b):
    """x (1,C,H,W), w TF layout (3,3,Cin,Cout): tf.layers.conv2d(padding='same')."""
    return F.conv2d(x, w.permute(3, 2, 0, 1), b, padding=1)


def _layer_norm(x, gamma, beta, eps=1e-12):
    """tf.contrib.layers.layer_norm on one sample (convgru.py:30-31): moments over (C,H,W)."""
    mean = x.mean()
    var = ((x - mean) ** 2).mean()
    return (x - mean) / torch.sqrt(var + eps) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def conv_gru_cell(x, h, p):
    """ConvGRUCell.__call__ (convgru.py:82-122) in its own shape: concatenate, convolve, split."""
    Fn = h.shape[1]
    g = _conv2d_same(torch.cat([x, h], 1), p["gates_w"], p["gates_b"])
    r = torch.sigmoid(_layer_norm(g[:, :Fn], p["reset_gamma"], p["reset_beta"]))
    u = torch.sigmoid(_layer_norm(g[:, Fn:], p["update_gamma"], p["update_beta"]))
    c = _conv2d_same(torch.cat([x, r * h], 1), p["out_w"], p["out_b"])
    y = torch.tanh(_layer_norm(c, p["out_gamma"], p["out_beta"]))
    return u * h + (1.0 - u) * y


def recurrent_reg(features, t8, gp):
    """features (N,H,W,C), t8 (N-1,D,8) -> regularised cost (D,H,W), plane by plane (model.py:563-589)."""
    cost = cost_volume(features, t8)                              # (C,D,H,W)
    _C, D, H, W = cost.shape
    f = [int(gp[k]["out_b"].shape[0]) for k in ("gru1", "gru2", "gru3")]
    s = [torch.zeros((1, n, H, W), dtype=cost.dtype) for n in f]
    planes = []
    for d in range(D):
        s[0] = conv_gru_cell(-cost[:, d][None], s[0], gp["gru1"])
        s[1] = conv_gru_cell(s[0], s[1], gp["gru2"])
        s[2] = conv_gru_cell(s[1], s[2], gp["gru3"])
        planes.append(_conv2d_same(s[2], gp["prob_w"], gp["prob_b"])[0, 0])
    return torch.stack(planes, 0)


def classification_loss(reg, gt, depth_start, depth_interval):
    """loss.py:223-247 on one sample: reg (D,H,W) -> softmax over planes -> masked cross entropy against the
    one-hot of round((gt - start) / interval); gt (H,W), zeros = invalid."""
    D = reg.shape[0]
    prob = torch.softmax(reg, 0)
    mask = (gt != 0).to(reg.dtype)
    index = torch.round(mask * (gt - depth_start) / depth_interval).to(torch.int64)
    onehot = torch.zeros_like(prob)
    inside = (index >= 0) & (index < D)
    onehot.scatter_(0, index.clamp(0, D - 1)[None], inside[None].to(reg.dtype))
    ce = -(onehot * torch.log(prob)).sum(0)
    return (mask * ce).sum() / (mask.sum() + 1e-7)
