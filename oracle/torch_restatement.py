"""Multi-threaded CPU restatement (torch-CPU, fp32) of the features->depth path, used ONLY as the
timed ``cpu_baseline`` of bench.py ("kind": "port") and validated against the strict numpy oracle
in tests/test_cpu_restatement.py.  TEST / MEASUREMENT INFRASTRUCTURE: never imported by
``mvsnet_amd``.  Parity unpinned by the reference (see mvsnet_oracle.py header).

The TensorFlow reference cannot run here (python2 + TF 1.12 absent), so this is labelled
"CPU restatement", not "TensorFlow reference" (BASELINE.md section 3).  Citations as in
mvsnet_oracle.py: warp = homography_warping.py:211-253, variance = model.py:436-462,
RegNetUS0 = mvsnetworks.py:122-158 + network.py:278-348,492-509, soft-argmin = model.py:471-498.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import mvsnet_oracle as O


def _warp_all_planes(src, t8):
    """src (C,H,W) tensor; t8 (D,8) -> (D,C,H,W): tf.contrib.image.transform BILINEAR, zero fill
    per tap == grid_sample(bilinear, zeros, align_corners=True) at the same sample points."""
    C, H, W = src.shape
    D = t8.shape[0]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing="ij")
    t = t8.view(D, 8, 1, 1)
    proj = t[:, 6] * xs + t[:, 7] * ys + 1.0
    sx = (t[:, 0] * xs + t[:, 1] * ys + t[:, 2]) / proj
    sy = (t[:, 3] * xs + t[:, 4] * ys + t[:, 5]) / proj
    gx = sx * (2.0 / max(W - 1, 1)) - 1.0
    gy = sy * (2.0 / max(H - 1, 1)) - 1.0
    grid = torch.stack([gx, gy], dim=-1)                     # (D,H,W,2)
    return F.grid_sample(src[None].expand(D, C, H, W), grid, mode="bilinear", padding_mode="zeros",
                         align_corners=True)


def cost_volume(features, transforms, view_num):
    """features (N,H,W,C) numpy, transforms (N-1,D,8) numpy -> (C,D,H,W) tensor."""
    f = torch.from_numpy(np.ascontiguousarray(features)).permute(0, 3, 1, 2).contiguous()
    t8 = torch.from_numpy(np.ascontiguousarray(transforms, dtype=np.float32))
    D = t8.shape[1]
    ref = f[0][None]                                          # (1,C,H,W)
    S = ref.expand(D, -1, -1, -1).clone()
    Q = (ref * ref).expand(D, -1, -1, -1).clone()
    for v in range(1, f.shape[0]):
        w = _warp_all_planes(f[v], t8[v - 1])
        S += w
        Q += w * w
    n = float(view_num)
    cost = Q / n - (S * S) / (n * n)                          # model.py:458-461
    return cost.permute(1, 0, 2, 3).contiguous()              # (C,D,H,W)


def _pad_same(x, stride):
    pads = []
    for n in reversed(x.shape[-3:]):                          # F.pad order: W, H, D
        _, pb, pa = O.same_pad(int(n), 3, stride)
        pads += [pb, pa]
    return F.pad(x, pads)


def _conv(x, w, stride):
    wt = torch.from_numpy(np.ascontiguousarray(w)).permute(4, 3, 0, 1, 2).contiguous()   # (Co,Ci,kd,kh,kw)
    return F.conv3d(_pad_same(x, stride), wt, stride=stride)


def _deconv(x, w):
    wt = torch.from_numpy(np.ascontiguousarray(w)).permute(4, 3, 0, 1, 2).contiguous()   # (Ci,Co,kd,kh,kw)
    y = F.conv_transpose3d(x, wt, stride=2)
    D, H, W = x.shape[-3:]
    return y[..., : 2 * D, : 2 * H, : 2 * W]


def _bn_relu(x, p, eps=1e-5):
    g = torch.from_numpy(np.asarray(p["gamma"], np.float32))
    b = torch.from_numpy(np.asarray(p["beta"], np.float32))
    return F.relu(F.batch_norm(x, None, None, g, b, training=True, eps=eps))


def regnet_us0(cost, params):
    """cost (C,D,H,W) tensor -> (D,H,W) tensor."""
    x = cost[None]
    cb = lambda t, n, s: _bn_relu(_conv(t, params[n]["w"], s), params[n])
    db = lambda t, n: _bn_relu(_deconv(t, params[n]["w"]), params[n])
    c1_0 = cb(x, "3dconv1_0", 2); c2_0 = cb(c1_0, "3dconv2_0", 2); c3_0 = cb(c2_0, "3dconv3_0", 2)
    c0_1 = cb(x, "3dconv0_1", 1); c1_1 = cb(c1_0, "3dconv1_1", 1); c2_1 = cb(c2_0, "3dconv2_1", 1)
    c3_1 = cb(c3_0, "3dconv3_1", 1)
    c4 = db(c3_1, "3dconv4_0") + c2_1
    c5 = db(c4, "3dconv5_0") + c1_1
    c6 = db(c5, "3dconv6_0") + c0_1
    return _conv(c6, params["3dconv6_2"]["w"], 1)[0, 0]


def softargmin_prob(reg, depth_start, depth_interval):
    D = reg.shape[0]
    P = torch.softmax(-reg, dim=0)
    z = torch.from_numpy(O.depth_values(D, depth_start, depth_interval, False, np.float32))
    depth = (P * z[:, None, None]).sum(0)
    idx = (depth - float(depth_start)) / float(depth_interval)
    l0 = idx.floor().long().clamp(0, D - 1); r0 = idx.ceil().long().clamp(0, D - 1)
    l1 = (l0 - 1).clamp(0, D - 1); r1 = (r0 + 1).clamp(0, D - 1)
    g = lambda i: torch.gather(P, 0, i[None])[0]
    return depth, g(l0) + g(r0) + g(l1) + g(r1)


@torch.no_grad()
def inference_mem_from_features(features, cams, depth_num, depth_start, depth_interval, regnet_params):
    """Same contract as mvsnet_oracle.inference_mem_from_features (non-inverse depth), fp32,
    all host cores.  Returns numpy depth (H,W), prob (H,W)."""
    N = features.shape[0]
    Hs = np.stack([O.get_homographies(cams[0], cams[v], depth_num, depth_start, depth_interval, np.float32)
                   for v in range(1, N)])
    T = O.homography_to_transform8(Hs, np.float32)
    cost = cost_volume(features, T, N)
    reg = regnet_us0(cost, regnet_params)
    depth, prob = softargmin_prob(reg, depth_start, depth_interval)
    return depth.numpy(), prob.numpy()
