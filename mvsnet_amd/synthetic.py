"""Deterministic synthetic workloads.

No dataset or checkpoint exists offline, so every test, ``smoke()`` and ``bench.py``
input is generated here: feature maps / images, DTU-like cameras, and randomly
initialised regulariser weights.  Pure numpy (``RandomState`` streams are frozen across
numpy versions), no GPU, no oracle import.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

# Named workloads: (view_num N, depth planes D, feature H, feature W, depth_interval)
# M  = BASELINE.json metric config; c1/c2/c3 = configs[0..2]; toy = golden-fixture size.
WORKLOADS = {
    "M": dict(view_num=5, depth_num=192, height=128, width=160, interval=2.5 * 1.06),
    "c1": dict(view_num=3, depth_num=32, height=128, width=160, interval=15.9),
    "c2": dict(view_num=5, depth_num=192, height=216, width=288, interval=2.5 * 1.06),
    "c3": dict(view_num=5, depth_num=256, height=300, width=400, interval=2.5 * 0.8),
    "toy": dict(view_num=3, depth_num=8, height=16, width=16, interval=60.0),
    "small": dict(view_num=3, depth_num=16, height=32, width=48, interval=30.0),
}

DEPTH_START = 425.0
PIVOT_DEPTH = 650.0      # sources rotate about the point (0, 0, 650 mm)
FOCAL_AT_160 = 361.54    # DTU focal length at 1/4 of a 640-wide image


def base_divisor(network_mode: str) -> float:
    """network_mode -> base_divisor, mvsnet/cnn_wrapper/network.py:75-85.

    'semilite' is written `4/3` there, in a Python 2.7 module without `from __future__ import division`
    (network.py:9 imports only print_function), so at the reference's run time it is the INTEGER 1:
    semilite towers and regulariser have the channel counts of 'normal' (base_filter 8), and that is
    what a reference semilite checkpoint holds.  Only the ConvGRU filter counts differ (model.py:641:
    halved for every mode but 'normal').  'semilite-py3' (not a reference mode) keeps the true-division
    reading, base_filter 6: channel counts 6/12/24/48 that exercise the shape-generic kernels."""
    return {"normal": 1.0, "semilite": 1.0, "semilite-py3": 4.0 / 3.0, "lite": 2.0, "ultralite": 4.0,
            "fat": 0.5, "ultrafat": 0.25}[network_mode]


def base_filter(network_mode: str) -> int:
    """max(1, int(8 / base_divisor)), mvsnet/cnn_wrapper/mvsnetworks.py:57-58,126-127."""
    return max(1, int(8 / base_divisor(network_mode)))


def view_angles(view_num: int):
    if view_num == 3:
        return [-8.0, 8.0]
    if view_num == 5:
        return [-12.0, -6.0, 6.0, 12.0]
    n = view_num - 1
    return [(-1.0) ** i * 12.0 * (i // 2 + 1) / max(1, (n + 1) // 2) for i in range(n)]


def make_cams(view_num, height, width, depth_num, depth_start=DEPTH_START, interval=2.65):
    """cams (N,2,4,4) float32 in the reference layout: cams[v,0] = world->camera
    extrinsic, cams[v,1,:3,:3] = K at feature resolution, cams[v,1,3] = (depth_min,
    depth_interval, depth_num, depth_max)."""
    f = FOCAL_AT_160 * width / 160.0
    K = np.array([[f, 0, width / 2.0], [0, f, height / 2.0], [0, 0, 1.0]])
    P = np.array([0.0, 0.0, PIVOT_DEPTH])
    cams = np.zeros((view_num, 2, 4, 4), np.float64)
    for v, theta in enumerate([0.0] + view_angles(view_num)):
        a = math.radians(theta)
        Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        R = Ry.T
        t = P - R @ P
        cams[v, 0, :3, :3] = R
        cams[v, 0, :3, 3] = t
        cams[v, 0, 3, 3] = 1.0
        cams[v, 1, :3, :3] = K
        cams[v, 1, 3] = (depth_start, interval, depth_num, depth_start + interval * depth_num)
    return cams.astype(np.float32)


def make_features(view_num, height, width, channels, seed=0):
    """F_v ~ N(0,1); sources blended with the reference (0.5/0.5) so that variances are
    neither degenerate nor pure noise."""
    rs = np.random.RandomState(seed)
    f = rs.standard_normal((view_num, height, width, channels)).astype(np.float32)
    f[1:] = 0.5 * f[:1] + 0.5 * f[1:]
    return f


def make_images(view_num, height, width, seed=0):
    rs = np.random.RandomState(seed + 100)
    return rs.standard_normal((view_num, height, width, 3)).astype(np.float32)


def _he(rs, shape, fan_in):
    return (rs.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)


def make_regnet_params(network_mode="normal", seed=1, in_channels=None, random_affine=False):
    """Randomly initialised RegNetUS0 parameters in TensorFlow variable layouts:
    conv kernels (3,3,3,Cin,Cout), transposed-conv kernels (3,3,3,Cout,Cin), BN gamma/beta
    (Cout,).  Topology: mvsnet/cnn_wrapper/mvsnetworks.py:122-158."""
    b = base_filter(network_mode)
    cin0 = 4 * b if in_channels is None else in_channels
    rs = np.random.RandomState(seed)
    spec = [
        ("3dconv1_0", "conv", cin0, 2 * b), ("3dconv2_0", "conv", 2 * b, 4 * b),
        ("3dconv3_0", "conv", 4 * b, 8 * b), ("3dconv0_1", "conv", cin0, b),
        ("3dconv1_1", "conv", 2 * b, 2 * b), ("3dconv2_1", "conv", 4 * b, 4 * b),
        ("3dconv3_1", "conv", 8 * b, 8 * b), ("3dconv4_0", "deconv", 8 * b, 4 * b),
        ("3dconv5_0", "deconv", 4 * b, 2 * b), ("3dconv6_0", "deconv", 2 * b, b),
        ("3dconv6_2", "conv", b, 1),
    ]
    params = {}
    for name, kind, cin, cout in spec:
        shape = (3, 3, 3, cin, cout) if kind == "conv" else (3, 3, 3, cout, cin)
        p = {"w": _he(rs, shape, 27 * cin)}
        if name != "3dconv6_2":
            if random_affine:
                p["gamma"] = (1.0 + 0.2 * rs.standard_normal(cout)).astype(np.float32)
                p["beta"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
            else:
                p["gamma"] = np.ones(cout, np.float32)
                p["beta"] = np.zeros(cout, np.float32)
        params[name] = p
    return params


def gru_filters(network_mode="normal"):
    """mvsnet/model.py:641-645: base_divisor is 1 for 'normal', 2 otherwise."""
    d = 1 if network_mode == "normal" else 2
    return int(16 / d), int(4 / d), int(2 / d)


def make_gru_params(network_mode="normal", seed=2, in_channels=32, random_affine=False):
    """ConvGRU x3 + prob_conv parameters (mvsnet/convgru.py:82-122, model.py:641-660,701):
    gates_w (3,3,Cin+F,2F), out_w (3,3,Cin+F,F), biases, LayerNorm gamma/beta (F,)."""
    rs = np.random.RandomState(seed)
    params = {}
    cin = in_channels
    for i, F in enumerate(gru_filters(network_mode), start=1):
        c = cin + F
        p = {"gates_w": _he(rs, (3, 3, c, 2 * F), 9 * c), "out_w": _he(rs, (3, 3, c, F), 9 * c)}
        for nm, n in (("gates_b", 2 * F), ("out_b", F)):
            p[nm] = (0.05 * rs.standard_normal(n)).astype(np.float32) if random_affine else np.zeros(n, np.float32)
        for nm in ("reset", "update", "out"):
            p[nm + "_gamma"] = (1 + 0.2 * rs.standard_normal(F)).astype(np.float32) if random_affine else np.ones(F, np.float32)
            p[nm + "_beta"] = (0.1 * rs.standard_normal(F)).astype(np.float32) if random_affine else np.zeros(F, np.float32)
        params["gru%d" % i] = p
        cin = F
    params["prob_w"] = _he(rs, (3, 3, cin, 1), 9 * cin)
    params["prob_b"] = (0.05 * rs.standard_normal(1)).astype(np.float32) if random_affine else np.zeros(1, np.float32)
    return params


def make_unet_params(network_mode="normal", seed=3):
    """UNetDS2GN parameters (mvsnet/cnn_wrapper/mvsnetworks.py:53-115) in TF layouts:
    conv (k,k,Cin,Cout), transposed conv (k,k,Cout,Cin), GroupNorm gamma/beta."""
    from .feature_net import UNET_LAYERS  # topology table lives with the torch module
    b = base_filter(network_mode)
    rs = np.random.RandomState(seed)
    chans = {"data": 3}
    params = {}
    for name, kind, srcs, k, mult, _stride in UNET_LAYERS:
        cin = sum(chans[s] for s in srcs)
        cout = mult * b
        chans[name] = cout
        shape = (k, k, cout, cin) if kind == "dg" else (k, k, cin, cout)
        p = {"w": _he(rs, shape, k * k * cin)}
        if kind != "c":
            p["gamma"] = (1.0 + 0.1 * rs.standard_normal(cout)).astype(np.float32)
            p["beta"] = (0.05 * rs.standard_normal(cout)).astype(np.float32)
        params[name] = p
    return params


@dataclass
class Workload:
    name: str
    view_num: int
    depth_num: int
    height: int
    width: int
    channels: int
    depth_start: float
    depth_interval: float
    features: np.ndarray     # (N,H,W,C)
    cams: np.ndarray         # (N,2,4,4)

    @property
    def depth_end(self):
        return self.depth_start + (self.depth_num - 1) * self.depth_interval


def make_workload(name="M", network_mode="normal", seed=0) -> Workload:
    w = WORKLOADS[name]
    C = 4 * base_filter(network_mode)
    feats = make_features(w["view_num"], w["height"], w["width"], C, seed)
    cams = make_cams(w["view_num"], w["height"], w["width"], w["depth_num"], DEPTH_START, w["interval"])
    return Workload(name, w["view_num"], w["depth_num"], w["height"], w["width"], C,
                    DEPTH_START, float(np.float32(w["interval"])), feats, cams)


def write_session(path, n_images=8, height=512, width=640, view_num=5, depth_num=192, interval=2.5 * 1.06, seed=0):
    """A synthetic on-disk session in the reference's format (mvsnet/mvs_data_generation/mvs_cluster.py:72-127,
    cluster_generator.py:126-156): images/<i>.jpg, cameras/<i>.json (pose in metres, intrinsics in pixels),
    covisibility.json listing the view_num - 1 nearest images of every reference view with the depth range that makes
    `depth_num` planes of `interval` mm.  Cameras sit on an arc around the point (0, 0, 650 mm), as make_cams places
    them; the images are smooth random textures (JPEG-friendly, unlike white noise).  For end-to-end throughput runs
    (bench.py `session_depth_maps_per_s`) and tests; no dataset exists offline."""
    import json
    import os
    from PIL import Image
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(path, "images"), exist_ok=True)
    os.makedirs(os.path.join(path, "cameras"), exist_ok=True)
    focal = FOCAL_AT_160 * (width / 160.0)
    covis = {}
    for i in range(n_images):
        low = rs.randint(0, 256, size=(height // 16 + 1, width // 16 + 1, 3)).astype(np.uint8)
        img = np.asarray(Image.fromarray(low).resize((width, height), Image.BICUBIC)).astype(np.int16)
        img = np.clip(img + rs.randint(-6, 7, size=img.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(path, "images", "%d.jpg" % i), quality=90)
        theta = math.radians(-12.0 + 24.0 * i / max(1, n_images - 1))
        c, s_ = math.cos(theta), math.sin(theta)
        R = np.array([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]])
        pivot = np.array([0.0, 0.0, PIVOT_DEPTH / 1000.0])              # metres
        t = pivot - R @ pivot                                           # x_cam = R (x - pivot) + pivot
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = R, t
        cam = {"pose": {"matrix": {"%d,%d" % (r, c_): float(pose[r, c_]) for r in range(4) for c_ in range(4)}},
               "intrinsics": {"fx": focal, "fy": focal, "px": width / 2.0, "py": height / 2.0}}
        with open(os.path.join(path, "cameras", "%d.json" % i), "w") as f:
            json.dump(cam, f)
        near = sorted((j for j in range(n_images) if j != i), key=lambda j: (abs(j - i), j))[:view_num - 1]
        covis[str(i)] = {"views": near, "min_depth": DEPTH_START, "max_depth": DEPTH_START + (depth_num - 1) * interval}
    with open(os.path.join(path, "covisibility.json"), "w") as f:
        json.dump(covis, f)
    return path
