"""Depth-map refinement (`--refinement`), mirroring mvsnet/model.py:753-811 with the two refinement
towers of mvsnet/cnn_wrapper/mvsnetworks.py:178-193 (RefineNetConv, "original") and :261-324
(RefineUNetConv, "unet") -- SURVEY 8f row f3.  Small 2D convolutions on one image: host glue in
PyTorch-ROCm like the feature extractor, not a HIP target.

Semantics restated from the reference / TF 1.12:
  * the initial depth is normalised to [0,1] by (d - start) / ((D-1) * interval); the tower predicts
    a normalised residual, re-scaled and (residual_refinement) added back (model.py:758-809);
  * `tf.image.resize_bilinear` with its default align_corners=False is the LEGACY mapping
    src = dst * (in / out) (no half-pixel shift), bottom/right index clamped (model.py:768-781);
  * tower layers are `Network.conv` / `deconv` with their defaults: bias, ReLU (none on the last
    layer), SAME padding (network.py:171-215,300-329).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .feature_net import _same_pad
from .synthetic import base_divisor

# name, kind ("c" conv / "d" transposed conv), sources, out-channel multiple of the base filter, stride, relu
REFINE_ORIGINAL = (
    ("refine_conv0", "c", ("concat_image",), 1, 1, True), ("refine_conv1", "c", ("refine_conv0",), 1, 1, True),
    ("refine_conv2", "c", ("refine_conv1",), 1, 1, True), ("refine_conv3", "c", ("refine_conv2",), 0, 1, False),
)
_U = "2dconv%s_refine"
REFINE_UNET = (
    (_U % "1_0", "c", ("concat_image",), 2, 2, True), (_U % "2_0", "c", (_U % "1_0",), 4, 2, True),
    (_U % "3_0", "c", (_U % "2_0",), 8, 2, True), (_U % "4_0", "c", (_U % "3_0",), 16, 2, True),
    (_U % "0_1", "c", ("concat_image",), 1, 1, True), (_U % "0_2", "c", (_U % "0_1",), 1, 1, True),
    (_U % "1_1", "c", (_U % "1_0",), 2, 1, True), (_U % "1_2", "c", (_U % "1_1",), 2, 1, True),
    (_U % "2_1", "c", (_U % "2_0",), 4, 1, True), (_U % "2_2", "c", (_U % "2_1",), 4, 1, True),
    (_U % "3_1", "c", (_U % "3_0",), 8, 1, True), (_U % "3_2", "c", (_U % "3_1",), 8, 1, True),
    (_U % "4_1", "c", (_U % "4_0",), 16, 1, True), (_U % "4_2", "c", (_U % "4_1",), 16, 1, True),
    (_U % "5_0", "d", (_U % "4_2",), 8, 2, True),
    (_U % "5_1", "c", (_U % "5_0", _U % "3_2"), 8, 1, True), (_U % "5_2", "c", (_U % "5_1",), 8, 1, True),
    (_U % "6_0", "d", (_U % "5_2",), 4, 2, True),
    (_U % "6_1", "c", (_U % "6_0", _U % "2_2"), 4, 1, True), (_U % "6_2", "c", (_U % "6_1",), 4, 1, True),
    (_U % "7_0", "d", (_U % "6_2",), 2, 2, True),
    (_U % "7_1", "c", (_U % "7_0", _U % "1_2"), 2, 1, True), (_U % "7_2", "c", (_U % "7_1",), 2, 1, True),
    (_U % "8_0", "d", (_U % "7_2",), 1, 2, True),
    (_U % "8_1", "c", (_U % "8_0", _U % "0_2"), 1, 1, True), (_U % "8_2", "c", (_U % "8_1",), 1, 1, True),
    (_U % "8_3", "c", (_U % "8_2",), 4, 1, True), (_U % "8_4", "c", (_U % "8_3",), 0, 1, False),
)


def refine_layers(network_type):
    if network_type == "original":
        return REFINE_ORIGINAL, 32
    if network_type == "unet":
        return REFINE_UNET, 8
    raise NotImplementedError(network_type)                       # model.py:800-801


def make_refine_params(network_type="original", network_mode="normal", in_channels=4, seed=4):
    """Seeded parameters in TF layouts: conv (3,3,Cin,Cout), transposed conv (3,3,Cout,Cin), bias."""
    table, base = refine_layers(network_type)
    b = max(1, int(base / base_divisor(network_mode)))
    rs = np.random.RandomState(seed)
    chans = {"concat_image": in_channels}
    params = {}
    for name, kind, srcs, mult, _stride, _relu in table:
        cin = sum(chans[s] for s in srcs)
        cout = mult * b if mult else 1
        chans[name] = cout
        shape = (3, 3, cout, cin) if kind == "d" else (3, 3, cin, cout)
        params[name] = {"w": (rs.standard_normal(shape) * np.sqrt(2.0 / (9 * cin))).astype(np.float32),
                        "b": (0.05 * rs.standard_normal(cout)).astype(np.float32)}
    return params


def resize_bilinear_tf1(x, out_h, out_w):
    """tf.image.resize_bilinear(x, [out_h, out_w]) (align_corners=False, TF 1.x) on NHWC tensors."""
    n, h, w, c = x.shape
    if (h, w) == (out_h, out_w):
        return x

    def axis(n_in, n_out):
        src = torch.arange(n_out, device=x.device, dtype=torch.float32) * (float(n_in) / float(n_out))
        i0 = torch.floor(src).to(torch.int64).clamp_(max=n_in - 1)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        return i0, i1, (src - i0.to(torch.float32))

    y0, y1, fy = axis(h, out_h)
    x0, x1, fx = axis(w, out_w)
    top = x[:, y0][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y0][:, :, x1] * fx[None, None, :, None]
    bot = x[:, y1][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y1][:, :, x1] * fx[None, None, :, None]
    return top * (1 - fy)[None, :, None, None] + bot * fy[None, :, None, None]


class RefineNet:
    """Functional inference module for either tower; `params` = {layer: {"w", "b"}} in TF layouts."""

    def __init__(self, params, network_type="original", device="cuda"):
        self.table, _ = refine_layers(network_type)
        self.network_type = network_type
        self.device = torch.device(device)
        self.layers = []
        for name, kind, srcs, _mult, stride, relu in self.table:
            w = torch.as_tensor(params[name]["w"], dtype=torch.float32).permute(3, 2, 0, 1).contiguous().to(self.device)
            b = torch.as_tensor(params[name]["b"], dtype=torch.float32).to(self.device)
            self.layers.append((name, kind, srcs, stride, relu, w, b))

    @torch.no_grad()
    def __call__(self, color_image, depth_image):
        """color_image (B,H,W,3), depth_image (B,H,W,1|2) channel-last -> (B,H,W,1)."""
        return refine_forward(self.table, self.layers, color_image.to(self.device), depth_image.to(self.device))


def refine_forward(table, layers, color_image, depth_image):
    """The tower itself; differentiable (the trainer runs it under torch autograd).  `layers`: tuples
    (name, kind, srcs, stride, relu, w torch-layout, b)."""
    x = torch.cat([color_image, depth_image], dim=3).permute(0, 3, 1, 2).contiguous()
    acts = {"concat_image": x}
    for name, kind, srcs, stride, relu, w, b in layers:
        x = acts[srcs[0]] if len(srcs) == 1 else torch.cat([acts[s] for s in srcs], dim=1)
        if kind == "d":
            n_h, n_w = x.shape[2], x.shape[3]
            y = F.conv_transpose2d(x, w, stride=stride)[:, :, :n_h * stride, :n_w * stride] + b[None, :, None, None]
        else:
            ph, pw = _same_pad(x.shape[2], 3, stride), _same_pad(x.shape[3], 3, stride)
            y = F.conv2d(F.pad(x, (pw[0], pw[1], ph[0], ph[1])), w, b, stride=stride)
        acts[name] = F.relu(y) if relu else y
    return acts[table[-1][0]].permute(0, 2, 3, 1).contiguous()


def trainable_refine_layers(params, network_type):
    """`params[name]` = {'w', 'b'} torch tensors in the TensorFlow layouts (leaves that require grad) -> (table, layers)
    for refine_forward."""
    table, _ = refine_layers(network_type)
    layers = [(name, kind, srcs, stride, relu, params[name]["w"].permute(3, 2, 0, 1), params[name]["b"])
              for name, kind, srcs, _mult, stride, relu in table]
    return table, layers


def depth_refine(init_depth_map, image, prob_map, depth_num, depth_start, depth_interval, refine_net,
                 upsample_depth=False, refine_with_confidence=False, residual_refinement=True, stereo_image=None):
    """model.py:753-811.  init_depth_map, prob_map (B,h,w,1); image (B,H,W,3), the centred reference
    image; stereo_image (B,H,W,3) the optional stereo partner (:777-789, training only in the reference).
    Returns (refined_depth_map, residual_depth_map)."""
    depth_start = float(depth_start); depth_interval = float(depth_interval)
    depth_scale = (depth_start + (float(depth_num) - 1.0) * depth_interval) - depth_start
    norm = (init_depth_map - depth_start) / depth_scale
    if upsample_depth:
        H, W = image.shape[1], image.shape[2]
        norm = resize_bilinear_tf1(norm, H, W)
        init_depth_map = resize_bilinear_tf1(init_depth_map, H, W)
        if refine_with_confidence:
            prob_map = resize_bilinear_tf1(prob_map, H, W)
    else:
        image = resize_bilinear_tf1(image, init_depth_map.shape[1], init_depth_map.shape[2])
        if stereo_image is not None:
            stereo_image = resize_bilinear_tf1(stereo_image, init_depth_map.shape[1], init_depth_map.shape[2])
    data = torch.cat([norm, prob_map], dim=3) if refine_with_confidence else norm
    if stereo_image is not None:
        data = torch.cat([data, stereo_image], dim=3)
    residual = refine_net(image, data) * depth_scale
    refined = residual + init_depth_map if residual_refinement else residual
    return refined, residual
