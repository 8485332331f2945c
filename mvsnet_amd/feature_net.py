"""UNetDS2GN 2D feature extractor on PyTorch-ROCm (north_star keeps it on torch).

Restates mvsnet/cnn_wrapper/mvsnetworks.py:53-115 with the layer semantics of
mvsnet/cnn_wrapper/network.py:217-276 (conv_gn: conv no-bias -> GroupNorm(G = C//8, eps 1e-5,
gamma, beta) -> ReLU), :350-409 (deconv_gn: no ReLU) and :171-215 (plain conv for conv10_2),
all with TensorFlow 'SAME' padding (asymmetric for stride 2).  It produces the hot path's input;
the HIP towers for inference are feature_net_hip.HipUNetDS2GN; this module
is the fallback-free PyTorch-ROCm tower that training differentiates (north_star keeps it on torch).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# name, kind (cg = conv_gn+ReLU, dg = deconv_gn, c = plain conv), sources, kernel, cout multiple
# of base_filter, stride.  Order is a valid topological order of mvsnetworks.py:60-115.
UNET_LAYERS = (
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


def unet_macs(H, W, base_filter=8, image_channels=3):
    """Algorithmic multiply-adds of one UNetDS2GN pass over one H x W image (mvsnetworks.py:53-115): every conv / transposed
    conv counted with the taps and channels the reference's layers have (a k3 s2 transposed conv touches 9/4 taps per output)."""
    shape = {"data": (H, W, image_channels)}
    total = 0.0
    for name, kind, srcs, k, mult, stride in UNET_LAYERS:
        h, w, _ = shape[srcs[0]]
        cin = sum(shape[s_][2] for s_ in srcs)
        cout = base_filter * mult
        if kind == "dg":
            ho, wo = 2 * h, 2 * w
            total += ho * wo * 2.25 * cin * cout
        else:
            ho, wo = -(-h // stride), -(-w // stride)
            total += ho * wo * k * k * cin * cout
        shape[name] = (ho, wo, cout)
    return total


def unet_layer_work(H, W, base_filter=8, image_channels=3):
    """Per layer of one UNetDS2GN pass over one H x W image: (name, multiply-adds, algorithmic bytes = every input tensor read
    once + the output written once, fp32) -- the two rooflines a layer can be priced against."""
    shape = {"data": (H, W, image_channels)}
    rows = []
    for name, kind, srcs, k, mult, stride in UNET_LAYERS:
        h, w, _ = shape[srcs[0]]
        cin = sum(shape[s_][2] for s_ in srcs)
        cout = base_filter * mult
        if kind == "dg":
            ho, wo = 2 * h, 2 * w
            macs = ho * wo * 2.25 * cin * cout
        else:
            ho, wo = -(-h // stride), -(-w // stride)
            macs = ho * wo * k * k * cin * cout
        shape[name] = (ho, wo, cout)
        rows.append((name, macs, 4.0 * (h * w * cin + ho * wo * cout)))
    return rows


def _same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


class UNetDS2GN:
    """Functional module (inference entry point; see unet_forward / trainable_layers for training).  ``params`` uses TensorFlow variable layouts
    (numpy or torch): conv 'w' (k,k,Cin,Cout), transposed conv 'w' (k,k,Cout,Cin), 'gamma',
    'beta' (Cout,)."""

    def __init__(self, params, device="cuda", dtype=torch.float32, hip_group_norm=False):
        """`hip_group_norm`: GroupNorm of the layers with >= 8 channels through HipGroupNorm (the narrow-mode
        fallback of MVSNetWeights: torch's group_norm is 10x slower on channel-last tensors); off by default so that
        this module stays an independent cross-check of the HIP extractor."""
        self.device = torch.device(device)
        self.hip_group_norm = bool(hip_group_norm) and torch.device(device).type == "cuda"
        self.layers = []
        for name, kind, srcs, k, _mult, stride in UNET_LAYERS:
            p = params[name]
            w = torch.as_tensor(p["w"], dtype=dtype)
            if kind == "dg":
                w = w.permute(3, 2, 0, 1)      # (k,k,Cout,Cin) -> torch conv_transpose (Cin,Cout,k,k)
            else:
                w = w.permute(3, 2, 0, 1)      # (k,k,Cin,Cout) -> torch conv (Cout,Cin,k,k)
            w = w.contiguous().to(self.device)
            g = b = None
            if kind != "c":
                g = torch.as_tensor(p["gamma"], dtype=dtype).to(self.device)
                b = torch.as_tensor(p["beta"], dtype=dtype).to(self.device)
            self.layers.append((name, kind, srcs, k, stride, w, g, b))
        self.out_channels = self.layers[-1][5].shape[0]

    @torch.no_grad()
    def __call__(self, images):
        """images (V,H,W,3) channel-last float32 -> features (V,H/4,W/4,C) contiguous.
        Each view is normalised independently (GroupNorm is per sample), so running the V
        towers of mvsnet/model.py:392-406 as one batch is equivalent."""
        return unet_forward(self.layers, images.to(self.device), hip_group_norm=self.hip_group_norm)


class HipGroupNorm(torch.autograd.Function):
    """GroupNorm(8 channels per group, eps) [+ ReLU] on libmvsnet_hip.so for the training towers: four HBM passes
    (per-channel float64 sums -> normalise; backward: sums of gz and gz*xhat -> input gradient) instead of
    torch's group_norm, whose moments kernel alone takes ~0.4 ms per layer on channel-last tensors (12 of the
    14.5 ms of a 3-view tower forward).  x (V,C,H,W) in channels_last memory."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu):
        from . import _lib
        lib = _lib.load()
        x = x.contiguous(memory_format=torch.channels_last)
        V, C, H, W = x.shape
        stats = torch.zeros((V, 2, C), device=x.device, dtype=torch.float64)
        y = torch.empty_like(x)                             # preserves channels_last
        g_, b_ = gamma.detach().contiguous(), beta.detach().contiguous()
        _lib.check(lib.mvs_gn_stats_f32(_lib.ptr(x.permute(0, 2, 3, 1)), V, H * W, C, _lib.ptr(stats), _lib.stream_ptr()),
                   "mvs_gn_stats_f32")
        _lib.check(lib.mvs_gn_apply_f32(_lib.ptr(x.permute(0, 2, 3, 1)), _lib.ptr(stats), _lib.ptr(g_), _lib.ptr(b_),
                                        float(eps), int(relu), V, H * W, C, _lib.ptr(y.permute(0, 2, 3, 1)),
                                        _lib.stream_ptr()), "mvs_gn_apply_f32")
        ctx.save_for_backward(x, stats, g_, b_)
        ctx.cfg = (float(eps), int(relu))
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        x, stats, gamma, beta = ctx.saved_tensors
        eps, relu = ctx.cfg
        V, C, H, W = x.shape
        g = g.contiguous(memory_format=torch.channels_last)
        sums = torch.zeros(lib.mvs_gn_bwd_sums_doubles(V, C), device=x.device, dtype=torch.float64)   # (slots, V, 2, C)
        dx = torch.empty_like(x)
        xp, gp = _lib.ptr(x.permute(0, 2, 3, 1)), _lib.ptr(g.permute(0, 2, 3, 1))
        _lib.check(lib.mvs_gn_bwd_reduce_f32(xp, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps, relu, gp, V, H * W, C,
                                             _lib.ptr(sums), _lib.stream_ptr()), "mvs_gn_bwd_reduce_f32")
        _lib.check(lib.mvs_gn_bwd_apply_f32(xp, _lib.ptr(stats), _lib.ptr(gamma), _lib.ptr(beta), eps, relu, gp, _lib.ptr(sums),
                                            V, H * W, C, _lib.ptr(dx.permute(0, 2, 3, 1)), _lib.stream_ptr()),
                   "mvs_gn_bwd_apply_f32")
        tot = sums.view(-1, 2, C).sum(0).to(torch.float32)
        return dx, tot[1], tot[0], None, None


def unet_forward(layers, images, hip_group_norm=False):
    """The tower itself; differentiable (training, SURVEY 8f f4, runs it under torch autograd with the
    layer tensors being views of the trainer's flat parameter buffer).  `layers`: tuples
    (name, kind, srcs, k, stride, w torch-layout, gamma, beta).  `hip_group_norm`: GroupNorm(+ReLU) through
    HipGroupNorm (the trainer's choice) instead of torch's group_norm + relu."""
    x = images.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    acts = {"data": x}
    for name, kind, srcs, k, stride, w, g, b in layers:
        x = acts[srcs[0]] if len(srcs) == 1 else torch.cat([acts[s] for s in srcs], dim=1)
        if kind == "dg":
            n_h, n_w = x.shape[2], x.shape[3]
            y = F.conv_transpose2d(x, w, stride=stride)          # full length s*(n-1)+k
            pb_h = _same_pad(n_h * stride, k, stride)[0]
            pb_w = _same_pad(n_w * stride, k, stride)[0]
            y = y[:, :, pb_h:pb_h + n_h * stride, pb_w:pb_w + n_w * stride]
        else:
            ph = _same_pad(x.shape[2], k, stride)
            pw = _same_pad(x.shape[3], k, stride)
            if ph[0] == ph[1] and pw[0] == pw[1]:
                y = F.conv2d(x, w, stride=stride, padding=(ph[0], pw[0]))
            else:
                y = F.conv2d(F.pad(x, (pw[0], pw[1], ph[0], ph[1])), w, stride=stride)
        if kind != "c":
            C = y.shape[1]
            if hip_group_norm and C % 8 == 0 and 256 % (C // 4) == 0:     # the shapes mvs_gn_*_f32 tile
                y = HipGroupNorm.apply(y, g, b, 1e-5, kind == "cg")
            else:
                y = F.group_norm(y, max(1, C // 8), g, b, eps=1e-5)   # network.py:246-254
                if kind == "cg":
                    y = F.relu(y)
        acts[name] = y
    return acts["conv10_2"].permute(0, 2, 3, 1).contiguous()


def trainable_layers(params):
    """`params[name]` = {'w', 'gamma', 'beta'} torch tensors in the TensorFlow layouts (leaves or views that
    require grad) -> the `layers` list of unet_forward (kernels permuted to the torch layouts per call)."""
    layers = []
    for name, kind, srcs, k, _mult, stride in UNET_LAYERS:
        p = params[name]
        w = p["w"].permute(3, 2, 0, 1)           # conv (k,k,Cin,Cout) -> (Cout,Cin,k,k); deconv (k,k,Cout,Cin) -> (Cin,Cout,k,k)
        layers.append((name, kind, srcs, k, stride, w, p.get("gamma"), p.get("beta")))
    return layers
