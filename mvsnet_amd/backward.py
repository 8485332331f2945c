"""Training-time forward and backward of the plane-sweep hot path on the HIP library.

The reference trains through `inference` (mvsnet/model.py:257-372: towers -> eager variance cost volume
-> RegNetUS0 -> soft-argmin) and lets TensorFlow differentiate the graph (`opt.compute_gradients`,
mvsnet/train.py:428-429).  Here the hot path is one `torch.autograd.Function` whose forward and backward
are sequences of C-ABI calls: torch owns memory, the 2D towers (`feature_net.UNetDS2GN`, differentiated by
torch autograd, as north_star keeps them on PyTorch-ROCm) and the loss expressions (`loss.py`).

Layer bookkeeping of RegNetUS0 (mvsnetworks.py:122-158): every BatchNorm layer keeps its RAW output y,
its float64 batch sums and the folded affine; consumers apply BN+ReLU on load exactly as in inference.
Backward per layer:  g_a (gradient w.r.t. the post-ReLU activation, one or two consumers) -> BN+ReLU
backward (two HBM passes) -> g_y -> weight gradient (MFMA contraction over voxels) and input gradient
(the forward MFMA kernels: conv stride 2 <-> conv_transpose with the same kernel array, stride 1 with the
flipped / transposed kernel).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .model import BN_EPSILON, REGNET_ORDER, bn_finalize, conv3d, conv3d_pair, cost_volume

BN_LAYERS = REGNET_ORDER[:-1]


# ------------------------------------------------------------------------------------------------
# single-op wrappers
# ------------------------------------------------------------------------------------------------

def softargmin_bwd(reg, g_depth, depth_start, depth_interval, inverse_depth=False, g_prob=None):
    """reg (D,H,W), g_depth (H,W) [, g_prob (H,W)] -> g_reg (D,H,W)   (model.py:343-366, 45-144)"""
    lib = _lib.load()
    D, H, W = reg.shape
    g = torch.empty_like(reg)
    gd = _lib.f32(g_depth.contiguous()) if g_depth is not None else None
    gp = _lib.f32(g_prob.contiguous()) if g_prob is not None else None
    _lib.check(lib.mvs_softargmin_bwd_f32(_lib.ptr(reg), _lib.ptr(gd), _lib.ptr(gp), D, H, W,
                                          float(depth_start), float(depth_interval), int(bool(inverse_depth)),
                                          _lib.ptr(g), _lib.stream_ptr()), "mvs_softargmin_bwd_f32")
    return g


def bn_relu(y, affine=None, y2=None, affine2=None):
    """act(y*s+t) [+ act(y2*s2+t2)]: the normalised input a consumer layer sees."""
    lib = _lib.load()
    s, t = affine if affine is not None else (None, None)
    s2, t2 = affine2 if affine2 is not None else (None, None)
    out = torch.empty_like(y)
    Cn = y.shape[-1]
    _lib.check(lib.mvs_bn_relu_f32(_lib.ptr(y), _lib.ptr(s), _lib.ptr(t), _lib.ptr(y2), _lib.ptr(s2), _lib.ptr(t2),
                                   y.numel() // Cn, Cn, _lib.ptr(out), _lib.stream_ptr()), "mvs_bn_relu_f32")
    return out


class SyncBN:
    """Cross-replica BatchNorm for data-parallel training (SURVEY 8f f4 lists it as an addition; the reference's
    towers keep per-GPU statistics): the float64 per-channel sums of every BatchNorm layer -- forward [sum, sumsq],
    backward [sum gz, sum gz*xhat] -- are summed over the ranks and the voxel count is multiplied by the world size,
    exactly as if the replicas' volumes were one batch.  Per-variable gradients stay local (the flat-buffer
    all-reduce averages them afterwards).  RCCL reduces the tiny buffers in place; under gloo (CPU tests) they are
    staged through host memory."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.direct = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def all_reduce(self, t):
        if self.world == 1:
            return t
        if self.direct:
            self.dist.all_reduce(t, group=self.group)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        return t


def bn_relu_bwd(y, stats, affine, gamma, g1, g2=None, eps=BN_EPSILON, sync: Optional[SyncBN] = None, sums=None):
    """BatchNorm(batch statistics)+ReLU backward (network.py:492-509).  Returns (g_y, g_gamma, g_beta).
    `stats` are the sums the forward normalised with (global ones under `sync`).  `sums`: a ZEROED float64 buffer of
    mvs_bn_bwd_sum_slots() * 2 * C entries to use (regnet_backward hands out pieces of one slab: one fill for all layers)."""
    lib = _lib.load()
    Cn = y.shape[-1]
    vox = y.numel() // Cn
    world = sync.world if sync is not None else 1
    if sums is None:
        sums = torch.zeros((lib.mvs_bn_bwd_sum_slots(), 2, Cn), device=y.device, dtype=torch.float64)
    else:
        sums = sums.view(lib.mvs_bn_bwd_sum_slots(), 2, Cn)
    s, t = affine
    _lib.check(lib.mvs_bn_bwd_reduce_f32(_lib.ptr(y), _lib.ptr(stats), float(vox * world), float(eps), _lib.ptr(s), _lib.ptr(t),
                                         _lib.ptr(g1), _lib.ptr(g2), vox, Cn, _lib.ptr(sums), _lib.stream_ptr()),
               "mvs_bn_bwd_reduce_f32")
    g_y = torch.empty_like(y)
    g_gamma = torch.empty(Cn, device=y.device, dtype=torch.float32)
    g_beta = torch.empty_like(g_gamma)
    local = None
    if world > 1:                                      # the variables' gradients are this replica's own sums
        local = sums.sum(0).to(torch.float32)
        sync.all_reduce(sums)
    _lib.check(lib.mvs_bn_bwd_apply_f32(_lib.ptr(y), _lib.ptr(stats), float(vox * world), float(eps), _lib.ptr(s), _lib.ptr(t),
                                        _lib.ptr(gamma), _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(sums), vox, Cn,
                                        _lib.ptr(g_y), _lib.ptr(g_gamma), _lib.ptr(g_beta), _lib.stream_ptr()),
               "mvs_bn_bwd_apply_f32")
    if local is not None:
        g_gamma, g_beta = local[1].contiguous(), local[0].contiguous()
    return g_y, g_gamma, g_beta


_WS: Dict[torch.device, torch.Tensor] = {}


def conv3d_wgrad(big, small, stride):
    """dW(3,3,3,Cbig,Csmall) = sum_o big(stride*o + tap - pad) (x) small(o); see include/mvsnet_hip.h."""
    lib = _lib.load()
    D, H, W, Cb = big.shape
    Cs = small.shape[-1]
    need = lib.mvs_conv3d_wgrad_workspace_bytes(D, H, W, Cb, Cs, stride)
    if need == 0:
        raise _lib.MvsnetHipError("weight gradient not built for %d x %d channels, stride %d" % (Cb, Cs, stride))
    ws = _WS.get(big.device)
    if ws is None or ws.numel() < need:
        ws = _WS[big.device] = torch.empty(need, device=big.device, dtype=torch.uint8)
    dw = torch.empty((3, 3, 3, Cb, Cs), device=big.device, dtype=torch.float32)
    _lib.check(lib.mvs_conv3d_wgrad_f32(_lib.ptr(big), _lib.ptr(small), D, H, W, Cb, Cs, stride,
                                        C.c_void_p(ws.data_ptr()), ws.numel(), _lib.ptr(dw), _lib.stream_ptr()),
               "mvs_conv3d_wgrad_f32")
    return dw


_CVWS: Dict[torch.device, torch.Tensor] = {}


def cost_volume_bwd(ref, src, transforms, g1, g2=None, method="gather"):
    """Gradient of the warp + variance w.r.t. the feature maps: (g_ref (H,W,C), g_src (N-1,H,W,C)).
    method "gather": atomic-free two-pass kernels (bit-reproducible, C = 32 / 16); "scatter": float atomics."""
    lib = _lib.load()
    H, W, Cc = ref.shape
    n_src, D = transforms.shape[0], transforms.shape[1]
    need = lib.mvs_cost_volume_bwd_workspace_bytes(n_src + 1, D, H, W, Cc) if method == "gather" else 0
    if need:
        ws = _CVWS.get(ref.device)
        if ws is None or ws.numel() < need:
            ws = _CVWS[ref.device] = torch.empty(need, device=ref.device, dtype=torch.uint8)
        g_ref, g_src = torch.empty_like(ref), torch.empty_like(src)
        _lib.check(lib.mvs_cost_volume_bwd_gather_f32(
            _lib.ptr(ref), _lib.ptr(src), _lib.ptr(transforms), n_src + 1, D, H, W, Cc, _lib.ptr(g1), _lib.ptr(g2),
            C.c_void_p(ws.data_ptr()), ws.numel(), _lib.ptr(g_ref), _lib.ptr(g_src), _lib.stream_ptr()),
            "mvs_cost_volume_bwd_gather_f32")
        return g_ref, g_src
    g_ref = torch.zeros_like(ref)
    g_src = torch.zeros_like(src)
    _lib.check(lib.mvs_cost_volume_bwd_f32(_lib.ptr(ref), _lib.ptr(src), _lib.ptr(transforms), n_src + 1, D, H, W, Cc,
                                           _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(g_ref), _lib.ptr(g_src),
                                           _lib.stream_ptr()), "mvs_cost_volume_bwd_f32")
    return g_ref, g_src


def _pad_channels(x, w, axis, to=16):
    """The MFMA convolutions tile Cin in 16s: zero-pad an 8-channel gradient (and the kernel's Cin axis)."""
    cin = x.shape[-1]
    if cin >= to:
        return x, w
    xp = torch.zeros(x.shape[:-1] + (to,), device=x.device, dtype=x.dtype)
    xp[..., :cin] = x
    shp = list(w.shape); shp[axis] = to
    wp = torch.zeros(shp, device=w.device, dtype=w.dtype)
    wp.narrow(axis, 0, cin).copy_(w)
    return xp, wp


def conv_s1_input_grad(g_y, w):
    """Input gradient of y = conv3d(x, w (3,3,3,Cin,Cout), stride 1): conv3d(g_y, flip(w)^T)."""
    wt = w.flip(0, 1, 2).permute(0, 1, 2, 4, 3).contiguous()          # (3,3,3,Cout,Cin) read as Cin'=Cout, Cout'=Cin
    if g_y.shape[-1] == 8 and wt.shape[4] != 32:          # 8 -> 32 has its own kernel (conv3d_k8.hip: taps paired along w)
        g_y, wt = _pad_channels(g_y, wt, 3)
    return conv3d(g_y, wt, 1)


def conv_s2_input_grad(g_y, w):
    """Input gradient of a stride-2 SAME convolution = conv3d_transpose with the same kernel array."""
    return conv3d(g_y, w, transpose=True)


def deconv_input_grad(g_y, w):
    """Input gradient of conv3d_transpose (w (3,3,3,Cout,Cin)) = stride-2 convolution with the same array."""
    if g_y.shape[-1] == 8:
        g_y, w = _pad_channels(g_y, w, 3)
    return conv3d(g_y, w, 2)


# ------------------------------------------------------------------------------------------------
# RegNetUS0 with saved activations
# ------------------------------------------------------------------------------------------------

def regnet_forward_train(cost, p: Dict[str, Dict[str, torch.Tensor]], sync: Optional[SyncBN] = None):
    """cost (D,H,W,32); p[name] = {'w', 'gamma', 'beta'}.  Returns (reg (D,H,W), saved).  `sync`: cross-replica
    BatchNorm statistics (SyncBN)."""
    dev = cost.device
    y, st, aff = {}, {}, {}
    world = sync.world if sync is not None else 1
    # every layer's (2, Cout) float64 BatchNorm sums from ONE zeroed slab (round 6: one fill instead of eleven)
    slab = torch.zeros(sum(2 * max(v["w"].shape[3], v["w"].shape[4]) for v in p.values()), device=dev, dtype=torch.float64)
    used = [0]

    def zeros2(cout):
        o = used[0]
        used[0] += 2 * cout
        return slab[o:o + 2 * cout].view(2, cout)

    def layer(name, x, stride=1, x_aff=None, skip=None, skip_aff=None, transpose=False):
        cout = p[name]["w"].shape[3] if transpose else p[name]["w"].shape[4]
        s = zeros2(cout)
        out = conv3d(x, p[name]["w"], stride, x_aff, skip, skip_aff, s, transpose)
        if world > 1:
            sync.all_reduce(s)
        y[name], st[name] = out, s
        aff[name] = bn_finalize(s, (out.numel() // cout) * world, p[name]["gamma"], p[name]["beta"])

    if cost.shape[3] == 32 and p["3dconv0_1"]["w"].shape[4] == 8 and not (cost.shape[0] | cost.shape[1] | cost.shape[2]) & 1:
        # both consumers of the cost volume in one pass over it (mvs_conv3d_pair_f32)
        s01, s10 = zeros2(8), zeros2(16)
        y["3dconv0_1"], y["3dconv1_0"] = conv3d_pair(cost, p["3dconv0_1"]["w"], p["3dconv1_0"]["w"], s01, s10)
        for nm, s in (("3dconv0_1", s01), ("3dconv1_0", s10)):
            if world > 1:
                sync.all_reduce(s)
            st[nm] = s
            aff[nm] = bn_finalize(s, (y[nm].numel() // y[nm].shape[3]) * world, p[nm]["gamma"], p[nm]["beta"])
        fused = True
    else:
        fused = False
        layer("3dconv1_0", cost, 2)
    layer("3dconv2_0", y["3dconv1_0"], 2, aff["3dconv1_0"])
    layer("3dconv3_0", y["3dconv2_0"], 2, aff["3dconv2_0"])
    if not fused:
        layer("3dconv0_1", cost, 1)
    layer("3dconv1_1", y["3dconv1_0"], 1, aff["3dconv1_0"])
    layer("3dconv2_1", y["3dconv2_0"], 1, aff["3dconv2_0"])
    layer("3dconv3_1", y["3dconv3_0"], 1, aff["3dconv3_0"])
    layer("3dconv4_0", y["3dconv3_1"], 2, aff["3dconv3_1"], transpose=True)
    layer("3dconv5_0", y["3dconv4_0"], 2, aff["3dconv4_0"], y["3dconv2_1"], aff["3dconv2_1"], transpose=True)
    layer("3dconv6_0", y["3dconv5_0"], 2, aff["3dconv5_0"], y["3dconv1_1"], aff["3dconv1_1"], transpose=True)
    reg = conv3d(y["3dconv6_0"], p["3dconv6_2"]["w"], 1, aff["3dconv6_0"], y["3dconv0_1"], aff["3dconv0_1"])
    return reg[..., 0], (cost, y, st, aff, sync)


def regnet_backward(saved, p, g_reg):
    """g_reg (D,H,W) -> (grads {name: {'w','gamma','beta'}}, g_cost_a, g_cost_b): the cost volume has two
    consumers (3dconv0_1, 3dconv1_0); their input gradients are returned separately and summed on load by
    the cost-volume backward."""
    cost, y, st, aff, sync = saved
    G: Dict[str, Dict[str, torch.Tensor]] = {}
    g_reg = g_reg.contiguous()[..., None]
    nsl = _lib.load().mvs_bn_bwd_sum_slots()
    slab = torch.zeros(nsl * sum(2 * y[n_].shape[-1] for n_ in st), device=g_reg.device, dtype=torch.float64)
    used = [0]

    def bn_bwd(name, g1, g2=None):
        n_ = nsl * 2 * y[name].shape[-1]
        sums = slab[used[0]:used[0] + n_]
        used[0] += n_
        g_y, gg, gb = bn_relu_bwd(y[name], st[name], aff[name], p[name]["gamma"], g1, g2, sync=sync, sums=sums)
        G[name] = {"gamma": gg, "beta": gb}
        return g_y

    act = lambda n: bn_relu(y[n], aff[n])
    # 3dconv6_2: conv 8 -> 1 on  s6 = a(6_0) + a(0_1), no BN
    s6 = bn_relu(y["3dconv6_0"], aff["3dconv6_0"], y["3dconv0_1"], aff["3dconv0_1"])
    G["3dconv6_2"] = {"w": conv3d_wgrad(s6, g_reg, 1)}
    g_s6 = conv_s1_input_grad(g_reg, p["3dconv6_2"]["w"])
    del s6
    # 3dconv6_0: deconv 16 -> 8 on  s5 = a(5_0) + a(1_1)
    g_y = bn_bwd("3dconv6_0", g_s6)
    s5 = bn_relu(y["3dconv5_0"], aff["3dconv5_0"], y["3dconv1_1"], aff["3dconv1_1"])
    G["3dconv6_0"]["w"] = conv3d_wgrad(g_y, s5, 2)
    g_s5 = deconv_input_grad(g_y, p["3dconv6_0"]["w"])
    # 3dconv0_1: conv 32 -> 8 on the cost volume
    g_y = bn_bwd("3dconv0_1", g_s6)
    # roles swapped (rows = the 8-channel gradient with its taps, columns = the 32 volume channels): the kernel packs 8
    # taps per MFMA row tile; R(tap, c8, c32) = dW(2 - tap, c32, c8)
    G["3dconv0_1"]["w"] = conv3d_wgrad(g_y, cost, 1).flip(0, 1, 2).permute(0, 1, 2, 4, 3).contiguous()
    g_cost_a = conv_s1_input_grad(g_y, p["3dconv0_1"]["w"])
    del g_s6
    # 3dconv5_0: deconv 32 -> 16 on  s4 = a(4_0) + a(2_1)
    g_y = bn_bwd("3dconv5_0", g_s5)
    s4 = bn_relu(y["3dconv4_0"], aff["3dconv4_0"], y["3dconv2_1"], aff["3dconv2_1"])
    G["3dconv5_0"]["w"] = conv3d_wgrad(g_y, s4, 2)
    g_s4 = deconv_input_grad(g_y, p["3dconv5_0"]["w"])
    # 3dconv1_1: conv 16 -> 16 on a(1_0)
    a10 = act("3dconv1_0")
    g_y = bn_bwd("3dconv1_1", g_s5)
    G["3dconv1_1"]["w"] = conv3d_wgrad(a10, g_y, 1)
    g_a10_b = conv_s1_input_grad(g_y, p["3dconv1_1"]["w"])
    # 3dconv4_0: deconv 64 -> 32 on a(3_1)
    g_y = bn_bwd("3dconv4_0", g_s4)
    G["3dconv4_0"]["w"] = conv3d_wgrad(g_y, act("3dconv3_1"), 2)
    g_a31 = deconv_input_grad(g_y, p["3dconv4_0"]["w"])
    # 3dconv2_1: conv 32 -> 32 on a(2_0)
    a20 = act("3dconv2_0")
    g_y = bn_bwd("3dconv2_1", g_s4)
    G["3dconv2_1"]["w"] = conv3d_wgrad(a20, g_y, 1)
    g_a20_b = conv_s1_input_grad(g_y, p["3dconv2_1"]["w"])
    # 3dconv3_1: conv 64 -> 64 on a(3_0)
    g_y = bn_bwd("3dconv3_1", g_a31)
    G["3dconv3_1"]["w"] = conv3d_wgrad(act("3dconv3_0"), g_y, 1)
    g_a30 = conv_s1_input_grad(g_y, p["3dconv3_1"]["w"])
    # 3dconv3_0: conv stride 2, 32 -> 64 on a(2_0)
    g_y = bn_bwd("3dconv3_0", g_a30)
    G["3dconv3_0"]["w"] = conv3d_wgrad(a20, g_y, 2)
    g_a20_a = conv_s2_input_grad(g_y, p["3dconv3_0"]["w"])
    # 3dconv2_0: conv stride 2, 16 -> 32 on a(1_0)
    g_y = bn_bwd("3dconv2_0", g_a20_a, g_a20_b)
    G["3dconv2_0"]["w"] = conv3d_wgrad(a10, g_y, 2)
    g_a10_a = conv_s2_input_grad(g_y, p["3dconv2_0"]["w"])
    # 3dconv1_0: conv stride 2, 32 -> 16 on the cost volume
    g_y = bn_bwd("3dconv1_0", g_a10_a, g_a10_b)
    G["3dconv1_0"]["w"] = conv3d_wgrad(cost, g_y, 2)
    g_cost_b = conv_s2_input_grad(g_y, p["3dconv1_0"]["w"])
    return G, g_cost_a, g_cost_b


# ------------------------------------------------------------------------------------------------
# autograd seam
# ------------------------------------------------------------------------------------------------

def flatten_params(p) -> List[torch.Tensor]:
    out = []
    for n in REGNET_ORDER:
        out.append(p[n]["w"])
        if n in BN_LAYERS:
            out += [p[n]["gamma"], p[n]["beta"]]
    return out


def unflatten_params(flat) -> Dict[str, Dict[str, torch.Tensor]]:
    p, i = {}, 0
    for n in REGNET_ORDER:
        p[n] = {"w": flat[i]}; i += 1
        if n in BN_LAYERS:
            p[n]["gamma"], p[n]["beta"] = flat[i], flat[i + 1]; i += 2
    return p


class PlaneSweepDepth(torch.autograd.Function):
    """features (N,H,W,C) [view 0 = reference], transforms (N-1,D,8) -> depth map (H,W); differentiable
    w.r.t. the features and the RegNetUS0 parameters (cameras are data).  The probability map carries a gradient too
    (it is an input channel of the refinement network, model.py:753-811); its bucket indices do not."""

    @staticmethod
    def forward(ctx, features, transforms, depth_start, depth_interval, inverse_depth, sync, into, *flat):
        """`into`: None, or one tensor per entry of `flat` (the variables' slices of a flat gradient buffer): the backward then adds
        the parameter gradients there itself -- one launch for all of them (mvs_add_many_f32) -- and hands autograd nothing."""
        from .model import softargmin_prob
        ctx.into = into
        p = unflatten_params([t.detach() for t in flat])
        features = features.detach().contiguous()
        cost = cost_volume(features[0], features[1:], transforms, variant="eager")      # model.py:330-332
        reg, saved = regnet_forward_train(cost, p, sync)
        depth, prob = softargmin_prob(reg, depth_start, depth_interval, inverse_depth)
        ctx.saved = (features, transforms, reg, saved, p)
        ctx.scalars = (float(depth_start), float(depth_interval), bool(inverse_depth))
        return depth, prob

    @staticmethod
    def backward(ctx, g_depth, g_prob):
        features, transforms, reg, saved, p = ctx.saved
        start, interval, inverse = ctx.scalars
        g_reg = softargmin_bwd(reg, g_depth, start, interval, inverse, g_prob)
        G, ga, gb = regnet_backward(saved, p, g_reg)
        g_ref, g_src = cost_volume_bwd(features[0], features[1:], transforms, ga, gb)
        g_feat = torch.cat([g_ref[None], g_src], 0)
        flat = []
        for n in REGNET_ORDER:
            flat.append(G[n]["w"])
            if n in BN_LAYERS:
                flat += [G[n]["gamma"], G[n]["beta"]]
        ctx.saved = None
        if ctx.into is not None:
            import ctypes as C
            flat = [g.contiguous() for g in flat]
            n = len(flat)
            _lib.check(_lib.load().mvs_add_many_f32(n, (C.c_void_p * n)(*[g.data_ptr() for g in flat]),
                                                    (C.c_void_p * n)(*[t.data_ptr() for t in ctx.into]),
                                                    (C.c_longlong * n)(*[g.numel() for g in flat]), _lib.stream_ptr()), "mvs_add_many_f32")
            return (g_feat, None, None, None, None, None, None) + (None,) * n
        return (g_feat, None, None, None, None, None, None) + tuple(flat)


def plane_sweep_depth(features, transforms, depth_start, depth_interval, params, inverse_depth=False, sync=None,
                      accumulate_into_grads=False):
    """`sync`: a SyncBN for cross-replica BatchNorm statistics under torch.distributed (default: per-replica).
    `accumulate_into_grads`: the leaves carry pre-allocated contiguous `.grad` tensors (train.FlatParameters) and the backward adds
    the parameter gradients into them itself, in one launch; autograd then sees no gradient for them."""
    flat = flatten_params(params)
    into = None
    if accumulate_into_grads:
        into = [p.grad for p in flat]
        if any(g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != p.shape for g, p in zip(into, flat)):
            raise ValueError("accumulate_into_grads needs a contiguous float32 .grad of the variable's shape on every leaf")
    return PlaneSweepDepth.apply(features, transforms, depth_start, depth_interval, inverse_depth, sync, into, *flat)
