"""Host-side mirror of mvsnet/model.py: depth inference by plane sweep on the HIP library.

Public functions keep the reference's names, argument order and meaning
(`inference_mem` model.py:374, `inference` :257, `inference_winner_take_all` :601,
`get_probability_map` :20); the reference's hidden FLAGS inputs (view_num, batch_size, height,
width) become explicit keyword arguments, and tensors are torch device
tensors.  All arithmetic of the hot path runs in libmvsnet_hip.so; torch only owns memory,
streams and the 2D feature extractor.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .feature_net import UNetDS2GN
from .homography_warping import homography_transforms

REGNET_ORDER = ("3dconv1_0", "3dconv2_0", "3dconv3_0", "3dconv0_1", "3dconv1_1", "3dconv2_1",
                "3dconv3_1", "3dconv4_0", "3dconv5_0", "3dconv6_0", "3dconv6_2")
GRU_CELL_KEYS = ("gates_w", "gates_b", "reset_gamma", "reset_beta", "update_gamma", "update_beta",
                 "out_w", "out_b", "out_gamma", "out_beta")
BN_EPSILON = 1e-5      # mvsnet/cnn_wrapper/network.py:55


def _dev(a, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(device).contiguous()


def pad_regnet_params(params):
    """Zero-pads RegNetUS0 parameters of a narrower network_mode (lite / ultralite: base_filter 4 / 2,
    mvsnetworks.py:126-127; also 'semilite-py3', base_filter 6 -- the reference's own 'semilite' is base 8 at
    run time, synthetic.base_divisor) to the channel counts of 'normal' (base 8, 32-channel volume), the
    shapes the fp32-MFMA kernels tile.  Padded kernels are zero and padded gamma / beta are zero, so the
    extra channels carry exact zeros through conv, BatchNorm (0 * xhat + 0) and ReLU and the real channels
    see the same sums: same function, ~8x (lite) to ~25x (base_filter 6) faster than the shape-generic VALU
    kernels those modes would otherwise fall back to."""
    from .synthetic import make_regnet_params
    target = make_regnet_params("normal")
    out = {}
    for n in REGNET_ORDER:
        q = {}
        for key, a in params[n].items():
            a = np.asarray(a, np.float32)
            shape = np.asarray(target[n][key]).shape
            if any(sa > st for sa, st in zip(a.shape, shape)):
                return None                                   # wider than 'normal' (fat modes): no padding
            z = np.zeros(shape, np.float32)
            z[tuple(slice(0, k) for k in a.shape)] = a
            q[key] = z
        out[n] = q
    return out


class RegNetWeights:
    """RegNetUS0 parameters resident on the device, TensorFlow variable layouts
    (conv (3,3,3,Cin,Cout), conv3d_transpose (3,3,3,Cout,Cin), BN gamma/beta).  `cin_native` is the
    channel count of the feature maps the parameters were trained for; `cin` what the kernels run
    (32 after padding narrower modes, see pad_regnet_params)."""

    def __init__(self, params: Dict[str, dict], device="cuda", pad_to_mfma=True):
        self.device = torch.device(device)
        self.cin_native = int(np.asarray(params[REGNET_ORDER[0]]["w"]).shape[3])
        if pad_to_mfma and self.cin_native < 32:
            padded = pad_regnet_params(params)
            if padded is not None:
                params = padded
        self.w = [_dev(params[n]["w"], device) for n in REGNET_ORDER]
        self.gamma = [_dev(params[n]["gamma"], device) for n in REGNET_ORDER[:-1]]
        self.beta = [_dev(params[n]["beta"], device) for n in REGNET_ORDER[:-1]]
        self.cin = int(self.w[0].shape[3])
        self.base = int(self.w[3].shape[4])           # 3dconv0_1: Cin -> base_filter
        self.w_ptrs = _lib.ptr_array(self.w)
        self.g_ptrs = _lib.ptr_array(self.gamma)
        self.b_ptrs = _lib.ptr_array(self.beta)
        # one-off re-layout of the kernels into the MFMA kernels' LDS order
        lib = _lib.load()
        n = lib.mvs_regnet_prepared_floats(self.cin, self.base)
        self.prepared = torch.empty(max(n, 1), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(lib.mvs_regnet_prepare_f32(self.w_ptrs, self.cin, self.base, _lib.ptr(self.prepared),
                                                  _lib.stream_ptr()), "mvs_regnet_prepare_f32")


class GRUWeights:
    """ConvGRU x3 + prob_conv parameters on the device (convgru.py:82-122, model.py:701)."""

    def __init__(self, params: Dict[str, dict], device="cuda"):
        self.device = torch.device(device)
        self.tensors = []
        for cell in ("gru1", "gru2", "gru3"):
            for k in GRU_CELL_KEYS:
                self.tensors.append(_dev(params[cell][k], device))
        self.tensors.append(_dev(params["prob_w"], device))
        self.tensors.append(_dev(params["prob_b"], device))
        self.filters = tuple(int(params[c]["out_b"].shape[0]) for c in ("gru1", "gru2", "gru3"))
        self.cin = int(params["gru1"]["gates_w"].shape[2]) - self.filters[0]
        self.ptrs = _lib.ptr_array(self.tensors)


@dataclass
class MVSNetWeights:
    """Everything `inference_mem` / `inference_winner_take_all` need besides the inputs."""
    network_mode: str = "normal"
    unet: Optional[UNetDS2GN] = None
    regnet: Optional[RegNetWeights] = None
    gru: Optional[GRUWeights] = None
    refine: Optional[object] = None          # refine.RefineNet, only for --refinement (model.py:753-811)

    @classmethod
    def from_numpy(cls, network_mode="normal", unet=None, regnet=None, gru=None, device="cuda",
                   refine=None, refine_type="original", extractor="hip"):
        """`extractor`: "hip" = the 2D towers on the HIP library (feature_net_hip.HipUNetDS2GN, SURVEY 8f
        f2, ~12x the PyTorch/MIOpen module at 5 x 512 x 640), "torch" = feature_net.UNetDS2GN."""
        from .refine import RefineNet
        if extractor not in ("hip", "torch"):
            raise ValueError("extractor must be 'hip' or 'torch'")
        Extractor = UNetDS2GN
        if extractor == "hip":
            # the HIP towers tile output channels in 8s (base_filter 8 = 'normal' and wider); the narrower modes'
            # 4- and 2-channel layers also have GroupNorm groups smaller than 8 channels, which zero-padding would
            # change, so they stay on the PyTorch module
            narrow = unet is not None and any(np.asarray(p["w"]).shape[-1] % 8 for name, p in unet.items()
                                              if name in ("2dconv0_1", "2dconv1_0"))
            if not narrow:
                from .feature_net_hip import HipUNetDS2GN as Extractor
            else:
                Extractor = lambda p, d: UNetDS2GN(p, d, hip_group_norm=True)
        return cls(network_mode,
                   Extractor(unet, device) if unet is not None else None,
                   RegNetWeights(regnet, device) if regnet is not None else None,
                   GRUWeights(gru, device) if gru is not None else None,
                   RefineNet(refine, refine_type, device) if refine is not None else None)


# ------------------------------------------------------------------------------------------------
# single-op wrappers (each = one C-ABI call)
# ------------------------------------------------------------------------------------------------


def cost_volume(ref_feature, src_features, transforms, d_begin=0, d_count=None, variant="mem",
                negate=False, border="zeros", out=None):
    """Fused warp + variance (model.py:422-463 / :315-334).  ref (H,W,C), src (N-1,H,W,C),
    transforms (N-1,D,8) -> (d_count,H,W,C)."""
    lib = _lib.load()
    ref = _lib.f32(ref_feature, "ref_feature")
    src = _lib.f32(src_features, "src_features")
    H, W, Cc = ref.shape
    n_src, D = transforms.shape[0], transforms.shape[1]
    if d_count is None:
        d_count = D - d_begin
    if out is None:
        out = torch.empty((d_count, H, W, Cc), device=ref.device, dtype=torch.float32)
    _lib.check(lib.mvs_cost_volume_f32(
        _lib.ptr(ref), _lib.ptr(src), _lib.ptr(_lib.f32(transforms)), n_src + 1, D, d_begin,
        d_count, H, W, Cc, 0 if variant == "mem" else 1, int(bool(negate)),
        0 if border == "zeros" else 1, _lib.ptr(out), _lib.stream_ptr()), "mvs_cost_volume_f32")
    return out


def conv3d(x, w, stride=1, x_affine=None, skip=None, skip_affine=None, stats=None, transpose=False):
    """One 3x3x3 SAME conv / transposed conv (network.py:171-215,300-329) with the producer's
    BatchNorm+ReLU (and an optional additive skip) applied on load.  Returns raw output."""
    lib = _lib.load()
    D, H, W, Cin = x.shape
    Cout = w.shape[3] if transpose else w.shape[4]
    xs, xb = x_affine if x_affine is not None else (None, None)
    ss, sb = skip_affine if skip_affine is not None else (None, None)
    if transpose:
        y = torch.empty((2 * D, 2 * H, 2 * W, Cout), device=x.device, dtype=torch.float32)
        _lib.check(lib.mvs_deconv3d_f32(_lib.ptr(x), _lib.ptr(xs), _lib.ptr(xb), _lib.ptr(skip),
                                        _lib.ptr(ss), _lib.ptr(sb), _lib.ptr(w), D, H, W, Cin, Cout,
                                        _lib.ptr(y), _lib.ptr(stats), _lib.stream_ptr()),
                   "mvs_deconv3d_f32")
    else:
        o = lambda n: -(-n // stride)
        y = torch.empty((o(D), o(H), o(W), Cout), device=x.device, dtype=torch.float32)
        _lib.check(lib.mvs_conv3d_f32(_lib.ptr(x), _lib.ptr(xs), _lib.ptr(xb), _lib.ptr(skip),
                                      _lib.ptr(ss), _lib.ptr(sb), _lib.ptr(w), D, H, W, Cin, Cout,
                                      stride, _lib.ptr(y), _lib.ptr(stats), _lib.stream_ptr()),
                   "mvs_conv3d_f32")
    return y


def conv3d_pair(x, w1, w2, stats1=None, stats2=None):
    """3dconv0_1 (32->8, stride 1) and 3dconv1_0 (32->16, stride 2) of RegNetUS0 in one pass over the
    cost volume (mvsnetworks.py:130-134).  Returns the two raw outputs."""
    lib = _lib.load()
    D, H, W, Cin = x.shape
    C1, C2 = w1.shape[4], w2.shape[4]
    y1 = torch.empty((D, H, W, C1), device=x.device, dtype=torch.float32)
    y2 = torch.empty((D // 2, H // 2, W // 2, C2), device=x.device, dtype=torch.float32)
    _lib.check(lib.mvs_conv3d_pair_f32(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), D, H, W, Cin, C1, C2,
                                       _lib.ptr(y1), _lib.ptr(stats1), _lib.ptr(y2), _lib.ptr(stats2),
                                       _lib.stream_ptr()), "mvs_conv3d_pair_f32")
    return y1, y2


def bn_finalize(stats, count, gamma, beta, eps=BN_EPSILON):
    """(2,C) float64 sums -> per-channel (scale, shift) of training-mode BN (network.py:496-506)."""
    lib = _lib.load()
    Cn = gamma.shape[0]
    scale = torch.empty(Cn, device=gamma.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    _lib.check(lib.mvs_bn_finalize_f32(_lib.ptr(stats), Cn, float(count), _lib.ptr(gamma),
                                       _lib.ptr(beta), float(eps), _lib.ptr(scale), _lib.ptr(shift),
                                       _lib.stream_ptr()), "mvs_bn_finalize_f32")
    return scale, shift


def regnet_us0(cost_volume_, weights: RegNetWeights, workspace=None, out=None):
    """RegNetUS0 (mvsnetworks.py:122-158): (D,H,W,Cin) -> filtered cost volume (D,H,W); a 5-D (B,D,H,W,Cin) input is a
    batch whose BatchNorm layers share their statistics over (B,D,H,W) as in the reference (network.py:496-506)."""
    lib = _lib.load()
    if cost_volume_.dim() == 5:
        B, D, H, W, Cin = cost_volume_.shape
        if Cin == weights.cin_native and Cin != weights.cin:
            cost_volume_ = torch.nn.functional.pad(cost_volume_, (0, weights.cin - Cin)).contiguous()
            Cin = weights.cin
        if Cin != weights.cin:
            raise _lib.MvsnetHipError("cost volume has %d channels, weights expect %d" % (Cin, weights.cin))
        need = B * lib.mvs_regnet_workspace_bytes(D, H, W, Cin, weights.base)
        if workspace is None:
            workspace = torch.empty(need, device=cost_volume_.device, dtype=torch.uint8)
        if out is None:
            out = torch.empty((B, D, H, W), device=cost_volume_.device, dtype=torch.float32)
        _lib.check(lib.mvs_regnet_us0_batch_f32(
            _lib.ptr(_lib.f32(cost_volume_)), B, D, H, W, Cin, weights.base, weights.w_ptrs,
            _lib.ptr(weights.prepared), weights.g_ptrs, weights.b_ptrs, BN_EPSILON,
            C.c_void_p(workspace.data_ptr()), workspace.numel(), _lib.ptr(out), _lib.stream_ptr()),
            "mvs_regnet_us0_batch_f32")
        return out
    D, H, W, Cin = cost_volume_.shape
    if Cin == weights.cin_native and Cin != weights.cin:      # narrower mode running zero-padded (RegNetWeights)
        cost_volume_ = torch.nn.functional.pad(cost_volume_, (0, weights.cin - Cin)).contiguous()
        Cin = weights.cin
    if Cin != weights.cin:
        raise _lib.MvsnetHipError("cost volume has %d channels, weights expect %d" % (Cin, weights.cin))
    need = lib.mvs_regnet_workspace_bytes(D, H, W, Cin, weights.base)
    if workspace is None:
        workspace = torch.empty(need, device=cost_volume_.device, dtype=torch.uint8)
    if out is None:
        out = torch.empty((D, H, W), device=cost_volume_.device, dtype=torch.float32)
    _lib.check(lib.mvs_regnet_us0_prepared_f32(
        _lib.ptr(_lib.f32(cost_volume_)), D, H, W, Cin, weights.base, weights.w_ptrs,
        _lib.ptr(weights.prepared), weights.g_ptrs, weights.b_ptrs, BN_EPSILON,
        C.c_void_p(workspace.data_ptr()), workspace.numel(), _lib.ptr(out), _lib.stream_ptr()),
        "mvs_regnet_us0_prepared_f32")
    return out


def softargmin_prob(filtered_cost_volume, depth_start, depth_interval, inverse_depth=False,
                    depth_out=None, prob_out=None):
    """model.py:471-498: softmax(-reg) over depth, soft-argmin and the 4-bucket probability."""
    lib = _lib.load()
    reg = _lib.f32(filtered_cost_volume)
    D, H, W = reg.shape
    if depth_out is None:
        depth_out = torch.empty((H, W), device=reg.device, dtype=torch.float32)
    if prob_out is None:
        prob_out = torch.empty((H, W), device=reg.device, dtype=torch.float32)
    _lib.check(lib.mvs_softargmin_prob_f32(_lib.ptr(reg), D, H, W, float(depth_start),
                                           float(depth_interval), int(bool(inverse_depth)),
                                           _lib.ptr(depth_out), _lib.ptr(prob_out),
                                           _lib.stream_ptr()), "mvs_softargmin_prob_f32")
    return depth_out, prob_out


def get_probability_map(cv, depth_map, depth_start, depth_interval, inverse_depth=False,
                        num_buckets=4):
    """get_probability_map_slice (model.py:45-144) on an explicit probability volume
    cv (D,H,W) and depth_map (H,W).  Small torch glue; the fused kernel above is what
    inference_mem uses."""
    D = cv.shape[0]
    if inverse_depth:
        end = depth_start + (D - 1.0) * depth_interval
        inv_s, inv_e = 1.0 / depth_start, 1.0 / end
        idx = (1.0 / depth_map - inv_e) / ((inv_s - inv_e) / (D - 1.0))
        l0 = D - torch.ceil(idx).long() - 1
        r0 = D - torch.floor(idx).long() - 1
    else:
        idx = (depth_map - depth_start) / depth_interval
        l0 = torch.floor(idx).long()
        r0 = torch.ceil(idx).long()
    l0 = l0.clamp(0, D - 1); r0 = r0.clamp(0, D - 1)
    l1 = (l0 - 1).clamp(0, D - 1); r1 = (r0 + 1).clamp(0, D - 1)
    g = lambda i: torch.gather(cv, 0, i[None])[0]
    out = g(l0) + g(r0)
    if num_buckets == 4:
        out = out + (g(l1) + g(r1))
    return out


# ------------------------------------------------------------------------------------------------
# plans: pre-allocated buffers for one (D,H,W,C,N) problem; launch-only afterwards
# ------------------------------------------------------------------------------------------------


class DepthPlan:
    """Owns every device buffer of one features->depth problem so that `run` only enqueues
    kernels (no allocation, no synchronisation): usable under hipGraph capture and on several
    streams (one plan per stream)."""

    def __init__(self, view_num, depth_num, height, width, channels, weights: MVSNetWeights,
                 regularization="3DCNN", device="cuda", views=1):
        """views > 1 (GRU only): that many independent reference views per sweep (run_gru_batch)."""
        lib = _lib.load()
        self.N, self.D, self.H, self.W, self.C = view_num, depth_num, height, width, channels
        self.weights = weights
        self.regularization = regularization
        self.views = int(views)
        if not (1 <= self.views <= 8):
            raise ValueError("views must be 1..8 (mvs_gru_wta_batch_f32 takes at most 8 reference views per sweep)")
        if self.views != 1 and regularization != "GRU":
            raise NotImplementedError("views > 1 is the recurrent sweep's batch of reference views")
        dev = torch.device(device)
        f = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)
        self.transforms_v = f(self.views, view_num - 1, depth_num, 8)
        self.depth_v = f(self.views, height, width)
        self.prob_v = f(self.views, height, width)
        self.transforms, self.depth, self.prob = self.transforms_v[0], self.depth_v[0], self.prob_v[0]
        if regularization == "3DCNN":
            # narrower network modes run zero-padded to the MFMA kernels' 32-channel volume (RegNetWeights)
            self.C = channels = max(channels, weights.regnet.cin)
            self.fpad = torch.zeros((view_num, height, width, channels), device=dev, dtype=torch.float32)
            self.cost = f(depth_num, height, width, channels)
            self.reg = f(depth_num, height, width)
            nbytes = lib.mvs_regnet_workspace_bytes(depth_num, height, width, channels, weights.regnet.base)
            self.workspace = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        elif regularization == "GRU":
            f1, f2, f3 = weights.gru.filters
            nbytes = lib.mvs_gru_workspace_bytes(height, width, channels, f1, f2, f3)
            self.workspace = torch.empty(nbytes * self.views, device=dev, dtype=torch.uint8)
            # side streams + calibration for the current stream: the one call that synchronises.  The set is shared by the plans of a
            # stream and released with the last of them (close() or garbage collection).
            tok = _lib.gru_prepare()
            self._gru_keys = [tok] if tok is not None else []
            self._finalizer = weakref.finalize(self, DepthPlan._drop_sets, self._gru_keys)
        else:
            raise NotImplementedError(regularization)      # predictlib.py:97-98

    @staticmethod
    def _drop_sets(keys):
        while keys:
            _lib.gru_unref(keys.pop())

    def close(self):
        """Gives the stream sets of the recurrent sweep back (mvs_gru_release when this was their last user).  The plan stays
        usable: a later run_gru prepares (a share of) a set again and re-arms the finalizer."""
        if getattr(self, "_finalizer", None) is not None:
            self._finalizer()
            self._finalizer = None

    def _gru_prepare_here(self):
        """A plan may run on another stream than it was built on: that stream gets (a share of) a set too.  One token per
        (device, stream); a stream that was refused a set (MVS_E_NO_SLOT, remembered in _lib) adds nothing."""
        if torch.cuda.is_current_stream_capturing():
            return
        key = (torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))
        if getattr(self, "_finalizer", None) is None:      # after close(): the list is empty and nothing would release what follows
            self._gru_keys = []
            self._finalizer = weakref.finalize(self, DepthPlan._drop_sets, self._gru_keys)
        if key not in [k[:2] for k in self._gru_keys if k is not None]:
            tok = _lib.gru_prepare()
            if tok is not None:
                self._gru_keys.append(tok)

    def set_cameras(self, cams, depth_start, depth_interval, depth_end, inverse_depth, view=0):
        lib = _lib.load()
        _lib.check(lib.mvs_homography_transforms_f32(
            _lib.ptr(_lib.f32(cams)), self.N, self.D, float(depth_start), float(depth_interval),
            float(depth_end), int(bool(inverse_depth)), None, _lib.ptr(self.transforms_v[view]),
            _lib.stream_ptr()), "mvs_homography_transforms_f32")

    def run_3dcnn(self, features, depth_start, depth_interval, inverse_depth=False, variant="mem"):
        """features (N,H,W,C): cost volume -> RegNetUS0 -> soft-argmin.  Cameras must be set."""
        if features.shape[-1] != self.C:
            self.fpad[..., :features.shape[-1]].copy_(features)       # padded channels stay zero
            features = self.fpad
        cost_volume(features[0], features[1:], self.transforms, 0, self.D, variant, out=self.cost)
        regnet_us0(self.cost, self.weights.regnet, self.workspace, self.reg)
        softargmin_prob(self.reg, depth_start, depth_interval, inverse_depth, self.depth, self.prob)
        return self.depth, self.prob

    def run_depth(self, features, cams, depth_start, depth_interval, depth_end, inverse_depth=False, variant="mem"):
        """features (N,H,W,C), cams (N,2,4,4) -> (depth, prob): set_cameras + run_3dcnn as ONE library call
        (mvs_depth_from_features_f32: the homography launch also clears the BatchNorm sums)."""
        lib = _lib.load()
        if features.shape[-1] != self.C:
            self.fpad[..., :features.shape[-1]].copy_(features)       # padded channels stay zero
            features = self.fpad
        w = self.weights.regnet
        _lib.check(lib.mvs_depth_from_features_f32(
            _lib.ptr(_lib.f32(features)), _lib.ptr(_lib.f32(cams)), self.N, self.D, self.H, self.W, self.C, w.base,
            float(depth_start), float(depth_interval), float(depth_end), int(bool(inverse_depth)),
            0 if variant == "mem" else 1, w.w_ptrs, _lib.ptr(w.prepared), w.g_ptrs, w.b_ptrs, BN_EPSILON,
            _lib.ptr(self.transforms), _lib.ptr(self.cost), C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(),
            _lib.ptr(self.reg), _lib.ptr(self.depth), _lib.ptr(self.prob), _lib.stream_ptr()),
            "mvs_depth_from_features_f32")
        return self.depth, self.prob

    def run_gru(self, features, depth_values):
        lib = _lib.load()
        self._gru_prepare_here()
        g = self.weights.gru
        f1, f2, f3 = g.filters
        dv = (C.c_float * self.D)(*[float(v) for v in depth_values])
        _lib.check(lib.mvs_gru_wta_f32(
            _lib.ptr(features[0]), _lib.ptr(features[1:]), _lib.ptr(self.transforms), self.N, self.D,
            self.H, self.W, self.C, f1, f2, f3, g.ptrs, dv, C.c_void_p(self.workspace.data_ptr()),
            self.workspace.numel(), _lib.ptr(self.depth), _lib.ptr(self.prob), _lib.stream_ptr()),
            "mvs_gru_wta_f32")
        return self.depth, self.prob

    def run_gru_batch(self, features, depth_values):
        """The sweep for `len(features)` (<= self.views) independent reference views in the same launches
        (mvs_gru_wta_batch_f32): features[v] (N,H,W,C), depth_values[v] (D,); cameras set per view with
        set_cameras(..., view=v).  Returns (depth (n,H,W), prob (n,H,W))."""
        lib = _lib.load()
        self._gru_prepare_here()
        g = self.weights.gru
        f1, f2, f3 = g.filters
        n = len(features)
        if not (1 <= n <= self.views):
            raise ValueError("this plan holds %d view(s), got %d" % (self.views, n))
        keep = [(_lib.f32(f, "features")[0].contiguous(), f[1:].contiguous()) for f in features]    # alive until the launches are enqueued
        ref = _lib.ptr_array([k[0] for k in keep])
        src = _lib.ptr_array([k[1] for k in keep])
        tr = _lib.ptr_array([self.transforms_v[v] for v in range(n)])
        flat = [float(x) for dv in depth_values for x in dv]
        if len(flat) != n * self.D:
            raise ValueError("depth_values must hold %d x %d floats" % (n, self.D))
        dv = (C.c_float * len(flat))(*flat)
        _lib.check(lib.mvs_gru_wta_batch_f32(
            ref, src, tr, n, self.N, self.D, self.H, self.W, self.C, f1, f2, f3, g.ptrs, dv,
            C.c_void_p(self.workspace.data_ptr()), self.workspace.numel(), _lib.ptr(self.depth_v), _lib.ptr(self.prob_v),
            _lib.stream_ptr()), "mvs_gru_wta_batch_f32")
        return self.depth_v[:n], self.prob_v[:n]


_PLAN_CACHE: Dict[tuple, DepthPlan] = {}


def _plan(view_num, D, H, W, Cc, weights, regularization, device):
    key = (view_num, D, H, W, Cc, id(weights), regularization, str(device))
    p = _PLAN_CACHE.get(key)
    if p is None:
        if len(_PLAN_CACHE) > 4:
            _PLAN_CACHE.clear()
        p = _PLAN_CACHE[key] = DepthPlan(view_num, D, H, W, Cc, weights, regularization, device)
    return p


def _scalar(x):
    if torch.is_tensor(x):
        x = x.reshape(-1)
        if x.numel() != 1:
            raise NotImplementedError("batch_size > 1 is not supported (one reference view per call)")
        return float(x[0])
    a = np.asarray(x, dtype=np.float32).reshape(-1)
    if a.size != 1:
        raise NotImplementedError("batch_size > 1 is not supported (one reference view per call)")
    return float(a[0])


def _per_sample(x, batch):
    """depth_start / depth_interval / depth_end: a scalar or one value per sample (predictlib.set_shapes:190-197)."""
    if torch.is_tensor(x):
        a = x.detach().reshape(-1).to(torch.float32).cpu().numpy()
    else:
        a = np.asarray(x, dtype=np.float32).reshape(-1)
    if a.size == 1:
        return [float(a[0])] * batch
    if a.size != batch:
        raise ValueError("expected 1 or %d values, got %d" % (batch, a.size))
    return [float(v) for v in a]


def _batch_of(images, features, cams):
    t_ = features if features is not None else images
    if t_.dim() == 5:
        return int(t_.shape[0])
    return 1


def _sample(t_, b):
    """Sample b of a batched (5-D) tensor as a batch of one; a 4-D tensor is its own single sample."""
    return None if t_ is None else (t_[b:b + 1] if t_.dim() == 5 else t_)


def _features(images, cams, weights, features, view_num):
    """Runs the UNetDS2GN towers (model.py:392-406) unless precomputed features are given."""
    if features is None:
        if images.dim() != 5 or images.shape[0] != 1:
            raise NotImplementedError("images must be (1, view_num, H, W, 3); batch_size > 1 is not supported")
        if weights.unet is None:
            raise _lib.MvsnetHipError("weights.unet is required when features are not given")
        features = weights.unet(images[0, :view_num])
    elif features.dim() == 5:
        features = features[0]
    features = _lib.f32(features, "features")[:view_num].contiguous()
    if cams.dim() == 5:
        if cams.shape[0] != 1:
            raise NotImplementedError("batch_size > 1 is not supported")
        cams = cams[0]
    cams = cams[:view_num].to(device=features.device, dtype=torch.float32).contiguous()
    return features, cams


def inference_mem(images, cams, depth_num, depth_start, depth_interval, network_mode="normal",
                  is_master_gpu=True, training=True, trainable=True, inverse_depth=False, *,
                  weights: MVSNetWeights, view_num=None, features=None, variant="mem"):
    """mvsnet/model.py:374-502.  images (1,N,Himg,Wimg,3), cams (1,N,2,4,4), depth_start /
    depth_interval scalars or shape-(1,) -> (depth_map (1,H,W,1), prob_map (1,H,W,1)).

    `training` is accepted for signature parity; as in the reference's inference graph the
    BatchNorm layers use the statistics of the current volume (SURVEY.md 3.1 note), so only
    training=True semantics exist.  `features` (N,H,W,C) skips the 2D towers."""
    if not training:
        raise NotImplementedError("the reference inference graph runs BN with batch statistics "
                                  "(training=True); moving averages are never used")
    if weights.regnet is None:
        raise _lib.MvsnetHipError("weights.regnet is required for the 3DCNN regulariser")
    if view_num is None:
        view_num = (features if features is not None else images).shape[-4]
    D = int(depth_num)
    B = _batch_of(images, features, cams)
    if B > 1:
        # FLAGS.batch_size > 1 (model.py:350,431,479): per-sample towers, homographies, cost volumes and soft-argmin;
        # the regulariser's BatchNorm layers normalise over the whole batch (network.py:496-506).
        starts, intervals = _per_sample(depth_start, B), _per_sample(depth_interval, B)
        costs, Hh, Ww = [], None, None
        for b in range(B):
            f_b, c_b = _features(_sample(images, b), cams[b:b + 1], weights, _sample(features, b), view_num)
            _, Hh, Ww, Cc = f_b.shape
            end_b = float(np.float32(starts[b]) + (np.float32(D) - np.float32(1)) * np.float32(intervals[b]))   # model.py:378-379
            t8 = homography_transforms(c_b, D, starts[b], intervals[b], end_b, inverse_depth)
            costs.append(cost_volume(f_b[0], f_b[1:], t8, 0, D, variant))
        reg = regnet_us0(torch.stack(costs), weights.regnet)
        del costs
        depth = torch.empty((B, Hh, Ww, 1), device=reg.device, dtype=torch.float32)
        prob = torch.empty_like(depth)
        for b in range(B):
            d_b, p_b = softargmin_prob(reg[b], starts[b], intervals[b], inverse_depth)
            depth[b, :, :, 0], prob[b, :, :, 0] = d_b, p_b
        return depth, prob
    start, interval = _scalar(depth_start), _scalar(depth_interval)
    feats, cams_ = _features(images, cams, weights, features, view_num)
    _, H, W, Cc = feats.shape
    end = np.float32(start) + (np.float32(D) - np.float32(1)) * np.float32(interval)   # model.py:378-379
    plan = _plan(view_num, D, H, W, Cc, weights, "3DCNN", feats.device)
    depth, prob = plan.run_depth(feats, cams_, start, interval, float(end), inverse_depth, variant)
    return depth.reshape(1, H, W, 1).clone(), prob.reshape(1, H, W, 1).clone()


def inference(images, cams, depth_num, depth_start, depth_interval, network_mode="normal",
              is_master_gpu=True, trainable=True, inverse_depth=False, *, weights, view_num=None,
              features=None):
    """mvsnet/model.py:257-372 (the training-graph twin): identical to inference_mem except for
    the rounding of the variance, cost = Q/N - (S/N)^2 (:330-332).  Forward only."""
    return inference_mem(images, cams, depth_num, depth_start, depth_interval, network_mode,
                         is_master_gpu, True, trainable, inverse_depth, weights=weights,
                         view_num=view_num, features=features, variant="eager")


def wta_depth_values(depth_num, depth_start, depth_end, inverse_depth=False):
    """Depth of plane d inside the winner-take-all loop (model.py:605-607,706-715), float32."""
    D = int(depth_num)
    f = np.float32
    d_idx = np.arange(D, dtype=np.float32)
    start, end = f(depth_start), f(depth_end)
    if inverse_depth:
        inv_s, inv_e = f(1) / start, f(1) / end
        inv_interval = (inv_s - inv_e) / (f(D) - f(1))
        return (f(1) / (inv_s - d_idx * inv_interval)).astype(np.float32)
    interval = (end - start) / (f(D) - f(1))
    return (start + d_idx * interval).astype(np.float32)


def inference_winner_take_all(images, cams, depth_num, depth_start, depth_end, network_mode="normal",
                              is_master_gpu=True, reg_type="GRU", inverse_depth=False,
                              training=True, trainable=True, *, weights: MVSNetWeights,
                              view_num=None, features=None):
    """mvsnet/model.py:601-751: ConvGRU-regularised sweep with winner-take-all depth.
    -> (depth_map (1,H,W,1), prob_map (1,H,W,1))."""
    if reg_type != "GRU":
        raise NotImplementedError(reg_type)
    if weights.gru is None:
        raise _lib.MvsnetHipError("weights.gru is required for the GRU regulariser")
    if view_num is None:
        view_num = (features if features is not None else images).shape[-4]
    D = int(depth_num)
    B = _batch_of(images, features, cams)
    if B > 1:        # every op of the recurrent path is per sample (layer_norm normalises over (H,W,C) of ONE sample, convgru.py:30-31)
        starts, ends = _per_sample(depth_start, B), _per_sample(depth_end, B)
        outs = [inference_winner_take_all(_sample(images, b), cams[b:b + 1], D, starts[b], ends[b], network_mode,
                                          is_master_gpu, reg_type, inverse_depth, training, trainable, weights=weights,
                                          view_num=view_num, features=_sample(features, b)) for b in range(B)]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    start, end = _scalar(depth_start), _scalar(depth_end)
    feats, cams_ = _features(images, cams, weights, features, view_num)
    _, H, W, Cc = feats.shape
    interval = float((np.float32(end) - np.float32(start)) / (np.float32(D) - np.float32(1)))  # :606-607
    plan = _plan(view_num, D, H, W, Cc, weights, "GRU", feats.device)
    plan.set_cameras(cams_, start, interval, end, inverse_depth)
    depth, prob = plan.run_gru(feats, wta_depth_values(D, start, end, inverse_depth))
    return depth.reshape(1, H, W, 1).clone(), prob.reshape(1, H, W, 1).clone()
