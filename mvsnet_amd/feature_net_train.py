"""UNetDS2GN towers for TRAINING as one autograd node (SURVEY 8f f2 + f4; mvsnet/cnn_wrapper/mvsnetworks.py:53-115,
the towers of `inference`, mvsnet/model.py:270-292, differentiated by TensorFlow in the reference).

Forward: exactly the inference extractor (`feature_net_hip.HipUNetDS2GN`): one `mvs_conv2d_gn_f32` /
`mvs_deconv2d_gn_f32` launch per layer with the producer's GroupNorm (+ReLU) folded into the consumer's load; the raw
layer outputs it keeps anyway are what the backward needs.  (~0.7 ms for 3 x 480 x 640 against ~3.5 ms for the
MIOpen convolutions + separate GroupNorm passes.)

Backward, per layer in reverse: GroupNorm(+ReLU) backward on the HIP library (`mvs_gn_*_f32`); input gradients of the
3x3 layers on the forward HIP convolution kernels (flipped kernel / conv <-> transposed conv with the same array);
weight gradients (and the two 5x5 stride-2 layers) through ATen's `convolution_backward` (MIOpen) on the materialised
normalised inputs -- the remaining PyTorch-ROCm glue in the towers.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import _lib
from .feature_net import UNET_LAYERS, _same_pad

GN_EPS = 1e-5
_conv_bwd = torch.ops.aten.convolution_backward      # MIOpen's weight gradients (and the two 5x5 layers' input gradients)
_SHAPE_ONLY = {}          # (shape, device) -> uninitialised tensor handed to ATen where only the weight's shape matters


def _cl(t):
    """(V,H,W,C) contiguous -> the (V,C,H,W) channels_last view ATen wants (no copy)."""
    return t.permute(0, 3, 1, 2)


def flatten_unet_params(params) -> List[torch.Tensor]:
    flat = []
    for name, kind, *_ in UNET_LAYERS:
        flat.append(params[name]["w"])
        if kind != "c":
            flat += [params[name]["gamma"], params[name]["beta"]]
    return flat


_WEIGHT_PLANS = {}        # (device, ((layer, address, shape), ...)) -> _WeightPlan
_STATS_TABLES = {}        # (views, slots, layer offsets) -> the job table of mvs_gn_slots_to_channel_sums_many_f64


class _WeightPlan:
    """Every layer's kernel in the layouts the convolution kernels read -- for the forward pass AND for the input-gradient
    convolution of the backward pass -- laid out by ONE launch per step (`mvs_unet_prepare_many_f32`; ~60 launches before).
    Built once per set of variables (their addresses: the trainer's leaves are views of one flat buffer) and kept: the
    prepared copies live in one slab that every step overwrites in stream order, the job table is a handful of ctypes
    arrays.  (A forward pass whose backward has not run yet shares the slab with any later forward pass over the same
    variables: same values unless the variables were updated in between.)"""

    def __init__(self, weights, dev):
        import ctypes as C
        lib = _lib.load()
        chans, jobs = {"data": 4}, []                       # jobs: (key, kind, w, ks, c1, c2, cin_src, cout, floats)
        for name, kind, srcs, k, _mult, stride in UNET_LAYERS:
            w = weights[name]
            cins = [chans[s_] for s_ in srcs]
            cin_tot = sum(cins)
            if kind == "dg":
                cout = w.shape[2]
                n = lib.mvs_deconv2d_prepared_floats(cins[0], cout)
                if n:
                    jobs.append((("fwd", name), 2, w, 3, cins[0], 0, cins[0], cout, n))
            else:
                cout = w.shape[3]
                c1, c2 = cins[0], (cins[1] if len(cins) > 1 else 0)
                jobs.append((("fwd", name), 0, w, k, c1, c2, w.shape[2], cout, lib.mvs_conv2d_prepared_floats(k, c1, c2, cout)))
            chans[name] = cout
            # the input gradient on the forward kernels (HipTowers.backward): 3 x 3 layers that are not fed by the image
            if srcs != ("data",) and k == 3 and cin_tot % 8 == 0:
                if kind == "dg":                            # conv stride 2 with the same array (3,3,Cout,Cin)
                    jobs.append((("bwd", name), 0, w, 3, cout, 0, cout, cin_tot, lib.mvs_conv2d_prepared_floats(3, cout, 0, cin_tot)))
                elif stride == 1:                           # conv with the mirrored, transposed kernel
                    jobs.append((("bwd", name), 1, w, 3, cin_tot, 0, cin_tot, cout, lib.mvs_conv2d_prepared_floats(3, cout, 0, cin_tot)))
                else:                                       # transposed conv with the same array (3,3,Cin,Cout)
                    n = lib.mvs_deconv2d_prepared_floats(cout, cin_tot)
                    if n:
                        jobs.append((("bwd", name), 2, w, 3, cout, 0, cout, cin_tot, n))
        total = sum((j[8] + 63) // 64 * 64 for j in jobs)
        self.slab = torch.empty(total, dtype=torch.float32, device=dev)
        self.prepared, off = {}, 0
        for j in jobs:
            self.prepared[j[0]] = self.slab[off:off + j[8]]
            off += (j[8] + 63) // 64 * 64
        n = self.n = len(jobs)
        ints = lambda col: (C.c_int * n)(*[j[col] for j in jobs])
        self.args = (ints(1), (C.c_void_p * n)(*[j[2].data_ptr() for j in jobs]), ints(3), ints(4), ints(5), ints(6), ints(7),
                     (C.c_void_p * n)(*[self.prepared[j[0]].data_ptr() for j in jobs]))
        self.keep = [j[2] for j in jobs]                    # the addresses in the table stay valid

    def run(self, st):
        _lib.check(_lib.load().mvs_unet_prepare_many_f32(self.n, *self.args, st), "mvs_unet_prepare_many_f32")


def _weight_plan(weights, dev):
    key = (str(dev), tuple((n_, w.data_ptr(), tuple(w.shape)) for n_, w in weights.items()))
    plan = _WEIGHT_PLANS.get(key)
    if plan is None:
        if len(_WEIGHT_PLANS) > 8:                          # trainers come and go in tests: do not collect their slabs
            _WEIGHT_PLANS.clear()
        plan = _WEIGHT_PLANS[key] = _WeightPlan(weights, dev)
    return plan


class HipTowers(torch.autograd.Function):
    """images (V,H,W,3) float32 (centred) or uint8 (as decoded; standardised here) + the tower variables in TensorFlow layouts
    -> features (V,H/4,W/4,32)."""

    @staticmethod
    def forward(ctx, images, into, *flat):
        """`into`: None, or one tensor per entry of `flat` (the variables' slices of a flat gradient buffer): the backward then
        ACCUMULATES the parameter gradients there itself -- two launches for all 94 of them -- and hands autograd nothing
        (otherwise: a permute-copy per kernel and an accumulation launch per variable, ~190 launches)."""
        lib = _lib.load()
        ctx.into = into
        dev = images.device
        V, H, W, _ = images.shape
        if H % 16 or W % 16:
            raise ValueError("UNetDS2GN needs image sizes divisible by 16")
        slots = lib.mvs_gn_stat_slots()
        st = _lib.stream_ptr()
        if images.dtype == torch.uint8:                                          # decoded images: standardised into the padded layout
            data = torch.empty((V, H, W, 4), dtype=torch.float32, device=dev)
            ws = torch.empty(lib.mvs_center_images_workspace_bytes(V) // 8, dtype=torch.int64, device=dev)
            _lib.check(lib.mvs_center_images_u8_f32(_lib.ptr(images.contiguous()), V, H, W, _lib.ptr(data), _lib.ptr(ws), st),
                       "mvs_center_images_u8_f32")
        else:
            data = torch.zeros((V, H, W, 4), dtype=torch.float32, device=dev)   # image padded 3 -> 4 channels
            data[..., :3] = images.detach()
        P, i = {}, 0
        for name, kind, *_ in UNET_LAYERS:
            P[name] = {"w": flat[i].detach().contiguous()}; i += 1
            if kind != "c":
                P[name]["gamma"], P[name]["beta"] = flat[i].detach().contiguous(), flat[i + 1].detach().contiguous(); i += 2
        plan = _weight_plan({n_: P[n_]["w"] for n_ in P}, dev)
        plan.run(st)                                                             # all forward + input-gradient layouts: one launch
        acts: Dict[str, torch.Tensor] = {}
        # every layer's GroupNorm sums in ONE zeroed slab (round 6: one fill instead of 31)
        so_off, so_total = {}, 0
        for name, kind, _s, _k, mult, _st in UNET_LAYERS:
            if kind != "c":
                cout_ = P[name]["w"].shape[2] if kind == "dg" else P[name]["w"].shape[3]
                so_off[name] = (so_total, V * (cout_ // 8) * 2 * slots)
                so_total += so_off[name][1]
        so_slab = torch.zeros(so_total, dtype=torch.float64, device=dev)
        src_of = {"data": (data, None, None, None, 0)}                           # tensor, stats, gamma, beta, relu
        chans = {"data": 4}
        shapes = {"data": (H, W)}
        for name, kind, srcs, k, _mult, stride in UNET_LAYERS:
            w = P[name]["w"]
            h, wd_ = shapes[srcs[0]]
            cins = [chans[s] for s in srcs]
            if kind == "dg":
                cout = w.shape[2]
                ho, wo = 2 * h, 2 * wd_
            else:
                cout = w.shape[3]
                ho, wo = -(-h // stride), -(-wd_ // stride)
            y = torch.empty((V, ho, wo, cout), dtype=torch.float32, device=dev)
            so = so_slab[so_off[name][0]:so_off[name][0] + so_off[name][1]] if kind != "c" else None
            a = src_of[srcs[0]]
            prep = plan.prepared.get(("fwd", name))
            if kind == "dg":
                _lib.check(lib.mvs_deconv2d_gn_f32(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]), cins[0], a[4],
                                                   _lib.ptr(w), _lib.ptr(prep), V, h, wd_, cout, _lib.ptr(y),
                                                   _lib.ptr(so), st), "mvs_deconv2d_gn_f32")
            else:
                c1, c2 = cins[0], (cins[1] if len(cins) > 1 else 0)
                b = src_of[srcs[1]] if len(srcs) > 1 else (None, None, None, None, 0)
                _lib.check(lib.mvs_conv2d_gn_f32(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]), c1, a[4],
                                                 _lib.ptr(b[0]), _lib.ptr(b[1]), _lib.ptr(b[2]), _lib.ptr(b[3]), c2, b[4],
                                                 _lib.ptr(prep), V, h, wd_, cout, k, stride, _lib.ptr(y), _lib.ptr(so), st),
                           "mvs_conv2d_gn_f32")
            acts[name] = y
            chans[name], shapes[name] = cout, (ho, wo)
            src_of[name] = (y, so, P[name].get("gamma"), P[name].get("beta"), 1 if kind == "cg" else 0)
        # per-channel (V, 2, C) float64 sums of every raw output, from the slot sums the convolutions wrote (no second pass over
        # the activations): all layers in one launch, for the backward's GroupNorm kernels
        key = (V, slots, tuple(so_off[n_] for n_ in so_off))
        tab = _STATS_TABLES.get(key)
        if tab is None:
            import ctypes as C
            names = list(so_off)
            couts = [chans[n_] for n_ in names]
            cs_off, tot_ = [], 0
            for c_ in couts:
                cs_off.append(tot_); tot_ += V * 2 * c_
            n = len(names)
            tab = _STATS_TABLES[key] = (n, (C.c_longlong * n)(*[so_off[n_][0] for n_ in names]), (C.c_int * n)(*couts),
                                        (C.c_longlong * n)(*cs_off), tot_, dict(zip(names, zip(cs_off, couts))))
        n, slot_off_a, c_a, stat_off_a, cs_total, cs_where = tab
        cs_slab = torch.empty(cs_total, dtype=torch.float64, device=dev)
        _lib.check(lib.mvs_gn_slots_to_channel_sums_many_f64(n, _lib.ptr(so_slab), slot_off_a, c_a, V, slots, _lib.ptr(cs_slab), stat_off_a, st),
                   "mvs_gn_slots_to_channel_sums_many_f64")
        chan_stats = {n_: cs_slab[o_:o_ + V * 2 * c_].view(V, 2, c_) for n_, (o_, c_) in cs_where.items()}
        ctx.saved = (data, acts, P, chan_stats, plan)
        return acts["conv10_2"].clone()

    @staticmethod
    def backward(ctx, g_feat):
        lib = _lib.load()
        data, acts, P, chan_stats, plan = ctx.saved
        dev = data.device
        st = _lib.stream_ptr()
        kinds = {name: kind for name, kind, *_ in UNET_LAYERS}

        stats_of = chan_stats.__getitem__                                        # per-channel (V,2,C) float64 sums of raw y (forward)

        norm_cache: Dict[str, torch.Tensor] = {}

        def normalised(name):
            """what the consumers of `name` saw: GroupNorm(+ReLU) of the raw output (the padded image for 'data')"""
            if name == "data":
                return data
            if name not in norm_cache:
                y = acts[name]
                V, h, w, c = y.shape
                out = torch.empty_like(y)
                _lib.check(lib.mvs_gn_apply_f32(_lib.ptr(y), _lib.ptr(stats_of(name)), _lib.ptr(P[name]["gamma"]),
                                                _lib.ptr(P[name]["beta"]), GN_EPS, 1 if kinds[name] == "cg" else 0, V, h * w, c,
                                                _lib.ptr(out), st), "mvs_gn_apply_f32")
                norm_cache[name] = out
            return norm_cache[name]

        g_act: Dict[str, torch.Tensor] = {}                                     # gradient w.r.t. the normalised output

        def add_grad(name, g):
            if name == "data":
                return
            g_act[name] = g if name not in g_act else g_act[name] + g

        grads: Dict[str, Dict[str, torch.Tensor]] = {}
        # the (V, 2, C) gradient sums of every GroupNorm layer in one zeroed slab
        bs_off, bs_total = {}, 0
        for name, kind, *_r in UNET_LAYERS:
            if kind != "c":
                V_, _h, _w, c_ = acts[name].shape
                bs_off[name] = (bs_total, V_ * 2 * c_)
                bs_total += lib.mvs_gn_bwd_sums_doubles(V_, c_)             # (slots, V, 2, C): added up by the apply pass
        # ... followed by their totals over the views, (2, C) per layer: [d beta, d gamma]
        ps_off, ps_total = {}, 0
        for name, kind, *_r in UNET_LAYERS:
            if kind != "c":
                ps_off[name] = (ps_total, acts[name].shape[3])
                ps_total += 2 * acts[name].shape[3]
        zero_slab = torch.zeros(bs_total + ps_total, dtype=torch.float64, device=dev)
        bs_slab, ps_slab = zero_slab[:bs_total], zero_slab[bs_total:]
        for name, kind, srcs, k, _mult, stride in reversed(UNET_LAYERS):
            y = acts[name]
            V, ho, wo, cout = y.shape
            if kind == "c":
                g_y = g_feat.contiguous()
                grads[name] = {}
            else:
                g_a = g_act.pop(name).contiguous()
                sums = bs_slab[bs_off[name][0]:]                           # this layer's (slots, V, 2, C) start here
                relu = 1 if kind == "cg" else 0
                args = (_lib.ptr(y), _lib.ptr(stats_of(name)), _lib.ptr(P[name]["gamma"]), _lib.ptr(P[name]["beta"]), GN_EPS, relu,
                        _lib.ptr(g_a))
                tot = ps_slab[ps_off[name][0]:ps_off[name][0] + 2 * cout]
                _lib.check(lib.mvs_gn_bwd_reduce_f32(*args, V, ho * wo, cout, _lib.ptr(sums), st), "mvs_gn_bwd_reduce_f32")
                g_y = torch.empty_like(y)
                _lib.check(lib.mvs_gn_bwd_apply_tot_f32(*args, _lib.ptr(sums), _lib.ptr(tot), V, ho * wo, cout, _lib.ptr(g_y), st),
                           "mvs_gn_bwd_apply_tot_f32")
                grads[name] = {}                           # gamma / beta: views of the converted totals, after the loop
            # convolution backward on the materialised normalised inputs.  Weight gradient: ATen / MIOpen.  Input
            # gradient: the forward HIP kernels -- a stride-1 convolution's is the convolution with the flipped,
            # transposed kernel, a stride-2 convolution's IS the transposed convolution with the same kernel array
            # and vice versa (as for the 3D layers, backward.py); the 5x5 stride-2 layers stay on ATen.
            xs = [normalised(s) for s in srcs]
            x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=3)
            w_tf = P[name]["w"]
            xin, gy = _cl(x), _cl(g_y)
            cin_tot = x.shape[3]                                       # 4 for the image layers (their kernels have 3 input channels)
            need_gx = srcs != ("data",)
            g_x = None
            hip_gx = need_gx and k == 3 and cin_tot % 8 == 0
            if need_gx and not hip_gx:
                w_t = w_tf.permute(3, 2, 0, 1).contiguous()        # conv (Cout,Cin,k,k); transposed conv (Cin,Cout,k,k)
            else:                                                  # ATen computes the weight gradient only: it needs the kernel's SHAPE, not its values
                shp = (w_tf.shape[3], cin_tot if kind != "dg" else w_tf.shape[2], w_tf.shape[0], w_tf.shape[1])
                w_t = _SHAPE_ONLY.get((shp, dev))
                if w_t is None:
                    w_t = _SHAPE_ONLY[(shp, dev)] = torch.empty(shp, dtype=torch.float32, device=dev)
            if hip_gx:
                gxt = torch.empty((V, x.shape[1], x.shape[2], cin_tot), dtype=torch.float32, device=dev)
                prep = plan.prepared.get(("bwd", name))
                if kind == "dg" or stride == 1:                        # a convolution over g_y: stride 2 with the same array / stride 1 with the mirrored one
                    _lib.check(lib.mvs_conv2d_gn_f32(_lib.ptr(g_y), None, None, None, cout, 0, None, None, None, None, 0, 0,
                                                     _lib.ptr(prep), V, ho, wo, cin_tot, 3, 2 if kind == "dg" else 1, _lib.ptr(gxt), None, st),
                               "mvs_conv2d_gn_f32")
                else:                                                   # transposed conv with the same array (3,3,Cin,Cout)
                    _lib.check(lib.mvs_deconv2d_gn_f32(_lib.ptr(g_y), None, None, None, cout, 0, _lib.ptr(w_tf), _lib.ptr(prep),
                                                       V, ho, wo, cin_tot, _lib.ptr(gxt), None, st), "mvs_deconv2d_gn_f32")
                g_x = _cl(gxt)
            mask = [need_gx and not hip_gx, True, False]
            if kind == "dg":
                n_h, n_w = x.shape[1], x.shape[2]
                pb_h = _same_pad(n_h * stride, k, stride)[0]
                pb_w = _same_pad(n_w * stride, k, stride)[0]
                full_h, full_w = stride * (n_h - 1) + k, stride * (n_w - 1) + k
                gfull = F.pad(gy, (pb_w, full_w - pb_w - wo, pb_h, full_h - pb_h - ho))
                gx_a, g_w, _ = _conv_bwd(gfull, xin, w_t, None, [stride, stride], [0, 0], [1, 1], True,
                                                                   [0, 0], 1, mask)
            else:
                ph, pw = _same_pad(x.shape[1], k, stride), _same_pad(x.shape[2], k, stride)
                if ph[0] == ph[1] and pw[0] == pw[1]:
                    gx_a, g_w, _ = _conv_bwd(gy, xin, w_t, None, [stride, stride], [ph[0], pw[0]], [1, 1],
                                                                       False, [0, 0], 1, mask)
                else:
                    xp = F.pad(xin, (pw[0], pw[1], ph[0], ph[1]))
                    gx_a, g_w, _ = _conv_bwd(gy, xp, w_t, None, [stride, stride], [0, 0], [1, 1], False,
                                                                       [0, 0], 1, mask)
                    if mask[0]:
                        gx_a = gx_a[:, :, ph[0]:ph[0] + x.shape[1], pw[0]:pw[0] + x.shape[2]]
            if mask[0]:
                g_x = gx_a
            if ctx.into is not None:                                   # transposed into the flat buffer after the loop
                grads[name]["w_aten"] = g_w.contiguous()
            else:
                g_wtf = g_w.permute(2, 3, 1, 0)                       # back to the TensorFlow layout
                if srcs == ("data",):
                    g_wtf = g_wtf[:, :, :3]
                grads[name]["w"] = g_wtf.contiguous()
            if g_x is not None:
                g_x = g_x.permute(0, 2, 3, 1)                         # (V,H,W,Cin) view
                c0 = 0
                for s_name, xs_ in zip(srcs, xs):
                    c = xs_.shape[3]
                    add_grad(s_name, g_x[..., c0:c0 + c])
                    c0 += c
        ctx.saved = None
        if ctx.into is not None:
            import ctypes as C
            slot, i = {}, 0
            for name, kind, *_ in UNET_LAYERS:
                slot[name] = i
                i += 1 if kind == "c" else 3
            src, dst, dims = [], [], []
            for name, _kind, srcs, k, *_ in UNET_LAYERS:
                gw = grads[name]["w_aten"]                            # (B, A, k, k) -> (k, k, A [:3 for the image], B)
                src.append(gw.data_ptr()); dst.append(ctx.into[slot[name]].data_ptr())
                dims += [gw.shape[0], gw.shape[1], k * k, 3 if srcs == ("data",) else gw.shape[1]]
            n = len(src)
            _lib.check(lib.mvs_transpose_add_many_f32(n, (C.c_void_p * n)(*src), (C.c_void_p * n)(*dst), (C.c_int * (4 * n))(*dims), st),
                       "mvs_transpose_add_many_f32")
            src, dst, cnt = [], [], []
            base = ps_slab.data_ptr()
            for name, (o_, c_) in ps_off.items():                     # [d beta (C), d gamma (C)] float64 per layer
                src += [base + 8 * (o_ + c_), base + 8 * o_]
                dst += [ctx.into[slot[name] + 1].data_ptr(), ctx.into[slot[name] + 2].data_ptr()]
                cnt += [c_, c_]
            n = len(src)
            _lib.check(lib.mvs_add_f64_many_f32(n, (C.c_void_p * n)(*src), (C.c_void_p * n)(*dst), (C.c_int * n)(*cnt), st),
                       "mvs_add_f64_many_f32")
            return (None, None) + (None,) * len(ctx.into)
        ps32 = ps_slab.to(torch.float32)
        for name, (o_, c_) in ps_off.items():
            grads[name]["beta"], grads[name]["gamma"] = ps32[o_:o_ + c_], ps32[o_ + c_:o_ + 2 * c_]
        flat = []
        for name, kind, *_ in UNET_LAYERS:
            flat.append(grads[name]["w"])
            if kind != "c":
                flat += [grads[name]["gamma"], grads[name]["beta"]]
        return (None, None) + tuple(flat)


def hip_towers(images, params, accumulate_into_grads=False):
    """images (V,H,W,3) device tensor, params[name] = {'w','gamma','beta'} leaves in TF layouts -> (V,H/4,W/4,32).
    `accumulate_into_grads`: the leaves carry pre-allocated contiguous `.grad` tensors (train.FlatParameters: views of one flat
    buffer) and the backward adds the parameter gradients into them itself (see HipTowers.forward); autograd then sees no
    gradient for them -- for callers that read `.grad` afterwards, not for torch.autograd.grad."""
    flat = flatten_unet_params(params)
    into = None
    if accumulate_into_grads:
        into = [p.grad for p in flat]
        if any(g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.device != images.device or g.shape != p.shape
               for g, p in zip(into, flat)):
            raise ValueError("accumulate_into_grads needs a contiguous float32 .grad of the variable's shape on every leaf")
    return HipTowers.apply(images, into, *flat)
