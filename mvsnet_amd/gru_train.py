"""Training of the recurrent (ConvGRU) regulariser: `inference_prob_recurrent` (mvsnet/model.py:505-599) and
`mvsnet_classification_loss` (mvsnet/loss.py:223-267), SURVEY 8f row f4.

The reference's branch does not run as shipped (get_loss returns three values where its caller unpacks four,
train.py:355-364 vs :428; `non_zero_mean_absolute_diff` is not defined anywhere in loss.py): what is built here is
what that branch states, with the arity made consistent (loss, less_one, less_three, depth map — as the 3DCNN branch).
One more deliberate difference: the reference unrolls the sweep in Python (model.py:563) and calls
`tf.contrib.layers.layer_norm` without a scope, so TensorFlow uniquifies the default name per call (LayerNorm,
LayerNorm_1, … LayerNorm_{2D-1}): its training graph gives every plane its own gamma / beta, of which the inference
graph (a while_loop body traced once: LayerNorm, LayerNorm_1 | LayerNorm) can only ever load plane 0's.
Here the LayerNorm parameters are shared by all planes — the variables the inference sweep reads.

Where the work runs
  * warp + variance of ALL planes and its backward w.r.t. the feature maps: libmvsnet_hip.so
    (`mvs_cost_volume_f32` / `mvs_cost_volume_bwd_gather_f32`), as on the 3D-CNN training path;
  * the three ConvGRU cells: back-propagation through time over `depth_num` planes, restructured the way the
    inference kernels are: a cell's convolution over concat([x, h]) is split into an x part and an h
    part, and the network is walked cell by cell instead of plane by plane.  The x part of every plane is then ONE
    batched convolution (planes as the batch dimension), and so are the x-part input gradients and ALL weight / bias
    gradients in the backward pass (ATen / MIOpen, as for the 2D towers).  What is sequential in the plane index —
    the h-part convolutions, LayerNorms, gates, and their backward — runs in libmvsnet_hip.so
    (`mvs_gru_train_cell_fwd_f32` / `_bwd_f32`, csrc/gru_train.hip): two launches per plane and direction.
    `conv_gru_sweep` is the same sweep written with torch ops only (used for filter counts the kernels are not
    built for, and as the mid-level checker in the tests).  The inference-only HIP sweep (`mvs_gru_wta_f32`) keeps
    no activations and is not used here.
"""
from __future__ import annotations

import math
import os
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib
from .loss import less_one_percentage, less_three_percentage, original_loss

LN_EPSILON = 1e-12          # tf.contrib.layers.layer_norm's variance_epsilon (convgru.py:30-31)


class VarianceCostVolume(torch.autograd.Function):
    """features (N,H,W,C) [view 0 = reference], transforms (N-1,D,8) -> cost (D,H,W,C) = E[f^2] - E[f]^2 over the
    views (model.py:566-581), both directions on the HIP library."""

    @staticmethod
    def forward(ctx, features, transforms):
        from .model import cost_volume
        features = features.detach().contiguous()
        ctx.saved = (features, transforms)
        return cost_volume(features[0], features[1:], transforms, variant="eager")

    @staticmethod
    def backward(ctx, g_cost):
        from .backward import cost_volume_bwd
        features, transforms = ctx.saved
        g_ref, g_src = cost_volume_bwd(features[0], features[1:], transforms, g_cost.contiguous())
        ctx.saved = None
        return torch.cat([g_ref[None], g_src], 0), None


def _layer_norm(x, gamma, beta):
    """tf.contrib.layers.layer_norm on one sample: moments over (H,W,C), per-channel gamma / beta."""
    var, mean = torch.var_mean(x, unbiased=False)
    return (x - mean) * torch.rsqrt(var + LN_EPSILON) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)


def conv_gru_sweep(x_all, p):
    """One ConvGRUCell (convgru.py:82-122) over all planes: x_all (D,Cin,H,W) -> states (D,F,H,W), zero initial state
    (model.py:546-551)."""
    Fn = p["out_b"].shape[0]
    D, Cin = x_all.shape[0], x_all.shape[1]
    wg = p["gates_w"].permute(3, 2, 0, 1)                       # TF (3,3,Cin+F,2F) -> (2F,Cin+F,3,3)
    wo = p["out_w"].permute(3, 2, 0, 1)
    gx = F.conv2d(x_all, wg[:, :Cin], p["gates_b"], padding=1)  # x parts of every plane, batched
    ox = F.conv2d(x_all, wo[:, :Cin], p["out_b"], padding=1)
    wgh, woh = wg[:, Cin:].contiguous(), wo[:, Cin:].contiguous()
    h = x_all.new_zeros((1, Fn) + tuple(x_all.shape[2:]))
    states = []
    for d in range(D):
        g = gx[d:d + 1] + F.conv2d(h, wgh, padding=1)           # :89-94
        r = torch.sigmoid(_layer_norm(g[:, :Fn], p["reset_gamma"], p["reset_beta"]))       # :97,101
        u = torch.sigmoid(_layer_norm(g[:, Fn:], p["update_gamma"], p["update_beta"]))     # :98,102
        c = ox[d:d + 1] + F.conv2d(r * h, woh, padding=1)       # :107-111
        y = torch.tanh(_layer_norm(c, p["out_gamma"], p["out_beta"]))                      # :114-117
        h = u * h + (1.0 - u) * y                               # :120
        states.append(h)
    return torch.cat(states, 0)


HIP_FILTERS = (16, 8, 4, 2, 1)     # instances of csrc/gru_train.hip


CELL_FIELDS = ("gates_w", "gates_b", "out_w", "out_b", "reset_gamma", "reset_beta", "update_gamma", "update_beta",
               "out_gamma", "out_beta")


def _cell_forward(x, gates_w, gates_b, out_w, out_b, rg, rb, ug, ub, og, ob):
    """x (D,H,W,Cin) -> (states (D,H,W,F) [a view of the kept (D+1,H,W,F) buffer], what the backward needs)."""
    lib = _lib.load()
    P, dev = _lib.ptr, x.device
    D, H, W, Cin = x.shape
    Fn = int(out_b.shape[0])
    x = x.detach().contiguous()
    gates_w, out_w = gates_w.detach(), out_w.detach()
    # x parts of both convolutions for every plane: one batched convolution, channels [reset | update | candidate]
    wx = torch.cat([gates_w[:, :, :Cin, :], out_w[:, :, :Cin, :]], 3).permute(3, 2, 0, 1).contiguous()
    px = F.conv2d(x.permute(0, 3, 1, 2), wx, torch.cat([gates_b.detach(), out_b.detach()]), padding=1)
    px = px.permute(0, 2, 3, 1).contiguous()
    wgh, woh = gates_w[:, :, Cin:, :].contiguous(), out_w[:, :, Cin:, :].contiguous()
    ln = torch.stack([t.detach() for t in (rg, rb, ug, ub, og, ob)]).contiguous()
    sf, sb = _lib.C.c_int(), _lib.C.c_int()
    lib.mvs_gru_train_slots(_lib.C.byref(sf), _lib.C.byref(sb))
    new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    g, c, rh, h = new(D, H, W, 2 * Fn), new(D, H, W, Fn), new(D, H, W, Fn), new(D + 1, H, W, Fn)
    h[0].zero_()                                                          # model.py:546-551
    stats = torch.zeros((D, sf.value, 6), device=dev, dtype=torch.float64)
    _lib.check(lib.mvs_gru_train_cell_fwd_f32(P(px), P(wgh), P(woh), P(ln), D, H, W, Fn, P(g), P(c), P(rh), P(h),
                                              P(stats), _lib.stream_ptr()), "mvs_gru_train_cell_fwd_f32")
    return h[1:], (x, wx, wgh, woh, ln, g, c, rh, h, stats, sb.value)


WGRAD_CHANNELS = ((16, 32), (16, 32, 48))     # (Cin, Cout) instances of csrc/conv2d_wgrad.hip


def conv2d_wgrad(x, g, g_off, cout):
    """dW (3,3,Cin,cout) of a 3x3 SAME convolution over the batch of planes: x (N,H,W,Cin), g (N,H,W,Cg) of which the
    channels [g_off, g_off + cout) are the output gradient.  MFMA kernel where it is built, ATen otherwise."""
    N, H, W, Cin = x.shape
    if Cin in WGRAD_CHANNELS[0] and cout in WGRAD_CHANNELS[1] and g.shape[-1] % 4 == 0 and g_off % 4 == 0:
        lib = _lib.load()
        need = lib.mvs_conv2d_wgrad_workspace_bytes(N, H, W, Cin, cout)
        ws = torch.empty(need, device=x.device, dtype=torch.uint8)
        dw = torch.empty((3, 3, Cin, cout), device=x.device, dtype=torch.float32)
        _lib.check(lib.mvs_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(g), g.shape[-1], g_off, N, H, W, Cin, cout,
                                            _lib.C.c_void_p(ws.data_ptr()), need, _lib.ptr(dw), _lib.stream_ptr()),
                   "mvs_conv2d_wgrad_f32")
        return dw
    gp = g.permute(0, 3, 1, 2)[:, g_off:g_off + cout]
    w = x.new_empty((cout, Cin, 3, 3))
    g_w = torch.ops.aten.convolution_backward(gp, x.permute(0, 3, 1, 2), w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    return g_w.permute(2, 3, 1, 0)


def _weight_grads(x, h_prev, rh, gpx, part, Fn):
    """The ten parameter gradients of a cell (CELL_FIELDS order) from the kept px-gradients: batched over all planes."""
    g_wx = conv2d_wgrad(x, gpx, 0, 3 * Fn)
    g_wgh = conv2d_wgrad(h_prev, gpx, 0, 2 * Fn)
    g_woh = conv2d_wgrad(rh, gpx, 2 * Fn, Fn)
    g_b = gpx.sum((0, 1, 2))
    sums = part.sum((0, 2)).to(torch.float32)                             # (3 LayerNorms, [d beta, d gamma], F)
    return (torch.cat([g_wx[..., :2 * Fn], g_wgh], 2), g_b[:2 * Fn], torch.cat([g_wx[..., 2 * Fn:], g_woh], 2),
            g_b[2 * Fn:], sums[0, 1], sums[0, 0], sums[1, 1], sums[1, 0], sums[2, 1], sums[2, 0])


def _cell_backward(saved, gh, need_x_grad=True):
    """gh (D,H,W,F): gradient reaching every state from outside the recurrence.  Returns (g_x or None, the ten
    parameter gradients in CELL_FIELDS order).  (Putting the batched weight gradients on a second stream, under the
    launch-latency-bound sweep of the next cell down, was measured: no gain, the convolution's workgroups fill the
    CUs and the sweep's launches queue behind them.)"""
    lib = _lib.load()
    P = _lib.ptr
    x, wx, wgh, woh, ln, g, c, rh, h, stats, slots = saved
    D, H, W, Cin = x.shape
    Fn = c.shape[-1]
    dev = x.device
    gh = gh.contiguous()
    flip_t = lambda w: w.flip(0, 1).permute(0, 1, 3, 2).contiguous()      # the kernel of the input gradient
    wgh_t, woh_t = flip_t(wgh), flip_t(woh)
    gpx = torch.empty((D, H, W, 3 * Fn), device=dev, dtype=torch.float32)
    part = torch.zeros((D, 3, slots, 2, Fn), device=dev, dtype=torch.float64)
    scratch = torch.zeros((6, H, W, Fn), device=dev, dtype=torch.float32)
    _lib.check(lib.mvs_gru_train_cell_bwd_f32(P(gh), P(g), P(c), P(h), P(stats), P(wgh_t), P(woh_t), P(ln), D, H, W, Fn,
                                              P(gpx), P(part), P(scratch), None, None, _lib.stream_ptr()), "mvs_gru_train_cell_bwd_f32")
    # everything that is not sequential: batched convolutions over the planes
    g_x = None
    if need_x_grad:
        nchw = lambda t: t.permute(0, 3, 1, 2)
        g_x = torch.ops.aten.convolution_backward(nchw(gpx), nchw(x), wx, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                  [True, False, False])[0].permute(0, 2, 3, 1)
    return g_x, _weight_grads(x, h[:D], rh, gpx, part, Fn)


class ConvGRUSweep(torch.autograd.Function):
    """One ConvGRUCell over all planes on the HIP library: x (D,H,W,Cin) -> states (D,H,W,F), zero initial state."""

    @staticmethod
    def forward(ctx, x, *params):
        out, ctx.saved = _cell_forward(x, *params)
        return out

    @staticmethod
    def backward(ctx, gh):
        saved, ctx.saved = ctx.saved, None
        g_x, grads = _cell_backward(saved, gh, ctx.needs_input_grad[0])
        return (g_x,) + grads


def conv_gru_sweep_hip(x_nhwc, p):
    return ConvGRUSweep.apply(x_nhwc, *[p[k] for k in CELL_FIELDS])


# ---- the three cells as a wavefront on three streams ------------------------------------------------------------
CHUNK = int(os.environ.get("MVS_GRU_TRAIN_CHUNK", "32"))     # planes per hand-over between the cells
_STREAMS = {}


def _cell_streams(dev):
    st = _STREAMS.get(dev)
    if st is None:
        st = _STREAMS[dev] = [torch.cuda.Stream(dev) for _ in range(3)]
    return st


class RecurrentCells(torch.autograd.Function):
    """conv_gru1 -> conv_gru2 -> conv_gru3 over all planes: x (D,H,W,C) -> states of cell 3 (D,H,W,F3).

    Walking the network cell by cell leaves the GPU to one launch-latency-bound chain at a time.  Here every cell has
    its own HIP stream and works through the planes in chunks of CHUNK: as soon as cell k has finished a chunk (an
    event), cell k+1 forms the x parts of that chunk (one batched convolution) and sweeps it while cell k is already
    in the next chunk -- a wavefront over (chunk, cell), forward; backward the same in reverse (cell 3 first, chunks
    descending, the state gradient handed from chunk to chunk through dh_out -> dh_in, the x-gradient of a chunk
    being the next cell down's `gh` of that chunk).  The batched weight / bias gradients follow on the caller's
    stream once the streams have joined.  (Measured and dropped: a cell's batched x-part convolutions on a stream of
    their own, ahead of its sweep -- their workgroups fill the CUs and the sweeps' launches queue behind them:
    38.4 -> 41.2 ms.)"""

    @staticmethod
    def forward(ctx, x, *params):
        lib = _lib.load()
        P, dev = _lib.ptr, x.device
        D, H, W, _ = x.shape
        x = x.detach().contiguous()
        main = torch.cuda.current_stream(dev)
        streams = _cell_streams(dev)
        chunks = [(a, min(a + CHUNK, D)) for a in range(0, D, CHUNK)]
        sf, sb = _lib.C.c_int(), _lib.C.c_int()
        lib.mvs_gru_train_slots(_lib.C.byref(sf), _lib.C.byref(sb))
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        cells, inp = [], x
        for k in range(3):                                  # every long-lived buffer comes from the caller's stream
            gates_w, gates_b, out_w, out_b, rg, rb, ug, ub, og, ob = [t.detach() for t in params[10 * k:10 * k + 10]]
            Cin, Fn = int(inp.shape[-1]), int(out_b.shape[0])
            cl = dict(Cin=Cin, F=Fn, x=inp,
                      wx=torch.cat([gates_w[:, :, :Cin, :], out_w[:, :, :Cin, :]], 3).permute(3, 2, 0, 1).contiguous(),
                      bx=torch.cat([gates_b, out_b]), wgh=gates_w[:, :, Cin:, :].contiguous(), woh=out_w[:, :, Cin:, :].contiguous(),
                      ln=torch.stack([rg, rb, ug, ub, og, ob]).contiguous(),
                      g=new(D, H, W, 2 * Fn), c=new(D, H, W, Fn), rh=new(D, H, W, Fn), h=new(D + 1, H, W, Fn),
                      stats=torch.zeros((D, sf.value, 6), device=dev, dtype=torch.float64))
            cl["h"][0].zero_()                              # model.py:546-551
            cells.append(cl)
            inp = cl["h"][1:]
        start = main.record_event()
        above = None
        for k, cl in enumerate(cells):
            st, done = streams[k], []
            with torch.cuda.stream(st):
                st.wait_event(start)
                for j, (a, b) in enumerate(chunks):
                    if above is not None:
                        st.wait_event(above[j])
                    px = F.conv2d(cl["x"][a:b].permute(0, 3, 1, 2), cl["wx"], cl["bx"], padding=1).permute(0, 2, 3, 1).contiguous()
                    _lib.check(lib.mvs_gru_train_cell_fwd_f32(
                        P(px), P(cl["wgh"]), P(cl["woh"]), P(cl["ln"]), b - a, H, W, cl["F"], P(cl["g"][a:b]), P(cl["c"][a:b]),
                        P(cl["rh"][a:b]), P(cl["h"][a:b + 1]), P(cl["stats"][a:b]), _lib.stream_ptr()), "mvs_gru_train_cell_fwd_f32")
                    done.append(st.record_event())
            above = done
        for st in streams:
            main.wait_stream(st)
        ctx.saved = (cells, chunks, sb.value)
        return cells[2]["h"][1:]

    @staticmethod
    def backward(ctx, gh3):
        lib = _lib.load()
        P, dev = _lib.ptr, gh3.device
        cells, chunks, slots = ctx.saved
        ctx.saved = None
        D, H, W = cells[0]["g"].shape[:3]
        main = torch.cuda.current_stream(dev)
        streams = _cell_streams(dev)
        new = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        flip_t = lambda w: w.flip(0, 1).permute(0, 1, 3, 2).contiguous()      # the kernel of the input gradient
        cb = torch.ops.aten.convolution_backward
        nchw = lambda t: t.permute(0, 3, 1, 2)
        args = ([1, 1], [1, 1], [1, 1], False, [0, 0], 1)
        need_x = ctx.needs_input_grad[0]
        for k, cl in enumerate(cells):
            Fn = cl["F"]
            cl.update(gh=gh3.contiguous() if k == 2 else new(D, H, W, Fn), gpx=new(D, H, W, 3 * Fn),
                      part=torch.zeros((D, 3, slots, 2, Fn), device=dev, dtype=torch.float64), scratch=new(6, H, W, Fn),
                      carry=new(len(chunks) + 1, H, W, Fn), wgh_t=flip_t(cl["wgh"]), woh_t=flip_t(cl["woh"]))
            cl["carry"][len(chunks)].zero_()                 # nothing arrives at the last plane from beyond the sweep
        g_x = new(*cells[0]["x"].shape) if need_x else None
        start = main.record_event()
        above = None
        for k in (2, 1, 0):
            cl, st, done = cells[k], streams[k], [None] * len(chunks)
            below = cells[k - 1]["gh"] if k > 0 else g_x
            with torch.cuda.stream(st):
                st.wait_event(start)
                for j in reversed(range(len(chunks))):
                    a, b = chunks[j]
                    if above is not None:
                        st.wait_event(above[j])
                    _lib.check(lib.mvs_gru_train_cell_bwd_f32(
                        P(cl["gh"][a:b]), P(cl["g"][a:b]), P(cl["c"][a:b]), P(cl["h"][a:b]), P(cl["stats"][a:b]), P(cl["wgh_t"]),
                        P(cl["woh_t"]), P(cl["ln"]), b - a, H, W, cl["F"], P(cl["gpx"][a:b]), P(cl["part"][a:b]), P(cl["scratch"]),
                        P(cl["carry"][j + 1]), P(cl["carry"][j]), _lib.stream_ptr()), "mvs_gru_train_cell_bwd_f32")
                    if below is not None:                    # the x-gradient of this chunk: what the cell below waits for
                        gx = cb(nchw(cl["gpx"][a:b]), nchw(cl["x"][a:b]), cl["wx"], None, *args, [True, False, False])[0]
                        below[a:b].copy_(gx.permute(0, 2, 3, 1))
                    done[j] = st.record_event()
            above = done
        for st in streams:
            main.wait_stream(st)
        # not sequential and on nobody's critical path: weight / bias gradients, batched over all planes
        grads = ()
        for cl in cells:
            grads += _weight_grads(cl["x"], cl["h"][:D], cl["rh"], cl["gpx"], cl["part"], cl["F"])
        return (g_x,) + grads


def recurrent_regularisation(features, transforms, gru):
    """features (N,H,W,C), transforms (N-1,D,8), gru = {'gru1','gru2','gru3': cell tensors, 'prob_w','prob_b'} ->
    regularised cost `reg` (D,H,W) (model.py:563-592, before the softmax)."""
    cin = int(gru["gru1"]["gates_w"].shape[2]) - int(gru["gru1"]["out_b"].shape[0])
    feats = features
    if feats.shape[-1] not in (16, 32):                         # channel counts the HIP gather kernels tile
        feats = F.pad(feats, (0, (16 if feats.shape[-1] < 16 else 32) - feats.shape[-1]))
    cost = VarianceCostVolume.apply(feats, transforms)          # (D,H,W,C)
    s = -cost[..., :cin]                                        # the cells see -cost (model.py:584)
    cells = ("gru1", "gru2", "gru3")
    if all(int(gru[k]["out_b"].shape[0]) in HIP_FILTERS for k in cells):
        s = RecurrentCells.apply(s, *[gru[k][f] for k in cells for f in CELL_FIELDS])
    else:                                                       # filter counts without a kernel instance ('fat' modes)
        for k in cells:
            s = conv_gru_sweep(s.permute(0, 3, 1, 2), gru[k]).permute(0, 2, 3, 1)
    reg = F.conv2d(s.permute(0, 3, 1, 2), gru["prob_w"].permute(3, 2, 0, 1), gru["prob_b"], padding=1)   # :587-588
    return reg[:, 0]


def inference_prob_recurrent(features, transforms, gru):
    """prob_volume (D,H,W) = softmax over the planes of the regularised cost (model.py:591-592).  The feature
    towers are the caller's (as everywhere in this package: model.inference_mem takes features or images)."""
    return torch.softmax(recurrent_regularisation(features, transforms, gru), dim=0)


def mvsnet_classification_loss(prob_volume, gt_depth_image, depth_num, depth_start, depth_interval):
    """loss.py:223-267 for one sample.  prob_volume (D,H,W), gt_depth_image (1,H,W,1).
    Returns (masked_cross_entropy, masked_mae, less_one_accuracy, less_three_accuracy, wta_depth_map (1,H,W,1))."""
    gt = gt_depth_image
    dt, dev = gt.dtype, gt.device
    start = torch.as_tensor(depth_start, dtype=dt, device=dev).reshape(-1, 1, 1, 1)
    interval = torch.as_tensor(depth_interval, dtype=dt, device=dev).reshape(-1, 1, 1, 1)
    mask = (gt != 0).to(dt)
    valid = mask.sum() + 1e-7
    index = torch.round(mask * ((gt - start) / interval)).to(torch.int64)[0, :, :, 0]      # tf.round: half to even
    inside = ((index >= 0) & (index < depth_num)).to(dt)       # tf.one_hot: an index outside [0, D) is an all-zero row
    picked = torch.gather(prob_volume, 0, index.clamp(0, depth_num - 1)[None])[0]
    cross_entropy = -(inside * torch.log(picked))
    loss = (mask[0, :, :, 0] * cross_entropy).sum() / valid
    wta = torch.argmax(prob_volume, dim=0).to(dt)[None, :, :, None] * interval + start
    step = interval.abs().reshape(-1)
    # `non_zero_mean_absolute_diff` is not defined in the reference's loss.py; it is the paper's masked mean absolute
    # error in units of the interval, which loss.py keeps as original_loss (:14-27)
    mae = original_loss(gt, wta, step)
    return loss, mae, less_one_percentage(gt, wta, step), less_three_percentage(gt, wta, step), wta


def glorot_gru_params(template, seed=0):
    """tf.layers.conv2d defaults (glorot_uniform kernels, zero biases) and layer_norm's gamma = 1, beta = 0."""
    rs = np.random.RandomState(seed)

    def kernel(w):
        w = np.asarray(w)
        rf = int(np.prod(w.shape[:-2]))
        limit = math.sqrt(6.0 / (rf * w.shape[-2] + rf * w.shape[-1]))
        return rs.uniform(-limit, limit, w.shape).astype(np.float32)

    out: Dict[str, object] = {}
    for name in ("gru1", "gru2", "gru3"):
        p = template[name]
        q = {"gates_w": kernel(p["gates_w"]), "out_w": kernel(p["out_w"])}
        for k in ("gates_b", "out_b", "reset_beta", "update_beta", "out_beta"):
            q[k] = np.zeros_like(np.asarray(p[k], np.float32))
        for k in ("reset_gamma", "update_gamma", "out_gamma"):
            q[k] = np.ones_like(np.asarray(p[k], np.float32))
        out[name] = q
    out["prob_w"] = kernel(template["prob_w"])
    out["prob_b"] = np.zeros_like(np.asarray(template["prob_b"], np.float32))
    return out
