"""Training of MVSNet (3D-CNN or recurrent regulariser) on MI355X: host-side mirror of mvsnet/train.py.

    python -m mvsnet_amd.train --train_data_root <root with train/ and val/ session folders> --model_dir <out>
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m mvsnet_amd.train ...

What the reference does (train.py:412-523) and how it maps here:
  * one TF graph with `num_gpus` towers, each pulling its own batch; tower gradients are averaged on the host
    graph (`average_gradients`, train.py:155-187) and applied once per step  ->  one process per GPU
    (`torch.distributed`, backend "nccl" = RCCL), every rank draws its own clusters (rank r takes clusters
    r, r+P, ...), ONE all-reduce of the flat gradient buffer per step, then the optimiser update as one HIP
    launch over the flat parameter buffer (all variables are views into it);
  * loss = mvsnet_regression_loss on the 1/4-resolution depth map (train.py:350-353), flags loss_type / alpha /
    beta / eta / grad_loss (train.py:121-134);
  * tf.train.exponential_decay(base_lr, global_step, stepvalue, gamma) (train.py:256-257), RMSProp / momentum /
    Adam with TensorFlow's defaults and update formulas (train.py:258-266);
  * checkpoints `<model_dir>/<regularization>/<network_mode>/model.ckpt-<step>` (train.py:360-365) written in
    TensorFlow's own bundle format with the reference's variable names (tf_checkpoint.py); the name set is the
    reference graph's `tf.global_variables()` (what its `tf.train.Saver` restores, predictlib.py:69-76): the
    trainable variables, optimiser slots `<var>/RMSProp`, `<var>/RMSProp_1`, ... (Adam: also `beta1_power`,
    `beta2_power`), `global_step`, and the `<layer>/bn/moving_mean|moving_variance` pairs that
    tf.layers.batch_normalization creates (network.py:492-509) although `training=True` never reads or updates
    them (written at their initial zeros / ones).  Export is unpinned: no TensorFlow exists offline to restore one.
The forward / backward of the hot path run in libmvsnet_hip.so (backward.py); the 2D towers are
feature_net.unet_forward under torch autograd (north_star keeps them on PyTorch-ROCm).
network_mode: 'normal' and 'semilite' (= 'normal' channel counts under the reference's Python 2 division) natively;
'lite' (the reference's default) / 'ultralite' zero-padded to the
'normal' shapes (padded entries provably stay zero); wider modes raise NotImplementedError.
Training through the refinement network (`--refinement`, train.py:317-349: all / refine_only / main_only; the towers of
refine.py under torch autograd, the probability-map gradient through mvs_softargmin_bwd_f32).
`--regularization GRU` (train.py:355-364): gru_train.py — cost volume forward / backward on the HIP library, the
ConvGRU cells under torch autograd, classification loss; the generator then also yields every cluster with the depth
sweep reversed (flip_cams, train.py:194-203).
"""
from __future__ import annotations

import argparse
import math
import os
import time
from typing import Dict

import numpy as np
import torch

from . import _lib, tf_checkpoint
from .feature_net import trainable_layers, unet_forward
from .homography_warping import homography_transforms
from .loss import mvsnet_regression_loss
from .synthetic import base_filter, make_regnet_params, make_unet_params


def glorot_uniform_like(params, seed=0):
    """tf.layers.conv2d/conv3d default kernel initialiser (glorot_uniform: limit = sqrt(6/(fan_in+fan_out)),
    fan = receptive field x channels); gamma = 1, beta = 0 (network.py:257-267)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, p in params.items():
        w = np.asarray(p["w"])
        rf = int(np.prod(w.shape[:-2]))
        limit = math.sqrt(6.0 / (rf * w.shape[-2] + rf * w.shape[-1]))
        q = {"w": rs.uniform(-limit, limit, w.shape).astype(np.float32)}
        if "gamma" in p:
            q["gamma"] = np.ones_like(np.asarray(p["gamma"], np.float32))
            q["beta"] = np.zeros_like(np.asarray(p["beta"], np.float32))
        out[name] = q
    return out


def exponential_decay(base_lr, global_step, decay_steps, decay_rate):
    """tf.train.exponential_decay, staircase=False (train.py:256-257)."""
    return base_lr * decay_rate ** (float(global_step) / float(decay_steps))


OPTIMIZER_SLOTS = {"rmsprop": ("RMSProp", "RMSProp_1"), "momentum": ("Momentum",), "adam": ("Adam", "Adam_1")}


class FlatParameters:
    """All trainable variables in one device buffer, TensorFlow layouts, addressed by the reference's variable
    names; `grad` is a second flat buffer the leaves' .grad tensors are views of, so one all-reduce and one
    optimiser launch cover the whole model."""

    def __init__(self, unet, regnet, network_mode, device, refine=None, refinement=None, gru=None):
        self.names = tf_checkpoint.variable_names(network_mode, "GRU" if gru is not None else "3DCNN",
                                                  refinement if refine is not None else None)
        groups = {"unet": unet, "regnet": regnet, "refine": refine, "gru": gru}
        # prob_conv sits beside the cells in the GRU dictionary: key ("gru", None, "prob_w")
        lookup = lambda g, layer, field: groups[g][field] if layer is None else groups[g][layer][field]
        self.index = []                                   # (key, var_name, offset, shape)
        off = 0
        for key in sorted(self.names, key=lambda k: self.names[k]):
            group, layer, field = key
            a = np.asarray(lookup(group, layer, field), np.float32)
            self.index.append((key, self.names[key], off, a.shape))
            off += a.size
        self.numel = off
        self.data = torch.empty(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        host = np.empty(off, np.float32)
        for key, _v, o, shape in self.index:
            group, layer, field = key
            host[o:o + int(np.prod(shape))] = np.asarray(lookup(group, layer, field), np.float32).ravel()
        self.data.copy_(torch.from_numpy(host))
        self.leaves: Dict[tuple, torch.Tensor] = {}
        for key, _v, o, shape in self.index:
            n = int(np.prod(shape))
            leaf = self.data[o:o + n].view(shape).requires_grad_(True)
            leaf.grad = self.grad[o:o + n].view(shape)
            self.leaves[key] = leaf

    def group(self, group):
        out: Dict[str, Dict[str, torch.Tensor]] = {}
        for (g, layer, field), leaf in self.leaves.items():
            if g == group and layer is None:
                out[field] = leaf
            elif g == group:
                out.setdefault(layer, {})[field] = leaf
        return out

    def named_arrays(self, flat):
        host = flat.detach().cpu().numpy()
        return {v: host[o:o + int(np.prod(shape))].reshape(shape) for _k, v, o, shape in self.index}


class Trainer:
    def __init__(self, network_mode="normal", device="cuda", optimizer="rmsprop", base_lr=1e-3, stepvalue=70000,
                 gamma=0.5, loss_type="power", alpha=0.25, beta=0.0, eta=0.02, grad_loss=True, init=None, seed=0,
                 sync_bn=False, refinement=False, refinement_network="unet", upsample_before_refinement=True,
                 refine_with_confidence=True, refinement_train_mode="all", refine_with_stereo=False,
                 regularization="3DCNN"):
        from . import ensure_miopen_workaround
        ensure_miopen_workaround("mvsnet_amd.train")      # before the first ATen convolution of this process (narrow towers, 2D backward)
        if regularization not in ("3DCNN", "GRU"):
            raise NotImplementedError("regularization %r" % regularization)
        if regularization == "GRU" and refinement:
            raise NotImplementedError("refinement is only applicable with the 3DCNN regulariser (train.py:77-79)")
        self.regularization = regularization
        if optimizer not in OPTIMIZER_SLOTS:
            raise NotImplementedError("Optimizer %s is not implemented" % optimizer)       # train.py:268-271
        self.network_mode, self.device = network_mode, torch.device(device)
        self.optimizer, self.base_lr, self.stepvalue, self.gamma = optimizer, base_lr, stepvalue, gamma
        self.loss_args = dict(loss_type=loss_type, alpha=alpha, beta=beta, eta=eta, grad_loss=grad_loss)
        if init is None:
            init = {"unet": glorot_uniform_like(make_unet_params(network_mode), seed)}
            if regularization == "GRU":
                from .gru_train import glorot_gru_params
                from .synthetic import make_gru_params
                init["gru"] = glorot_gru_params(make_gru_params(network_mode, in_channels=4 * base_filter(network_mode)), seed + 1)
            else:
                init["regnet"] = glorot_uniform_like(make_regnet_params(network_mode), seed + 1)
        self.refinement = False
        self.sync = None
        self.native_shapes = {}
        if regularization == "GRU":                     # no BatchNorm, no refinement, no padding on this branch
            self.params = FlatParameters(init["unet"], None, network_mode, self.device, gru=init["gru"])
            self._finish_init(optimizer)
            return
        # narrower modes ('lite' is the reference's default, train.py:82) train zero-padded to the shapes the
        # MFMA kernels tile (model.pad_regnet_params): padded kernels / gamma / beta receive exactly zero
        # gradients (their inputs or their BN scale are zero), so they stay zero; checkpoints hold the native shapes
        from .model import pad_regnet_params
        regnet = init["regnet"]
        self.native_shapes = {}
        names = tf_checkpoint.variable_names(network_mode, "3DCNN")
        for (group, layer, field), var in names.items():
            if group == "regnet":
                self.native_shapes[var] = tuple(np.asarray(regnet[layer][field]).shape)
        if np.asarray(regnet["3dconv1_0"]["w"]).shape[3] < 32:
            padded = pad_regnet_params(regnet)
            if padded is None:
                raise NotImplementedError("network_mode %r: no weight-gradient kernels for these channel counts" % network_mode)
            regnet = padded
        elif np.asarray(regnet["3dconv1_0"]["w"]).shape[3] > 32:
            raise NotImplementedError("network_mode %r: no weight-gradient kernels for these channel counts" % network_mode)
        # depth refinement (train.py:317-349): the refinement tower's variables join the flat buffer
        self.refinement = bool(refinement)
        self.refine_cfg = (refinement_network, bool(upsample_before_refinement), bool(refine_with_confidence), refinement_train_mode)
        self.refine_with_stereo = bool(refine_with_stereo)
        if refinement_train_mode not in ("all", "refine_only", "main_only"):
            raise ValueError("refinement_train_mode must be all, refine_only or main_only")
        refine = None
        if self.refinement:
            from .refine import make_refine_params
            refine = init.get("refine")
            if refine is None:
                tmpl = make_refine_params(refinement_network, network_mode, in_channels=(5 if refine_with_confidence else 4) + (3 if refine_with_stereo else 0))
                g = glorot_uniform_like(tmpl, seed + 2)
                refine = {k: {"w": g[k]["w"], "b": np.zeros_like(np.asarray(tmpl[k]["b"], np.float32))} for k in tmpl}   # tf.layers: zero biases
        self.params = FlatParameters(init["unet"], regnet, network_mode, self.device, refine, refinement_network)
        self._finish_init(optimizer)
        if sync_bn and self.world > 1:
            from .backward import SyncBN
            self.sync = SyncBN()

    def _finish_init(self, optimizer):
        n = self.params.numel
        ones = optimizer == "rmsprop"                    # TF's RMSProp `rms` slot starts at one
        self.slots = [torch.ones(n, device=self.device) if (ones and i == 0) else torch.zeros(n, device=self.device)
                      for i in range(len(OPTIMIZER_SLOTS[optimizer]))]
        self.global_step = 0
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1

    # -- one optimisation step ------------------------------------------------------------------------
    def loss(self, images, cams, depth_image, depth_num, full_depth=None, sync="self"):
        """images (N,H,W,3), cams (N,2,4,4) at the output scale, depth_image (H/4,W/4,1) GT.  Returns
        (loss, less_one, less_three, depth_map) exactly as get_loss (train.py:307-353), batch 1.
        `images`: float32 = centred as the reference's generator hands them out; uint8 = the cropped images as decoded
        (TrainingPrefetcher(center=False)): standardised per image and channel on the device (utils.py:33-38; by the HIP towers
        themselves, mvs_center_images_u8_f32).
        `sync`: the cross-replica BatchNorm reducer; "self" = the trainer's own (None unless --sync_bn)."""
        sync = self.sync if sync == "self" else sync
        from .backward import plane_sweep_depth
        images = torch.as_tensor(images, device=self.device)
        raw = images if images.dtype == torch.uint8 else None
        hip_towers_path = self.device.type == "cuda" and self.network_mode == "normal"
        if raw is None:
            images = images.to(torch.float32)
        elif not (hip_towers_path and not self.refinement):   # somebody needs the float images: the ATen towers, the refinement's guide
            from .inference import center_images_device
            images = center_images_device(raw)
        cams_t = torch.as_tensor(cams, dtype=torch.float32, device=self.device)
        gt = torch.as_tensor(depth_image, dtype=torch.float32, device=self.device)[None]
        depth_start, depth_interval = float(cams[0][1][3][0]), float(cams[0][1][3][1])
        depth_end = float(cams[0][1][3][3])
        if hip_towers_path:
            from .feature_net_train import hip_towers       # HIP forward / GroupNorm backward, ATen convolution backward
            # the towers' backward adds its 94 parameter gradients into the flat buffer itself (two launches, no autograd accumulation)
            feats = hip_towers(raw if raw is not None else images, self.params.group("unet"),
                               accumulate_into_grads=torch.is_grad_enabled())
        else:                                               # narrower towers (channel counts below the HIP kernels' tiling)
            feats = unet_forward(trainable_layers(self.params.group("unet")), images,
                                 hip_group_norm=self.device.type == "cuda")
        transforms = homography_transforms(cams_t, depth_num, depth_start, depth_interval)
        if self.regularization == "GRU":                  # train.py:355-364
            from .gru_train import inference_prob_recurrent, mvsnet_classification_loss
            prob_volume = inference_prob_recurrent(feats, transforms, self.params.group("gru"))
            loss, _mae, l1, l3, wta = mvsnet_classification_loss(prob_volume, gt, depth_num, [depth_start], [depth_interval])
            return loss, l1, l3, wta[0, :, :, 0]
        if feats.shape[-1] < 32:
            feats = torch.nn.functional.pad(feats, (0, 32 - feats.shape[-1]))      # padded channels: zero cost, zero gradient
        depth, _prob = plane_sweep_depth(feats, transforms, depth_start, depth_interval, self.params.group("regnet"),
                                         sync=sync, accumulate_into_grads=torch.is_grad_enabled())
        est = depth[None, :, :, None]
        ds = torch.tensor([depth_start], device=self.device)
        de = torch.tensor([depth_end], device=self.device)
        loss, l1, l3, _dbg = mvsnet_regression_loss(est, gt, ds, de, **self.loss_args)
        if self.refinement:                               # train.py:317-349
            from .refine import depth_refine, refine_forward, trainable_refine_layers
            net, upsample, conf, mode = self.refine_cfg
            table, layers = trainable_refine_layers(self.params.group("refine"), net)
            refined, _residual = depth_refine(est, images[0:1], _prob[None, :, :, None], depth_num, depth_start, depth_interval,
                                              lambda c, d: refine_forward(table, layers, c, d), upsample_depth=upsample,
                                              refine_with_confidence=conf,
                                              stereo_image=images[1:2] if (self.refine_with_stereo and images.shape[0] > 1) else None)
            if upsample:
                if full_depth is None:
                    raise ValueError("upsample_before_refinement needs the full-resolution ground truth")
                target = torch.as_tensor(full_depth, dtype=torch.float32, device=self.device)[None]
            else:
                target = gt
            loss1, l1r, l3r, _ = mvsnet_regression_loss(refined, target, ds, de, **self.loss_args)
            if mode == "refine_only":
                loss = loss1 + 1e-9 * loss
                l1, l3 = l1r, l3r
            elif mode == "main_only":
                loss = loss + 1e-12 * loss1
            else:
                loss = (loss + loss1) / 2
                l1, l3 = l1r, l3r
        return loss, l1, l3, depth

    def learning_rate(self):
        return exponential_decay(self.base_lr, self.global_step, self.stepvalue, self.gamma)

    def reduce_gradients(self):
        """average_gradients (train.py:155-187): ONE sum all-reduce of the flat gradient buffer (RCCL on the
        GPUs, gloo in the CPU tests); returns the 1/world factor the update kernel applies."""
        if self.world > 1:
            g = self.params.grad
            if g.is_cuda and torch.distributed.get_backend() != "nccl":      # rehearsals over gloo: stage through the host
                h = g.cpu()
                torch.distributed.all_reduce(h)
                g.copy_(h)
            else:
                torch.distributed.all_reduce(g)
        return 1.0 / self.world

    def apply_gradients(self):
        """average over ranks + optimiser update (train.py:444), one launch each."""
        lib = _lib.load()
        p, g = self.params.data, self.params.grad
        scale = self.reduce_gradients()
        lr, n = self.learning_rate(), self.params.numel
        P = _lib.ptr
        if self.optimizer == "rmsprop":
            rc = lib.mvs_rmsprop_step_f32(P(p), P(g), P(self.slots[0]), P(self.slots[1]), n, lr, 0.9, 0.0, 1e-10, scale,
                                          _lib.stream_ptr())
        elif self.optimizer == "momentum":
            rc = lib.mvs_momentum_step_f32(P(p), P(g), P(self.slots[0]), n, lr, 0.9, scale, _lib.stream_ptr())
        else:
            t = self.global_step + 1
            lr_t = lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
            rc = lib.mvs_adam_step_f32(P(p), P(g), P(self.slots[0]), P(self.slots[1]), n, lr_t, 0.9, 0.999, 1e-8, scale,
                                       _lib.stream_ptr())
        _lib.check(rc, "optimizer step")
        g.zero_()
        self.global_step += 1

    def train_step(self, images, cams, depth_image, depth_num, full_depth=None):
        loss, l1, l3, _ = self.loss(images, cams, depth_image, depth_num, full_depth)
        loss.backward()
        if self.refinement and self.refine_cfg[3] == "refine_only":     # the main network's variables are not trainable
            for (group, _l, _f), _v, o, shape in self.params.index:
                if group != "refine":
                    self.params.grad[o:o + int(np.prod(shape))].zero_()
        self.apply_gradients()
        return loss.detach(), l1.detach(), l3.detach()

    @torch.no_grad()
    def validate_step(self, images, cams, depth_image, depth_num, full_depth=None):
        """Validation is a per-replica computation (train.py:373-409 runs it on one tower): BatchNorm uses this
        replica's statistics even under --sync_bn, so a rank validating alone issues NO collective (its peers are
        already in the next training step, whose BatchNorm all-reduces have the same shapes and would pair up)."""
        loss, l1, l3, _ = self.loss(images, cams, depth_image, depth_num, full_depth, sync=None)
        return loss, l1, l3

    # -- checkpoints ----------------------------------------------------------------------------------
    def save(self, model_dir, regularization=None):
        """train.py:360-365: <model_dir>/<regularization>/<network_mode>/model.ckpt-<global step>."""
        ck = tf_checkpoint.ckpt_path(model_dir, regularization or self.regularization, self.network_mode)
        os.makedirs(os.path.dirname(ck), exist_ok=True)
        prefix = tf_checkpoint.model_path(ck, self.global_step)
        def native(arrays):                           # padded regulariser variables back to their own shapes
            return {v: np.ascontiguousarray(a[tuple(slice(0, k) for k in self.native_shapes.get(v, a.shape))])
                    for v, a in arrays.items()}
        tensors = native(self.params.named_arrays(self.params.data))
        for slot, buf in zip(OPTIMIZER_SLOTS[self.optimizer], self.slots):
            for var, arr in native(self.params.named_arrays(buf)).items():
                tensors[var + "/" + slot] = arr
        tensors["global_step"] = np.asarray(self.global_step, np.int64)
        for var in [v for v in tensors if v.endswith("/bn/gamma")]:      # created by tf.layers.batch_normalization, never
            n = tensors[var].shape                                      # updated (no UPDATE_OPS dependency)
            tensors[var[:-len("gamma")] + "moving_mean"] = np.zeros(n, np.float32)
            tensors[var[:-len("gamma")] + "moving_variance"] = np.ones(n, np.float32)
        if self.optimizer == "adam":                  # tf.train.AdamOptimizer's non-slot accumulators after `global_step` updates
            tensors["beta1_power"] = np.asarray(0.9 ** (self.global_step + 1), np.float32)
            tensors["beta2_power"] = np.asarray(0.999 ** (self.global_step + 1), np.float32)
        tf_checkpoint.write_checkpoint(prefix, tensors)
        return prefix

    def restore(self, prefix):
        """load_model (train.py:139-153): variables, optimiser slots when present, global_step."""
        have = {name for name, _shape, _dt in tf_checkpoint.list_variables(prefix)}
        want = [v for _k, v, _o, _s in self.params.index]
        slot_names = [(v + "/" + s) for s in OPTIMIZER_SLOTS[self.optimizer] for v in want]
        names = want + [s for s in slot_names if s in have] + (["global_step"] if "global_step" in have else [])
        vals = tf_checkpoint.read_checkpoint(prefix, names)

        def fill(buf, suffix):
            host = buf.detach().cpu().numpy()
            for _k, v, o, shape in self.params.index:
                if v + suffix in vals:
                    a = np.asarray(vals[v + suffix], np.float32)
                    view = host[o:o + int(np.prod(shape))].reshape(shape)
                    if suffix == "/RMSProp":
                        view[...] = 1.0                    # padded entries of the rms slot keep TensorFlow's initial one
                    elif a.shape != tuple(shape):
                        view[...] = 0.0
                    view[tuple(slice(0, k) for k in a.shape)] = a
            buf.copy_(torch.from_numpy(host))

        fill(self.params.data, "")
        for slot, buf in zip(OPTIMIZER_SLOTS[self.optimizer], self.slots):
            fill(buf, "/" + slot)
        if "global_step" in vals:
            self.global_step = int(np.asarray(vals["global_step"]).reshape(-1)[0])


# ------------------------------------------------------------------------------------------------
# program entrance (train.py:526-535)
# ------------------------------------------------------------------------------------------------

def build_parser():
    p = argparse.ArgumentParser(description="Train MVSNet on MI355X")
    a = p.add_argument
    a("--train_data_root", required=True); a("--model_dir", required=True)
    a("--model_load_dir", default=None); a("--ckpt_step", type=int, default=None)
    a("--view_num", type=int, default=3); a("--max_d", type=int, default=192)
    a("--width", type=int, default=640); a("--height", type=int, default=480)
    a("--sample_scale", type=float, default=0.25); a("--interval_scale", type=float, default=1.0)
    a("--base_image_size", type=int, default=8)
    a("--regularization", default="3DCNN"); a("--optimizer", default="rmsprop")
    a("--refinement", action="store_true"); a("--network_mode", default="lite")
    a("--refinement_network", default="unet"); a("--refinement_train_mode", default="all")
    a("--no_upsample_before_refinement", action="store_true"); a("--no_refine_with_confidence", action="store_true")
    a("--refine_with_stereo", action="store_true")
    a("--epoch", type=int, default=1); a("--max_steps_per_epoch", type=int, default=None)
    a("--base_lr", type=float, default=0.001); a("--display", type=int, default=1)
    a("--stepvalue", type=int, default=70000); a("--snapshot", type=int, default=5000)
    a("--gamma", type=float, default=0.5); a("--val_batch_size", type=int, default=100)
    a("--train_steps_per_val", type=int, default=500); a("--dataset_fraction", type=float, default=1.0)
    a("--loss_type", default="power"); a("--alpha", type=float, default=0.25); a("--beta", type=float, default=0.0)
    a("--eta", type=float, default=0.02); a("--no_grad_loss", action="store_true")
    a("--seed", type=int, default=0)
    a("--loader_workers", type=int, default=None,
      help="worker processes that prepare the next clusters while the GPU trains (default: the host's share; 0 = threads)")
    a("--no_prefetch", action="store_true", help="prepare every cluster on the training thread, as one generator would")
    a("--sync_bn", action="store_true", help="BatchNorm statistics over all replicas (not in the reference)")
    return p


def train(args):
    from .mvs_data_generation import ClusterGenerator
    from .shard import shard_indices
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    tr = Trainer(args.network_mode, "cuda", args.optimizer, args.base_lr, args.stepvalue, args.gamma, args.loss_type,
                 args.alpha, args.beta, args.eta, not args.no_grad_loss, seed=args.seed, sync_bn=args.sync_bn, refinement=args.refinement,
                 refinement_network=args.refinement_network, upsample_before_refinement=not args.no_upsample_before_refinement,
                 refine_with_confidence=not args.no_refine_with_confidence, refinement_train_mode=args.refinement_train_mode,
                 refine_with_stereo=args.refine_with_stereo, regularization=args.regularization)
    if args.ckpt_step:
        ck = tf_checkpoint.ckpt_path(args.model_load_dir or args.model_dir, args.regularization, args.network_mode)
        tr.restore(tf_checkpoint.model_path(ck, args.ckpt_step))
    gen_args = lambda mode: dict(data_dir=args.train_data_root, view_num=args.view_num, image_width=args.width,
                                 image_height=args.height, depth_num=args.max_d, interval_scale=args.interval_scale,
                                 base_image_size=args.base_image_size, mode=mode, output_scale=args.sample_scale,
                                 sessions_frac=args.dataset_fraction, seed=args.seed,
                                 flip_cams=args.regularization == "GRU")                            # train.py:194-196
    mk = lambda mode: ClusterGenerator(**gen_args(mode))
    train_gen, val_gen = mk("train"), mk("val")
    mine = shard_indices(len(train_gen.clusters), rank, world)
    steps = len(mine) if args.max_steps_per_epoch is None else min(len(mine), args.max_steps_per_epoch)
    if world > 1:                                   # every rank must take the same number of steps
        t = torch.tensor([steps], device="cuda"); torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
        steps = int(t.item())
    from .mvs_data_generation import TrainingPrefetcher
    # the input pipeline (train.py:208-247): the next clusters are decoded / resized / cropped by worker processes while this
    # thread launches the step; the images travel as uint8 and are standardised on the device (Trainer.loss).  A cluster that
    # fails to load is skipped -- except with several ranks, where a skipped step would desynchronise the all-reduce
    feed = TrainingPrefetcher(train_gen, gen_args("train"), mine[:steps], workers="inline" if args.no_prefetch else args.loader_workers,
                              center=False, strict=world > 1)
    for epoch in range(args.epoch):
        t0 = time.time()
        for step, (images, cams, depth, _full) in feed:
            loss, l1, l3 = tr.train_step(images, cams, depth, args.max_d, _full)
            if args.regularization == "GRU":           # the same cluster with the sweep reversed (cluster_generator.py:303-304)
                from .mvs_data_generation import flip_cams
                loss, l1, l3 = tr.train_step(images, flip_cams(cams, args.max_d), depth, args.max_d, _full)
            if step % args.display == 0 and rank == 0:
                print("epoch, %d, step %d, total_step %d, loss = %.4f, (< 1px) = %.4f, (< 3px) = %.4f (%.3f sec/step)"
                      % (epoch, step, tr.global_step, float(loss), float(l1), float(l3), time.time() - t0), flush=True)
            bad = not math.isfinite(float(loss))
            if world > 1:                              # decide together: a rank leaving alone strands its peers in the next all-reduce
                flag = torch.tensor([1.0 if bad else 0.0], device="cuda")
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
                bad = bool(flag.item() > 0)
            if bad:
                if world > 1:
                    torch.distributed.destroy_process_group()
                raise SystemExit(1)                    # train.py:486-488
            if rank == 0 and (tr.global_step % args.snapshot == 0 or step == steps - 1):
                print("Saving model to", tr.save(args.model_dir, args.regularization), flush=True)
            if rank == 0 and (step + 1) % args.train_steps_per_val == 0:
                vals = []
                val_feed = TrainingPrefetcher(val_gen, gen_args("val"), range(min(len(val_gen.clusters), args.val_batch_size)),
                                              workers="inline" if args.no_prefetch else args.loader_workers, center=False)
                for _k, (vi, vcam, vd, _vf) in val_feed:   # the other ranks wait in their next all-reduce meanwhile: same pipeline
                    vals.append([float(x) for x in tr.validate_step(vi, vcam, vd, args.max_d, _vf)])
                val_feed.close()
                if vals:
                    m = np.mean(np.asarray(vals), axis=0)
                    print("VAL STEP COMPLETED. Average loss: %g, Average less one: %g, Average less three: %g"
                          % (m[0], m[1], m[2]), flush=True)
            t0 = time.time()                           # sec/step: from the end of one step to the end of the next
    feed.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def main(argv=None):
    train(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
