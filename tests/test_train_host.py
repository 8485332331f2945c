"""CPU tests of the training host logic: the train/val branches of the cluster generator
(cluster_generator.py:61-64,166-223), the learning-rate schedule and initialisers (train.py:248-271), the flat
parameter buffer, and gradient averaging across two ranks over gloo (average_gradients, train.py:155-187).
No compute calls into the HIP library here."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _helpers import add_depths, make_session

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_train_and_val_modes(tmp_path):
    from mvsnet_amd import mvs_data_generation as G
    for mode, names in (("train", ("a", "b")), ("val", ("c",))):
        for k, nm in enumerate(names):
            s = make_session(str(tmp_path / mode / nm), seed=k)
            add_depths(s, seed=k)
    gen = G.ClusterGenerator(str(tmp_path), view_num=3, image_width=64, image_height=48, depth_num=16,
                             base_image_size=8, mode="train", seed=3)
    assert len(gen.clusters) == 6                                         # 3 clusters with views per session
    items = list(gen)
    assert len(items) == 6
    images, cams, depth, full = items[0]
    assert images.shape == (3, 48, 64, 3) and images.dtype == np.float32
    assert abs(float(images[0, :, :, 0].mean())) < 1e-4                    # centred per image and channel
    assert cams.shape == (3, 2, 4, 4)
    assert depth.shape == (12, 16, 1) and full.shape == (48, 64, 1)
    assert np.array_equal(depth[:, :, 0], full[::4, ::4, 0])              # nearest-neighbour 1/4 down-sampling
    vals = full[full > 0]
    assert vals.min() > 400 and vals.max() <= 900 and (full == 0).any()   # masked to (min_depth, max_depth]
    # same seed -> same order; another seed -> (almost surely) another order
    order = lambda g: [(c.session_dir, c.ref_index) for c in g.clusters]
    again = G.ClusterGenerator(str(tmp_path), 3, 64, 48, 16, base_image_size=8, mode="train", seed=3)
    assert order(again) == order(gen)
    assert len(G.ClusterGenerator(str(tmp_path), 3, 64, 48, 16, base_image_size=8, mode="val").clusters) == 3
    flipped = list(G.ClusterGenerator(str(tmp_path), 3, 64, 48, 16, base_image_size=8, mode="val", flip_cams=True))
    assert len(flipped) == 6
    c0, c1 = flipped[0][1], flipped[1][1]
    assert c1[0, 1, 3, 1] == -c0[0, 1, 3, 1]
    assert math.isclose(c1[0, 1, 3, 0], c0[0, 1, 3, 0] + 15 * c0[0, 1, 3, 1], rel_tol=1e-6)
    with pytest.raises(ValueError):
        G.ClusterGenerator(str(tmp_path), mode="bogus")


@pytest.mark.parametrize("workers", [2, 0, "inline"])
def test_training_prefetcher_hands_out_the_generators_batches_in_order(tmp_path, workers):
    """The input pipeline of the training loop (train.py:208-247: tf.data from_generator + parallel_interleave + prefetch):
    worker processes (2), threads (0) or the calling thread ("inline") prepare the clusters; what comes out is what
    `prepare_training` gives on this thread, in the order asked for -- centred float32, or uint8 for the device
    (center=False: the cropped images before utils.py:33-38).  A cluster whose ground truth is missing is skipped
    (cluster_generator.py:217-220), or raised when `strict` (several ranks)."""
    from mvsnet_amd import mvs_data_generation as G
    for k, nm in enumerate(("a", "b")):
        s = make_session(str(tmp_path / "train" / nm), seed=k)
        add_depths(s, seed=k)
    args = dict(data_dir=str(tmp_path), view_num=3, image_width=64, image_height=48, depth_num=16, base_image_size=8,
                mode="train", seed=3)
    gen = G.ClusterGenerator(**args)
    order = [4, 0, 5, 2, 1, 3, 0]
    for center in (True, False):
        feed = G.TrainingPrefetcher(gen, args, order, workers=workers, ahead=3, center=center)
        got = list(feed)
        feed.close()
        assert [p_ for p_, _b in got] == list(range(len(order)))
        for (p_, batch), ci in zip(got, order):
            want = gen.prepare_training(gen.clusters[ci], center=center)
            assert len(batch) == 4 and all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(batch, want))
        assert got[0][1][0].dtype == (np.float32 if center else np.uint8)
    # uint8 + the device's standardisation = the centred batch (float64 moments against numpy's float32: 2e-5 at this size)
    u8 = gen.prepare_training(gen.clusters[0], center=False)[0]
    f32 = gen.prepare_training(gen.clusters[0])[0]
    x = u8.astype(np.float64)
    np.testing.assert_allclose((x - x.mean((1, 2), keepdims=True)) / (x.std((1, 2), keepdims=True) + 1e-8), f32, atol=2e-5)
    # a cluster without ground truth
    c = gen.clusters[order[2]]
    os.remove(c.depth_path(c.ref_index))
    feed = G.TrainingPrefetcher(gen, args, order, workers=workers, ahead=2)
    assert [p_ for p_, _b in feed] == [0, 1, 3, 4, 5, 6]
    feed.close()
    with pytest.raises(OSError):
        list(G.TrainingPrefetcher(gen, args, order, workers=workers, ahead=2, strict=True))


def test_schedule_and_initialisers():
    from mvsnet_amd import train as T
    from mvsnet_amd.synthetic import make_regnet_params
    assert T.exponential_decay(1e-3, 0, 70000, 0.5) == 1e-3
    assert math.isclose(T.exponential_decay(1e-3, 70000, 70000, 0.5), 5e-4)
    assert math.isclose(T.exponential_decay(1e-3, 35000, 70000, 0.5), 1e-3 * 0.5 ** 0.5)   # not a staircase
    init = T.glorot_uniform_like(make_regnet_params("normal"), seed=0)
    w = init["3dconv0_1"]["w"]
    limit = math.sqrt(6.0 / (27 * 32 + 27 * 8))
    assert w.shape == (3, 3, 3, 32, 8) and np.abs(w).max() <= limit and np.abs(w).max() > 0.9 * limit
    assert np.all(init["3dconv0_1"]["gamma"] == 1) and np.all(init["3dconv0_1"]["beta"] == 0)
    assert "gamma" not in init["3dconv6_2"]
    with pytest.raises(NotImplementedError):
        T.Trainer(optimizer="sgd", device="cpu")


def test_flat_parameters_views_and_names():
    from mvsnet_amd import train as T
    tr = T.Trainer("normal", device="cpu", seed=1)
    fp = tr.params
    assert fp.numel == sum(int(np.prod(s)) for _k, _v, _o, s in fp.index)
    reg = fp.group("regnet")
    w = reg["3dconv0_1"]["w"]
    assert w.shape == (3, 3, 3, 32, 8) and w.requires_grad and w.is_leaf
    (w * 2.0).sum().backward()                                            # autograd accumulates INTO the flat buffer
    off = [o for k, _v, o, _s in fp.index if k == ("regnet", "3dconv0_1", "w")][0]
    assert torch.all(fp.grad[off:off + w.numel()] == 2.0) and fp.grad.sum() == 2.0 * w.numel()
    names = fp.named_arrays(fp.data)
    assert "3dconv0_1/kernel" in names and "3dconv0_1/bn/gamma" in names and "2dconv1_0/gn/beta" in names
    assert names["3dconv0_1/kernel"].shape == (3, 3, 3, 32, 8)
    assert tr.slots[0].min() == 1.0 and tr.slots[1].max() == 0.0          # TF: rms starts at one, momentum at zero


_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
from mvsnet_amd import shard as sh
dist = sh.init_process_group("gloo")
rank, local, world = sh.rank_world()
from mvsnet_amd import train as T
tr = T.Trainer("normal", device="cpu", seed=5)            # same seed -> identical replicas
assert tr.world == 2
w = tr.params.group("regnet")["3dconv6_2"]["w"]
(w * float(rank + 1)).sum().backward()                    # rank 0: grad 1, rank 1: grad 2
scale = tr.reduce_gradients()
avg = (tr.params.grad * scale)
off = [o for k, _v, o, _s in tr.params.index if k == ("regnet", "3dconv6_2", "w")][0]
ck = float(tr.params.data.double().sum())
t = torch.tensor([ck], dtype=torch.float64); lo = t.clone(); hi = t.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({"avg": float(avg[off]), "total": float(avg.sum()), "n": int(w.numel()),
                      "replicas_equal": bool(lo.item() == hi.item())}))
dist.destroy_process_group()
"""


def test_gradient_averaging_world_size_2_over_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    from _helpers import run_ranks
    outs = run_ranks(_WORKER % {"root": ROOT}, env, world=2, timeout=240)
    assert all(rc == 0 for rc, _o, _e in outs), outs
    res = json.loads(outs[0][1].strip().splitlines()[-1])
    assert res["replicas_equal"]
    assert math.isclose(res["avg"], 1.5) and math.isclose(res["total"], 1.5 * res["n"])


def test_saved_checkpoint_holds_the_reference_graphs_global_variables(tmp_path):
    """predictlib.py:69-76 restores tf.train.Saver(tf.global_variables()): besides the trainable variables that
    list holds every BatchNorm layer's moving_mean / moving_variance (tf.layers.batch_normalization creates them,
    network.py:492-509, although training=True never reads them), the optimiser slots, Adam's beta powers and
    global_step.  Trainer.save must write that whole name set or the reference restore fails with NotFound."""
    from mvsnet_amd.train import Trainer
    from mvsnet_amd import tf_checkpoint as ck
    for opt, slots in (("rmsprop", ("RMSProp", "RMSProp_1")), ("adam", ("Adam", "Adam_1"))):
        tr = Trainer("normal", "cpu", opt, 1e-3, 10000, 0.9, "l1", 1.0, 1.0, 1.0, True, seed=0)
        prefix = tr.save(str(tmp_path / opt), "3DCNN")
        names = {n for n, _s, _d in ck.list_variables(prefix)}
        trainable = set(ck.variable_names("normal", "3DCNN").values())
        bn_layers = sorted(v[:-len("/gamma")] for v in trainable if v.endswith("/bn/gamma"))
        assert len(bn_layers) == 10                                  # RegNetUS0's conv_bn / deconv_bn layers
        want = set(trainable) | {"global_step"}
        want |= {v + "/" + s for v in trainable for s in slots}
        want |= {b + "/moving_mean" for b in bn_layers} | {b + "/moving_variance" for b in bn_layers}
        if opt == "adam":
            want |= {"beta1_power", "beta2_power"}
        assert names == want
        got = ck.read_checkpoint(prefix, [bn_layers[0] + "/moving_mean", bn_layers[0] + "/moving_variance"])
        assert np.all(got[bn_layers[0] + "/moving_mean"] == 0) and np.all(got[bn_layers[0] + "/moving_variance"] == 1)


def test_validation_never_takes_the_cross_replica_batchnorm_path():
    """A rank validating alone (train.py:373-409 validates on one tower) must not issue collectives its peers do not
    match: validate_step passes sync=None whatever --sync_bn configured."""
    import inspect
    from mvsnet_amd.train import Trainer
    src = inspect.getsource(Trainer.validate_step)
    assert "sync=None" in src
    seen = {}

    class T(Trainer):
        def __init__(self):
            self.sync = object()                                    # stands for an active SyncBN reducer

        def loss(self, *a, sync="self", **k):
            seen["sync"] = sync
            z = torch.zeros(())
            return z, z, z, z
    T().validate_step(None, None, None, 8)
    assert seen["sync"] is None
