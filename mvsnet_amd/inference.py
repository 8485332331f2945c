"""Entry point mirroring `python -m mvsnet.inference` (mvsnet/inference.py:83-145):

    python -m mvsnet_amd.inference --input_dir <session> [--output_dir ...] --view_num 5 --max_d 192 \
        --width 640 --height 512 --regularization 3DCNN [--weights weights.npz]

One process per GPU; under torch.distributed.run the clusters (reference views) of the session are
sharded round-robin across ranks with no collective on the data path.  --model_dir /
--ckpt_step restore a TensorFlow checkpoint of the reference (mvsnet_amd/tf_checkpoint.py, no TF
needed); without them (and without --weights) the networks are randomly initialised -- no checkpoint
exists offline -- which still exercises the full pipeline and output formats.
"""
from __future__ import annotations

import argparse
import logging
import os
import time

import numpy as np

logger = logging.getLogger("mvsnet_amd.inference")


def build_weights(config, device, weights_path=None, model_dir=None, ckpt_step=None, extractor="hip"):
    """Weights from (in order of preference) a TensorFlow checkpoint of the reference
    (<model_dir>/<regularization>/<network_mode>/model.ckpt-<ckpt_step>, inference.py:23-27 +
    utils.py:75-96), an .npz of parameter dictionaries, or a seeded random initialisation."""
    from . import synthetic as S
    from .model import MVSNetWeights
    if model_dir:
        from . import tf_checkpoint as ck
        prefix = ck.model_path(ck.ckpt_path(model_dir, config.regularization, config.network_mode), ckpt_step)
        params = ck.load_mvsnet_params(prefix, config.network_mode, config.regularization,
                                       refinement=config.refinement_network if config.refinement else None)
        logger.info("restored %s", prefix)
        return MVSNetWeights.from_numpy(config.network_mode, unet=params["unet"], regnet=params["regnet"],
                                        gru=params["gru"], device=device, refine=params.get("refine"),
                                        refine_type=config.refinement_network, extractor=extractor)
    if weights_path:
        z = np.load(weights_path, allow_pickle=True)
        unet, regnet, gru = z["unet"].item(), z["regnet"].item(), z["gru"].item()
    else:
        unet = S.make_unet_params(config.network_mode, seed=3)
        regnet = S.make_regnet_params(config.network_mode, seed=1)
        gru = S.make_gru_params(config.network_mode, seed=2,
                                in_channels=4 * S.base_filter(config.network_mode))
    refine = None
    if config.refinement:
        from .refine import make_refine_params
        refine = make_refine_params(config.refinement_network, config.network_mode,
                                    4 + int(config.refine_with_confidence), seed=4)
    return MVSNetWeights.from_numpy(config.network_mode, unet=unet, regnet=regnet, gru=gru, device=device,
                                    refine=refine, refine_type=config.refinement_network, extractor=extractor)


def center_images_device(u8):
    """mvs_data_generation/utils.py:33-38 (per-image, per-channel standardisation) on the device: (n,H,W,3) uint8 ->
    float32 (x - mean) / (sqrt(var) + 1e-8) with the moments accumulated in float64.  PyTorch restatement of
    mvs_center_images_u8_f32 (csrc/center_images.hip, which the HIP towers call themselves when they are given uint8): used for the
    torch extractor and as the bit-for-bit cross-check in tests.  (The reference's numpy float32 reductions keep running sums
    over the leading axes: ~1e-3 relative away from the exact moments at 640 x 512; tests/test_gpu_unet.py.)"""
    import torch
    x = u8.to(torch.float32)
    # float64 ACCUMULATION of float32 terms that are exact (grey levels and their squares are integers below 2^16): the sums are
    # exact, the moments carry float64 rounding only -- without materialising a float64 copy of the batch (round 3 did: 8 bytes
    # per element through five elementwise kernels)
    n = float(x.shape[1] * x.shape[2])
    mean = x.sum(dim=(1, 2), keepdim=True, dtype=torch.float64) / n
    var = ((x * x).sum(dim=(1, 2), keepdim=True, dtype=torch.float64) / n - mean * mean).clamp_(min=0.0)
    return ((x - mean.float()) / (var.sqrt().float() + 0.00000001)).contiguous()


_DEVICE_WARM = set()


def warm_device_one_offs(device, centre):
    """Once per process and device: the torch kernels of the session's glue (a stack, a copy; with `centre` -- the torch
    extractor -- the reductions and elementwise kernels of `center_images_device`) are launched on a 16 x 16 image so that their
    code objects load -- 40-100 ms of first-use cost each on this stack -- WHILE the worker processes decode the session's
    first images, not after they have arrived (tools/r6_first_pass.py: a one-scan process's first pass)."""
    import torch
    if (device, centre) in _DEVICE_WARM:
        return
    _DEVICE_WARM.add((device, centre))
    z = torch.zeros((2, 16, 16, 3), dtype=torch.uint8, device=device)
    f = center_images_device(z) if centre else z.to(torch.float32)
    torch.stack([f[0], f[1]]).clone()
    torch.cuda.Event(enable_timing=True).record()


class FeatureCache:
    """Per-image feature maps of a session, least-recently-used first out.

    `fill(group, tower)` makes every key of one group of reference views available and PINS them in `self.group` until the next
    `fill`: eviction only ever removes entries that the current group does not use, and `get` reads the pinned dict, so a key
    that was a hit when the group was formed cannot disappear before the group's last reference view has read it (round 3's FIFO
    of 256 entries evicted while it inserted: a session whose covisibility lists reach back more than `limit` images -- loop
    closures, score-ranked pair.txt -- raised KeyError outside the per-view try / except and ended the whole run)."""

    def __init__(self, limit=256):
        from collections import OrderedDict
        self.limit = max(1, int(limit))
        self.entries = OrderedDict()
        self.group = {}
        self.hits = self.misses = 0

    def fill(self, keyed_images, tower):
        """keyed_images: iterable of (key, image) over all views of the group (repeats allowed); tower(list of images) ->
        indexable batch of feature maps, called once for the images the cache misses."""
        self.group = {}
        need = {}
        for k_, img in keyed_images:
            if k_ in self.group or k_ in need:
                continue
            if k_ in self.entries:
                self.entries.move_to_end(k_)
                self.group[k_] = self.entries[k_]
                self.hits += 1
            else:
                need[k_] = img
                self.misses += 1
        if need:
            fb = tower(list(need.values()))
            for j, k_ in enumerate(need):
                self.group[k_] = self.entries[k_] = fb[j]
        while len(self.entries) > self.limit:                    # oldest first; the group's own keys are the newest
            self.entries.popitem(last=False)

    def get(self, keys):
        return [self.group[k_] for k_ in keys]


class SessionLoader:
    """Clusters of a session-format generator prepared with the per-IMAGE work on worker processes (host_pool.HostPool):
    `submit(c)` hands the cluster's images that no earlier cluster asked for to the pool and returns at once, `result(handle)`
    gives what `gen.prepare(c, center=False)` gives -- the same functions, run in the workers -- except that the image stacks
    come back as LISTS of per-view uint8 arrays (no (N,H,W,3) copy on this thread).  Image sizes (needed for the cluster's
    scale-to-cover factor before anything is decoded, mvs_cluster.py:178-192) come from the files' headers."""

    def __init__(self, gen, pool, limit=192, pin_device=None):
        from collections import OrderedDict
        self.gen, self.pool, self.limit, self.pin_device = gen, pool, limit, pin_device
        self.cams = {}                               # (session, index, depth range) -> camera: one build per image and call
        self.images = OrderedDict()                  # (session, index, rescale) -> Future of (cropped, output image, shape, seconds)
        self.sizes = {}
        self.load_seconds = 0.0
        self._counted = set()

    def _size(self, c, i):
        key = (c.session_dir, i)
        hit = self.sizes.get(key)
        if hit is None:
            from PIL import Image
            with Image.open(c.image_path(i)) as im:  # header only: nothing is decoded
                hit = self.sizes[key] = (im.size[1], im.size[0], 3)
        return hit

    def submit(self, c):
        g = self.gen
        sizes = [self._size(c, i) for i in c.indices]
        c.original_image_shape = sizes[0]
        c.rescale = max(max(float(g.image_height) / s_[0] for s_ in sizes), max(float(g.image_width) / s_[1] for s_ in sizes))
        futs = []
        for i in c.indices:
            key = (c.session_dir, i, round(float(c.rescale), 12))
            f = self.images.get(key)
            if f is None:
                f = self.images[key] = self.pool.load_image(c.image_path(i), c.rescale, g.image_width, g.image_height,
                                                            g.base_image_size, g.output_scale, pin_device=self.pin_device)
                while len(self.images) > self.limit:
                    self.images.popitem(last=False)
            else:
                self.images.move_to_end(key)
            futs.append((key, f))
        return c, sizes, futs

    def result(self, handle):
        c, sizes, futs = handle
        ins, outs = [], []
        for key, f in futs:
            cr, oi, _shape, sec = f.result()
            if key not in self._counted:             # worker seconds, once per decoded image
                self._counted.add(key)
                self.load_seconds += sec
            ins.append(cr); outs.append(oi)
        cams = []
        for i in c.indices:                          # a camera is listed by ~view_num clusters: built once (round 5: 5x, on this thread)
            ck = (c.session_dir, i, c.min_depth, c.max_depth, c.depth_num, c.interval_scale)
            cam = self.cams.get(ck)
            if cam is None:
                cam = self.cams[ck] = c.load_camera(i)
            cams.append(cam)
        full_cams, out_cams = self.gen.cluster_cameras(c, cams, sizes)
        return outs, ins, out_cams, full_cams, c.ref_index


# Pinned host buffers are expensive to create (~1.5 ms each) and cheap to keep: the staging buffers of the uploads and the result
# buffers of the downloads live for the process, not for one compute_depth_maps call (a session is one call).
_PINNED_STAGING = {}
_PINNED_RESULTS = {}


def compute_depth_maps(input_dir, config=None, weights=None, device=None, timings=None, gru_views=4,
                       feature_cache_limit=256, host_workers=None, tower_stream=True, **kwargs):
    """mvsnet/inference.py:83-119.  Returns the number of depth maps this rank wrote.

    `timings` (a dict) receives the stage breakdown of the run in seconds: wall, load (decode + resize + crop + centre on the
    loader threads, summed over threads), wait_load (this thread blocked on the loaders), towers / hot_path / d2h (GPU time from
    stream events, host -> device copies of the images included in towers), host_gpu_submit (this thread enqueueing), write (file
    writers, summed over threads).  `gru_views`: reference views per recurrent sweep (mvs_gru_wta_batch_f32) with the GRU
    regulariser.  `host_workers`: worker PROCESSES for image decoding / rescaling and output encoding (host_pool; None = by
    core count, 0 = round 3's loader / writer threads inside this process); `timings` then also receives host_cpu = CPU
    seconds of this process and its workers.  `tower_stream`: the image uploads and the UNetDS2GN towers of a group of reference
    views go to a HIP stream of their own and the hot path waits for their event -- this thread runs several views ahead of the
    GPU, so the (launch-bound) towers of group g + 1 execute beside the hot path of group g instead of in line with it."""
    import threading
    import torch
    from . import predictlib as pl
    from . import shard as sh
    from .mvs_data_generation import make_generator

    config = config or pl.InferenceConfig()
    for k, v in kwargs.items():                       # predictlib.init_inference: kwargs -> flags
        if not hasattr(config, k):
            raise AttributeError("unknown flag %s" % k)
        setattr(config, k, v)
    rank, local_rank, world = sh.rank_world()
    if device is None:
        device = sh.bind_device(local_rank)           # one process drives one GPU
    else:
        device = torch.device(device)                 # 'cuda' without an index = the current device
        idx = device.index if device.index is not None else torch.cuda.current_device()
        torch.cuda.set_device(idx)                    # library launches go to the current device
        device = torch.device("cuda", idx)
    output_dir = pl.setup_output_dir(input_dir, config.output_dir)
    gen = make_generator(input_dir, config.view_num, config.width, config.height, config.max_d,
                           config.interval_scale, config.base_image_size, mode="inference",
                           output_scale=config.sample_scale,
                           max_clusters_per_session=config.max_clusters_per_session)
    clusters = sorted(gen.clusters, key=lambda c: (c.session_dir, c.ref_index))
    mine = sh.shard(clusters, rank, world)
    if weights is None:
        weights = build_weights(config, device)
    t_wall = time.perf_counter()
    tm = {"load": 0.0, "wait_load": 0.0, "host_gpu_submit": 0.0, "write": 0.0,
          # parts of host_gpu_submit (this thread): image upload + tower launches, feature stack, hot-path launches, result hand-over
          "submit_towers": 0.0, "submit_features": 0.0, "submit_depth": 0.0, "submit_finish": 0.0}
    tm_lock = threading.Lock()

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        try:
            return fn(*a)
        finally:
            with tm_lock:
                tm[name] += time.perf_counter() - t0

    # Per-image feature cache: the reference re-runs the UNetDS2GN tower on
    # every source image of every cluster (model.py:392-406); an image's features only depend on
    # the image and on the (rescale, crop) it received, so they are computed once per session and
    # re-used by all the reference views that list the image as a source.
    feature_cache = FeatureCache(feature_cache_limit)
    done = 0
    # Host pipeline: like the reference, whose generator runs in a tf.data thread with a prefetch buffer
    # (predictlib.py:48-51), image loading / resizing of the next clusters and the file writes of the previous ones run
    # on worker threads (numpy / PIL / file IO release the GIL); this thread only drives the GPU and never waits for it:
    # results go device -> PINNED host buffers with an asynchronous copy, and the writer thread that owns a buffer waits for
    # the copy's event (round 2 called .cpu() here, which held this thread -- and with it the next reference view's launches
    # -- until the GPU had finished the current one).
    from concurrent.futures import ThreadPoolExecutor
    from . import host_pool
    from .mvs_data_generation import Cluster
    n_loaders = max(2, min(8, (os.cpu_count() or 4) // 2))
    n_writers = 4
    # Round 4: the per-image work of the loaders and the whole of the writers run in worker PROCESSES (host_pool): with them on
    # threads of this process the thread that feeds the GPU spent more time waiting for the GIL than working.  The loader /
    # writer threads below remain as the thin ends of that pipe (they wait for futures and copy events), and as the whole
    # pipe for upstream-format projects and for host_workers = 0.
    pool = host_pool.get_pool(host_workers) if any(type(c_) is Cluster for c_ in mine) else None
    session_loader = SessionLoader(gen, pool, pin_device=device.index) if pool is not None else None
    cpu0 = None
    if timings is not None:
        try:
            import psutil
            cpu0 = (sum(psutil.Process().cpu_times()[:2]), pool.cpu_seconds() if pool is not None else 0.0)
        except Exception:                             # noqa: BLE001
            cpu0 = None
    loader, writer = ThreadPoolExecutor(max_workers=n_loaders), ThreadPoolExecutor(max_workers=n_writers)
    pending, writes, pool_writes = [], [], []
    ahead = 16
    slots = threading.BoundedSemaphore(8)             # pinned result buffers in flight
    ev_marks = []                                     # per reference view: events at the stage boundaries

    device_center = lambda c_: type(c_) is Cluster        # session format: uint8 up, standardised on the device

    class _Handle:                                        # a cluster whose images are on their way through the worker processes
        def __init__(self, h):
            self.h = h

        def result(self):
            return session_loader.result(self.h)

    def submit_next(it):
        for c_ in it:
            if device_center(c_) and session_loader is not None:
                try:
                    pending.append((c_, _Handle(session_loader.submit(c_))))
                except Exception as e:                    # an unreadable header: skip-and-log like a failed load
                    logger.warning("skipping cluster %s/%d: %s", c_.session_dir, c_.ref_index, e)
                    continue
            elif device_center(c_):
                pending.append((c_, loader.submit(timed, "load", gen.prepare, c_, False)))
            else:
                pending.append((c_, loader.submit(timed, "load", gen.prepare, c_)))
            return

    def keys_of(c, in_images):
        ids = getattr(c, "indices", None) or [p_ for p_ in getattr(c, "paths", [])[0::2]]
        return [(c.session_dir, ids[v], round(float(c.rescale), 9), in_images[v].shape) if v < len(ids) else ("view", id(c), v)
                for v in range(config.view_num)]

    staging = _PINNED_STAGING                         # (shape, dtype) -> [pinned buffers, their last copy's event, next index]

    def to_device(imgs):
        """list / array of images -> device float32, standardised there when they come as uint8.  The upload goes through two
        alternating pinned staging buffers (asynchronous copy; a buffer is re-filled only after its previous copy's event)."""
        imgs = list(imgs)
        n, shp, dt = len(imgs), imgs[0].shape, imgs[0].dtype
        pins = [getattr(im, "pinned", None) for im in imgs]
        if all(p_ is not None for p_ in pins):        # already in pinned memory (host_pool.PinnedArray): no copy on this thread
            dst = torch.empty((n,) + tuple(shp), dtype=pins[0].dtype, device=device)
            for j, p_ in enumerate(pins):
                dst[j].copy_(p_, non_blocking=True)
            return dst
        st_ = staging.setdefault((shp, dt), [[None, None], [None, None], 0])
        i_ = st_[2]; st_[2] ^= 1
        cap = 0 if st_[0][i_] is None else st_[0][i_].shape[0]
        if cap < n:
            st_[0][i_] = torch.empty((max(n, 16),) + shp, dtype=torch.from_numpy(np.empty(0, dt)).dtype).pin_memory()
        elif st_[1][i_] is not None:
            st_[1][i_].synchronize()
        buf = st_[0][i_]
        view = buf.numpy()
        for j, im in enumerate(imgs):
            view[j] = im
        t_ = buf[:n].to(device, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(); st_[1][i_] = ev
        return t_

    towers_take_u8 = bool(getattr(weights.unet, "takes_uint8", False))    # HipUNetDS2GN standardises uint8 input in the library

    def images_to_device(imgs):
        t_ = to_device(imgs)
        if t_.dtype == torch.uint8:
            return t_ if towers_take_u8 else center_images_device(t_)
        return t_.to(torch.float32)

    def prefetch_features(group):
        """The images of these reference views that the cache misses go through the towers as ONE batch (a tower pass is ~31
        launches on a ~25 us floor each: one image costs nearly as much as sixteen)."""
        feature_cache.fill(((k_, res_[1][v]) for c_, res_ in group for v, k_ in enumerate(keys_of(c_, res_[1]))),
                           lambda imgs: weights.unet(images_to_device(imgs)))

    # the towers' stream (see `tower_stream`); feature maps are allocated there and read on the compute stream: record_stream keeps
    # the caching allocator from handing a freed map's memory out again while the compute stream may still be reading it
    t_stream = torch.cuda.Stream(device) if (tower_stream and weights.unet is not None) else None

    def features_of(c, in_images):
        """(N,H/4,W/4,C) of one reference view from the per-image cache (pinned for the group by prefetch_features)."""
        maps = feature_cache.get(keys_of(c, in_images))
        if t_stream is not None:
            cur = torch.cuda.current_stream(device)
            for m_ in maps:
                m_.record_stream(cur)
        return torch.stack(maps).contiguous()

    pinned = _PINNED_RESULTS                          # shape -> free pinned (depth, prob) buffer pairs, re-used across reference views

    def finish(d, p, out_images, in_images, out_cams, full_cams, index, marks):
        """device results -> pinned host buffers (asynchronous copy on the compute stream, no host wait) -> writer thread"""
        slots.acquire()
        with tm_lock:
            free = pinned.setdefault(tuple(d.shape), [])
            pair = free.pop() if free else None
        if pair is None:
            pair = (torch.empty(d.shape, dtype=torch.float32).pin_memory(), torch.empty(p.shape, dtype=torch.float32).pin_memory())
        dh, ph = pair
        dh.copy_(d, non_blocking=True); ph.copy_(p, non_blocking=True)
        copied = torch.cuda.Event(enable_timing=timings is not None)
        copied.record()
        if marks is not None:
            marks.append(copied)

        def write():
            released = False

            def release():
                nonlocal released
                if not released:
                    released = True
                    with tm_lock:
                        pinned[tuple(dh.shape)].append(pair)
                    slots.release()
            try:
                copied.synchronize()
                full = bool(config.refinement and config.upsample_before_refinement)      # full-size outputs (predictlib.py:107-115)
                if full:
                    from .mvs_data_generation import center_image       # the reference writes the STANDARDISED input image here
                    img0 = center_image(in_images[0]) if in_images[0].dtype == np.uint8 else in_images[0]
                    args = (img0, full_cams[0], index, config.visualize, 1.0 / config.sample_scale)
                else:
                    args = (out_images[0], out_cams[0], index, config.visualize, None)
                if pool is None:
                    return timed("write", pl.write_output_slice, output_dir, dh.numpy(), ph.numpy(), *args)
                # private copies (the executor pickles them later, on its feeder thread), then the pinned pair is free again
                dn, pn = np.array(dh.numpy()), np.array(ph.numpy())
                release()
                f_ = pool.write_outputs(output_dir, dn, pn, np.asarray(args[0]), np.asarray(args[1]), *args[2:])
                with tm_lock:
                    pool_writes.append(f_)
            finally:
                release()
        writes.append(writer.submit(write))

    def mark():
        if timings is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    gru_batch = []                                    # GRU: reference views waiting to share one sweep
    tower_marks = []                                  # (start, end) events of tower passes not yet attributed to a flushed sweep

    def flush_gru():
        if not gru_batch:
            return
        from .model import DepthPlan, wta_depth_values
        t0 = time.perf_counter()
        feats0 = gru_batch[0][0]
        _, Hf, Wf, Cf = feats0.shape
        D = gru_batch[0][2]
        key = ("gru", config.view_num, D, Hf, Wf, Cf, id(weights), max(1, gru_views))
        plan = plan_cache.get(key)
        if plan is None:
            plan = plan_cache[key] = DepthPlan(config.view_num, D, Hf, Wf, Cf, weights, "GRU", device, views=max(1, gru_views))
        dvs = []
        for v, (f_, cams_, D_, start_, end_, _rest) in enumerate(gru_batch):
            interval_ = float((np.float32(end_) - np.float32(start_)) / (np.float32(D_) - np.float32(1)))      # model.py:606-607
            plan.set_cameras(cams_, start_, interval_, end_, config.inverse_depth, view=v)
            dvs.append(wta_depth_values(D_, start_, end_, config.inverse_depth))
        m0 = mark()
        dd, pp_ = plan.run_gru_batch([g_[0] for g_ in gru_batch], dvs)
        m1 = mark()
        for v, g_ in enumerate(gru_batch):
            # the tower passes (and uploads) of the groups that fed this sweep: their event pairs ride on its first view
            marks = [None, None, m0, m1] if (timings is not None and v == 0) else None
            if marks is not None and tower_marks:
                marks[0], marks[1] = tower_marks.pop(0)
                ev_marks.extend([[a_, b_, m0, m0] for a_, b_ in tower_marks])      # further groups: towers only (zero-length hot path)
                tower_marks.clear()
            finish(dd[v], pp_[v], *g_[5], marks)        # copied out in stream order, before the next sweep overwrites the plan's buffers
            if marks is not None:
                ev_marks.append(marks)
        gru_batch.clear()

    plan_cache = {}
    it = iter(mine)
    for _ in range(ahead):
        submit_next(it)
    warm_device_one_offs(device, centre=not towers_take_u8)
    chunk = 8                                         # reference views per tower pass (their new images form one batch)
    # (round 6 measured a ramp of 2, 4, 8 views for the first groups: no gain -- the worker processes decode a group's images in
    #  parallel, so the first group of eight is ready as soon as a group of two would be: 555-570 against 567-583 depth maps/s)
    while pending:
        group = []
        t0 = time.perf_counter()
        while pending and len(group) < chunk:
            c, fut = pending.pop(0)
            submit_next(it)
            try:
                group.append((c, fut.result()))
            except Exception as e:                    # skip-and-log per reference view
                logger.warning("skipping cluster %s/%d: %s", c.session_dir, c.ref_index, e)
        t1 = time.perf_counter()
        tm["wait_load"] += t1 - t0
        if not group:
            continue
        if t_stream is not None:
            with torch.cuda.stream(t_stream):
                m_start = mark()
                prefetch_features(group)
                m_towers = mark()
                towers_done = t_stream.record_event()
            torch.cuda.current_stream(device).wait_event(towers_done)
        else:
            m_start = mark()
            prefetch_features(group)
            m_towers = mark()
        tm["submit_towers"] += time.perf_counter() - t1
        # the group's cameras in ONE upload through pinned staging (a pageable .to(device) per reference view is a blocking
        # copy queued behind the previous view's kernels: it cost this thread ~0.6 ms per view)
        cams_group = to_device([np.asarray(res_[2], dtype=np.float32) for _c, res_ in group])
        first = True
        for gi, (c, (out_images, in_images, out_cams, full_cams, index)) in enumerate(group):
            start = time.time()
            t_a = time.perf_counter()
            features = features_of(c, in_images)
            tm["submit_features"] += time.perf_counter() - t_a
            cams = cams_group[gi]
            depth_start = float(out_cams[0, 1, 3, 0])     # predictlib.set_shapes :190-197
            depth_interval = float(out_cams[0, 1, 3, 1])
            depth_num = int(out_cams[0, 1, 3, 2])
            depth_end = float(out_cams[0, 1, 3, 3])
            rest = (out_images, in_images, out_cams, full_cams, index)
            batched_gru = (config.regularization == "GRU" and gru_views > 1 and not config.refinement and
                           (not gru_batch or (gru_batch[0][0].shape == features.shape and gru_batch[0][2] == depth_num)))
            if config.regularization == "GRU" and gru_views > 1 and not config.refinement and not batched_gru:
                flush_gru()                               # a view of another size starts a new batch
                batched_gru = True
            if batched_gru:
                if first and timings is not None:
                    tower_marks.append((m_start, m_towers))
                gru_batch.append((features, cams, depth_num, depth_start, depth_end, rest))
                if len(gru_batch) >= gru_views:
                    flush_gru()
            else:
                ref_image = images_to_device(in_images[0:1]) if config.refinement else None
                m_a = mark()
                t_b = time.perf_counter()
                d, p, _ = pl.get_depth_and_prob_map(None, cams[None], depth_start, depth_interval, config, weights,
                                                    depth_num=depth_num, depth_end=depth_end, features=features,
                                                    ref_image=ref_image)
                m_depth = mark()
                t_c = time.perf_counter()
                tm["submit_depth"] += t_c - t_b
                marks = [m_start if first else None, m_towers if first else None, m_a, m_depth] if timings is not None else None
                finish(d, p, *rest, marks)
                tm["submit_finish"] += time.perf_counter() - t_c
                if marks is not None:
                    ev_marks.append(marks)
            first = False
            done += 1
            logger.info("Depth inference %d/%d finished. (%.3f sec/step)", done, len(mine), time.time() - start)
        with tm_lock:
            tm["host_gpu_submit"] += time.perf_counter() - t1
    flush_gru()
    for w_ in writes:
        w_.result()                                   # surfaces write errors; all files are on disk on return
    for f_ in pool_writes:
        tm["write"] += f_.result()                    # worker seconds; raises what the worker raised
    if session_loader is not None:
        tm["load"] += session_loader.load_seconds
    loader.shutdown(); writer.shutdown()
    if timings is not None:
        torch.cuda.synchronize()
        timings.update(tm)
        timings["wall"] = time.perf_counter() - t_wall
        timings["depth_maps"] = done
        timings["loader_threads"], timings["writer_threads"] = n_loaders, n_writers
        timings["host_workers"] = pool.workers if pool is not None else 0
        if cpu0 is not None:
            import psutil
            timings["host_cpu"] = (sum(psutil.Process().cpu_times()[:2]) - cpu0[0]) + \
                                  ((pool.cpu_seconds() or 0.0) - (cpu0[1] or 0.0) if pool is not None else 0.0)
        tw = th = td = 0.0
        for m in ev_marks:
            if m[0] is not None and m[1] is not None:
                tw += m[0].elapsed_time(m[1]) * 1e-3
            th += m[2].elapsed_time(m[3]) * 1e-3
            if len(m) > 4:
                td += m[3].elapsed_time(m[4]) * 1e-3
        timings["towers"], timings["hot_path"], timings["d2h"] = tw, th, td
    return done


def main(argv=None):
    from . import ensure_miopen_workaround
    ensure_miopen_workaround("mvsnet_amd.inference")      # the torch extractor / refinement towers run ATen convolutions
    from . import predictlib as pl
    from . import shard as sh
    ap = argparse.ArgumentParser(description=__doc__)
    cfg = pl.InferenceConfig()
    for name, default in vars(cfg).items():
        if isinstance(default, bool):
            ap.add_argument("--" + name, type=lambda s: s.lower() in ("1", "true", "yes"), default=default)
        else:
            ap.add_argument("--" + name, type=type(default) if default is not None else str, default=default)
    ap.add_argument("--weights", default=None, help=".npz with 'unet', 'regnet', 'gru' parameter dicts")
    ap.add_argument("--model_dir", default=None, help="reference checkpoint root (TensorFlow V2 checkpoint, read without TF)")
    ap.add_argument("--ckpt_step", type=int, default=400000)
    ap.add_argument("--extractor", choices=("hip", "torch"), default="hip",
                    help="2D feature towers: HIP library kernels (default) or the PyTorch/MIOpen module")
    ap.add_argument("--gru_views", type=int, default=4, help="reference views per recurrent sweep (GRU regulariser)")
    ap.add_argument("--gpus", type=int, default=1,
                    help="N > 1 without a torch.distributed.run environment: start N one-GPU ranks of this command "
                         "(reference views sharded round-robin, no data-path collective)")
    ap.add_argument("--procs_per_gpu", type=int, default=1,
                    help="worker processes sharing each GPU (ranks = gpus x procs_per_gpu, collectives over gloo): the loop "
                         "load -> towers -> hot path -> write is bound by ONE Python thread per process (DESIGN section 5, "
                         "`session`), several processes per GPU fill it")
    ap.add_argument("--host_workers", type=int, default=-1,
                    help="worker processes for image decoding / rescaling and output encoding (mvsnet_amd/host_pool.py); "
                         "-1 = by core count, 0 = loader / writer threads inside this process")
    ap.add_argument("--passes", type=int, default=1,
                    help="run the whole input this many times and report all passes after the first together (throughput "
                         "measurements: the first pass pays plans, code objects and pinned buffers)")
    args = ap.parse_args(argv)
    if not (1 <= args.gru_views <= 8):                         # MVS_GRU_MAX_VIEWS of the library (mvs_gru_wta_batch_f32)
        ap.error("--gru_views must be 1..8 (reference views per recurrent sweep)")
    logging.basicConfig(level=os.environ.get("LOG_LEVEL", "INFO"))
    if args.procs_per_gpu > 1 and args.regularization == "GRU":
        # measured (round 3, 160x128, D = 192): 165 depth maps/s with one process, 53 with three -- the recurrent sweep is a
        # four-queue wavefront, and the queues of several processes share the GPU's four compute pipes
        logger.warning("--procs_per_gpu %d with the GRU regulariser: the sweeps of different processes share the GPU's compute "
                       "pipes and slow each other down; use --gru_views to put several reference views into one sweep instead",
                       args.procs_per_gpu)
    n_ranks = max(1, args.gpus) * max(1, args.procs_per_gpu)
    if n_ranks > 1 and "WORLD_SIZE" not in os.environ:         # before anything touches the GPU: the parent stays GPU-less
        import sys
        if args.procs_per_gpu > 1:                             # ranks share GPUs: RCCL wants one GPU per rank
            os.environ["MVS_DIST_BACKEND"] = "gloo"
        raise SystemExit(sh.launch_ranks(list(sys.argv[1:] if argv is None else argv), n_ranks, module="mvsnet_amd.inference",
                                         gpus=max(1, args.gpus)))
    weights_path = args.weights
    for name in vars(cfg):
        setattr(cfg, name, getattr(args, name))
    if cfg.input_dir is None:
        ap.error("--input_dir is required")
    dist = sh.init_process_group()
    # a single session, or a folder of sessions (inference.py:121-141)
    if os.path.isfile(os.path.join(cfg.input_dir, "covisibility.json")) or \
            os.path.isfile(os.path.join(cfg.input_dir, "pair.txt")):
        dirs = [cfg.input_dir]
    else:
        dirs = [os.path.join(cfg.input_dir, f) for f in sorted(os.listdir(cfg.input_dir))
                if not f.startswith(".") and not f.endswith(".txt")]
    import torch
    rank, local_rank, world = sh.rank_world()
    device = sh.bind_device(local_rank)
    weights = build_weights(cfg, device, weights_path, args.model_dir, args.ckpt_step, args.extractor)
    total, wall, t0 = 0, 0.0, time.perf_counter()
    for p_ in range(max(1, args.passes)):
        if p_ <= 1:                                   # the timed region = every pass after the first (or the only one)
            if dist is not None:
                dist.barrier()                        # the ranks start together: the rate is all maps / the slowest rank
            t0, total = time.perf_counter(), 0
        for d in dirs:
            total += compute_depth_maps(d, cfg, weights, device, gru_views=args.gru_views,
                                        host_workers=None if args.host_workers < 0 else args.host_workers)
        wall = time.perf_counter() - t0
    counts = sh.gather_counts(dist, total, device=device if dist is not None else "cpu")
    walls = sh.gather_counts(dist, wall, device=device if dist is not None else "cpu")
    devs = sh.gather_counts(dist, device.index, device=device if dist is not None else "cpu")      # the GPU each rank was bound to
    if rank == 0:
        import json
        logger.info("all dense finished: %d depth maps (%s per rank)", int(sum(counts)), counts)
        print(json.dumps({"depth_maps": int(sum(counts)), "ranks": len(counts), "procs_per_gpu": args.procs_per_gpu,
                          "devices": [int(x) for x in devs],
                          "seconds_slowest_rank": max(walls), "depth_maps_per_s": sum(counts) / max(max(walls), 1e-9),
                          "sec_per_step": max(walls) / max(sum(counts), 1), "passes": max(1, args.passes)}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
