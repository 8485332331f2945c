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


def compute_depth_maps(input_dir, config=None, weights=None, device=None, **kwargs):
    """mvsnet/inference.py:83-119.  Returns the number of depth maps this rank wrote."""
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
    # Per-image feature cache: the reference re-runs the UNetDS2GN tower on
    # every source image of every cluster (model.py:392-406); an image's features only depend on
    # the image and on the (rescale, crop) it received, so they are computed once per session and
    # re-used by all the reference views that list the image as a source.
    feature_cache = {}
    done = 0
    # Host pipeline: like the reference, whose generator runs in a tf.data thread with a prefetch buffer
    # (predictlib.py:48-51), image loading / resizing of the next clusters and the file writes of the previous ones run
    # on worker threads (numpy / PIL / file IO release the GIL); this thread only drives the GPU.
    from concurrent.futures import ThreadPoolExecutor
    loader, writer = ThreadPoolExecutor(max_workers=2), ThreadPoolExecutor(max_workers=2)
    pending, writes = [], []
    ahead = 3

    def submit_next(it):
        for c_ in it:
            pending.append((c_, loader.submit(gen.prepare, c_)))
            return

    it = iter(mine)
    for _ in range(ahead):
        submit_next(it)
    while pending:
        c, fut = pending.pop(0)
        submit_next(it)
        start = time.time()
        try:
            out_images, in_images, out_cams, full_cams, index = fut.result()
        except Exception as e:                        # skip-and-log per reference view
            logger.warning("skipping cluster %s/%d: %s", c.session_dir, c.ref_index, e)
            continue
        ids = getattr(c, "indices", None) or [p for p in getattr(c, "paths", [])[0::2]]
        feats = []
        for v in range(config.view_num):
            key = (ids[v], round(float(c.rescale), 9), in_images[v].shape) if v < len(ids) else None
            f = feature_cache.get(key) if key is not None else None
            if f is None:
                f = weights.unet(torch.as_tensor(in_images[v:v + 1], dtype=torch.float32, device=device))[0]
                if key is not None:
                    if len(feature_cache) >= 256:
                        feature_cache.pop(next(iter(feature_cache)))
                    feature_cache[key] = f
            feats.append(f)
        features = torch.stack(feats).contiguous()
        cams = torch.as_tensor(out_cams, dtype=torch.float32, device=device)[None]
        depth_start = float(out_cams[0, 1, 3, 0])     # predictlib.set_shapes :190-197
        depth_interval = float(out_cams[0, 1, 3, 1])
        depth_num = int(out_cams[0, 1, 3, 2])
        depth_end = float(out_cams[0, 1, 3, 3])
        ref_image = torch.as_tensor(in_images[0:1], dtype=torch.float32, device=device) if config.refinement else None
        d, p, _ = pl.get_depth_and_prob_map(None, cams, depth_start, depth_interval, config, weights,
                                            depth_num=depth_num, depth_end=depth_end, features=features,
                                            ref_image=ref_image)
        d_np, p_np = d.cpu().numpy(), p.cpu().numpy()
        if config.refinement and config.upsample_before_refinement:      # full-size outputs (predictlib.py:107-115)
            writes.append(writer.submit(pl.write_output_slice, output_dir, d_np, p_np, in_images[0], full_cams[0],
                                        index, config.visualize, 1.0 / config.sample_scale))
        else:
            writes.append(writer.submit(pl.write_output_slice, output_dir, d_np, p_np, out_images[0], out_cams[0],
                                        index, config.visualize))
        done += 1
        logger.info("Depth inference %d/%d finished. (%.3f sec/step)", done, len(mine), time.time() - start)
    for w_ in writes:
        w_.result()                                   # surfaces write errors; all files are on disk on return
    loader.shutdown(); writer.shutdown()
    return done


def main(argv=None):
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
    ap.add_argument("--gpus", type=int, default=1,
                    help="N > 1 without a torch.distributed.run environment: start N one-GPU ranks of this command "
                         "(reference views sharded round-robin, no data-path collective)")
    args = ap.parse_args(argv)
    logging.basicConfig(level=os.environ.get("LOG_LEVEL", "INFO"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # before anything touches the GPU: the parent stays GPU-less
        import sys
        raise SystemExit(sh.launch_ranks(list(sys.argv[1:] if argv is None else argv), args.gpus, module="mvsnet_amd.inference"))
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
    total = 0
    for d in dirs:
        total += compute_depth_maps(d, cfg, weights, device)
    counts = sh.gather_counts(dist, total, device=device if dist is not None else "cpu")
    if rank == 0:
        logger.info("all dense finished: %d depth maps (%s per rank)", int(sum(counts)), counts)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
