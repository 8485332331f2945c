"""Benchmark driver mirroring `python -m mvsnet.test` (mvsnet/test.py:92-200): depth inference on
sessions that ship ground-truth depth maps (`depths/<i>.png`, uint16 mm), scored with the
reference's loss and less-one / less-three accuracies, averaged and appended to a results CSV.

    python -m mvsnet_amd.test --input_dir <session or folder of sessions> [--results_path results.csv]
        [--view_num 5 --max_d 192 --width 640 --height 512 ...same flags as mvsnet_amd.inference]
"""
from __future__ import annotations

import argparse
import logging
import os

import numpy as np

logger = logging.getLogger("mvsnet_amd.test")
RESULTS_HEADER = "model_dir, ckpt_step, loss, less_one, less_three, debug \n"       # predictlib.py:226-228


def write_results(path, model_dir, ckpt_step, loss, less_one, less_three, debug):
    """predictlib.py:231-266: header written once, one line appended per run."""
    try:
        lines = open(path).readlines()
    except Exception:
        lines = []
    with open(path, "a+") as f:
        if not lines or lines[0] != RESULTS_HEADER:
            f.write(RESULTS_HEADER)
        f.write("{}, {}, {}, {}, {}, {} \n".format(model_dir, ckpt_step, loss, less_one, less_three, debug))


def benchmark_depth_maps(input_dir, config, weights, device, losses, less_ones, less_threes, out_debugs,
                         loss_type="original", grad_loss=True, write_output=False):
    """test.py:92-169 for one session directory; appends per-cluster metrics to the four lists."""
    import torch
    from . import predictlib as pl
    from .loss import mvsnet_regression_loss
    from .mvs_data_generation import make_generator
    from .refine import resize_bilinear_tf1

    gen = make_generator(input_dir, config.view_num, config.width, config.height, config.max_d,
                         config.interval_scale, config.base_image_size, mode="test",
                         output_scale=config.sample_scale,
                         max_clusters_per_session=config.max_clusters_per_session)
    done = 0
    # the next clusters are decoded / resized on loader threads while this one is on the GPU (the reference feeds one cluster per
    # sess.run, test.py:121-135; preparing five images takes 20-50 x as long as their depth map here)
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    loaders, ahead = ThreadPoolExecutor(max_workers=4), deque()
    todo = iter(gen.clusters)
    for c in todo:
        ahead.append((c, loaders.submit(gen.prepare, c)))
        if len(ahead) >= 6:
            break
    while ahead:
        c, fut = ahead.popleft()
        for nxt in todo:
            ahead.append((nxt, loaders.submit(gen.prepare, nxt)))
            break
        try:
            out_images, in_images, out_cams, full_cams, index, full_depth = fut.result()
        except Exception as e:
            logger.warning("skipping cluster %s/%d: %s", c.session_dir, c.ref_index, e)
            continue
        images = torch.as_tensor(in_images, dtype=torch.float32, device=device)[None]
        cams = torch.as_tensor(out_cams, dtype=torch.float32, device=device)[None]
        depth_start, depth_interval = float(out_cams[0, 1, 3, 0]), float(out_cams[0, 1, 3, 1])
        depth_num, depth_end = int(out_cams[0, 1, 3, 2]), float(out_cams[0, 1, 3, 3])
        d, p, _ = pl.get_depth_and_prob_map(images, cams, depth_start, depth_interval, config, weights,
                                            depth_num=depth_num, depth_end=depth_end)
        gt = torch.as_tensor(full_depth, dtype=torch.float32, device=device)[None]
        if not (config.refinement and config.upsample_before_refinement):              # test.py:105-108
            d = resize_bilinear_tf1(d, gt.shape[1], gt.shape[2])
        loss, less_one, less_three, debug = mvsnet_regression_loss(d, gt, [depth_start], [depth_end],
                                                                   loss_type=loss_type, grad_loss=grad_loss)
        losses.append(float(loss)); less_ones.append(float(less_one)); less_threes.append(float(less_three))
        out_debugs.append(float(debug) if debug is not None else 0.0)
        logger.info("Image %d loss = %s, less one = %s, less three = %s", index, losses[-1], less_ones[-1], less_threes[-1])
        if write_output:
            out_dir = pl.setup_output_dir(input_dir, config.output_dir)
            pl.write_output_slice(out_dir, d.cpu().numpy(), p.cpu().numpy(), out_images[0], out_cams[0], index)
        done += 1
    loaders.shutdown(wait=False)
    return done


def main(argv=None):
    from . import ensure_miopen_workaround
    ensure_miopen_workaround("mvsnet_amd.test")
    import torch
    from . import predictlib as pl, shard as sh
    from .inference import build_weights
    ap = argparse.ArgumentParser(description=__doc__)
    cfg = pl.InferenceConfig(max_clusters_per_session=100)
    for name, default in vars(cfg).items():
        if isinstance(default, bool):
            ap.add_argument("--" + name, type=lambda s: s.lower() in ("1", "true", "yes"), default=default)
        else:
            ap.add_argument("--" + name, type=type(default) if default is not None else str, default=default)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--model_dir", default=None)
    ap.add_argument("--ckpt_step", type=int, default=400000)
    ap.add_argument("--extractor", choices=("hip", "torch"), default="hip")
    ap.add_argument("--results_path", default="./results.csv")
    ap.add_argument("--loss_type", default="original")
    ap.add_argument("--grad_loss", type=lambda s: s.lower() in ("1", "true", "yes"), default=True)
    ap.add_argument("--write_output", type=lambda s: s.lower() in ("1", "true", "yes"), default=False)
    args = ap.parse_args(argv)
    logging.basicConfig(level=os.environ.get("LOG_LEVEL", "INFO"))
    for name in vars(cfg):
        setattr(cfg, name, getattr(args, name))
    if cfg.input_dir is None:
        ap.error("--input_dir is required")
    rank, local_rank, _world = sh.rank_world()
    device = sh.bind_device(local_rank)
    weights = build_weights(cfg, device, args.weights, args.model_dir, args.ckpt_step, args.extractor)
    dirs = [cfg.input_dir] if os.path.isfile(os.path.join(cfg.input_dir, "covisibility.json")) else \
        [os.path.join(cfg.input_dir, f) for f in sorted(os.listdir(cfg.input_dir)) if not f.startswith(".")]
    losses, less_ones, less_threes, debugs = [], [], [], []
    for d in dirs:
        benchmark_depth_maps(d, cfg, weights, device, losses, less_ones, less_threes, debugs,
                             args.loss_type, args.grad_loss, args.write_output)
    if not losses:
        raise SystemExit("no cluster with ground-truth depth found under %s" % cfg.input_dir)
    avg = [float(np.mean(v)) for v in (losses, less_ones, less_threes, debugs)]
    logger.info(" ** Average Loss = %s", avg[0]); logger.info(" ** Average Less one = %s", avg[1])
    logger.info(" ** Average Less three = %s", avg[2]); logger.info(" ** Average debug = %s", avg[3])
    write_results(args.results_path, args.model_dir, args.ckpt_step, *avg)
    return avg


if __name__ == "__main__":
    main()
