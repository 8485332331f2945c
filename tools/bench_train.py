"""Times one training step (images -> towers -> hot path forward + backward -> optimiser) and its parts on one
GPU: `python tools/bench_train.py [--views 3 --depth 192 --height 480 --width 640]` (train.py defaults)."""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, train as T, backward as B
from mvsnet_amd.homography_warping import homography_transforms


def timed(fn, iters):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=3); ap.add_argument("--depth", type=int, default=192)
    ap.add_argument("--height", type=int, default=480); ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--regularization", default="3DCNN")
    a = ap.parse_args()
    N, D, H, W = a.views, a.depth, a.height, a.width
    images = S.make_images(N, H, W); cams = S.make_cams(N, H // 4, W // 4, D)
    start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
    gt = np.full((H // 4, W // 4, 1), start + interval * D * 0.5, np.float32)
    if a.regularization == "GRU":
        tr = T.Trainer("normal", "cuda", regularization="GRU")
        tr.train_step(images, cams, gt, D)
        step = timed(lambda: tr.train_step(images, cams, gt, D), a.iters)
        torch.cuda.reset_peak_memory_stats()

        def fwd():
            with torch.no_grad():
                tr.loss(images, cams, gt, D)
        fwd(); hf = timed(fwd, a.iters)
        print({"gru_train_step_ms": round(step, 1), "gru_forward_ms": round(hf, 1),
               "peak_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
               "config": dict(views=N, depth=D, height=H, width=W)})
        return
    tr = T.Trainer("normal", "cuda")
    for _ in range(2):
        tr.train_step(images, cams, gt, D)
    step = timed(lambda: tr.train_step(images, cams, gt, D), a.iters)
    # hot path alone: features -> depth -> backward
    feats = torch.as_tensor(S.make_features(N, H // 4, W // 4, 32)).cuda().requires_grad_(True)
    t8 = homography_transforms(torch.as_tensor(cams).cuda(), D, start, interval)
    p = tr.params.group("regnet")
    g = torch.ones(H // 4, W // 4, device="cuda")

    def hot():
        d, _ = B.plane_sweep_depth(feats, t8, start, interval, p)
        d.backward(g)
    hot(); hp = timed(hot, a.iters)

    def fwd():
        with torch.no_grad():
            B.plane_sweep_depth(feats, t8, start, interval, p)
    fwd(); hf = timed(fwd, a.iters)
    print({"train_step_ms": round(step, 2), "hot_path_fwd_bwd_ms": round(hp, 2), "hot_path_fwd_ms": round(hf, 2),
           "config": dict(views=N, depth=D, height=H, width=W)})


if __name__ == "__main__":
    main()
