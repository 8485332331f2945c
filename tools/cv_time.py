"""Warp + variance cost volume alone at a BASELINE workload, timed with HIP events (and runnable under rocprofv3):
    python tools/cv_time.py [M|c1|c2] [--iters 20] [--planes D]
(the LDS-staged and MFMA-blend variants are lab kernels now: tests/test_gpu_lab.py, python -m mvsnet_amd.build --lab)
MVS_LIB_PATH=<another build>."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S                     # noqa: E402
from mvsnet_amd.model import cost_volume, homography_transforms   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="M")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--planes", type=int, default=0)
a = ap.parse_args()
w = S.make_workload(a.workload)
dev = "cuda"
feats = torch.as_tensor(w.features, device=dev)
cams = torch.as_tensor(w.cams, device=dev)
D = a.planes or w.depth_num
T = homography_transforms(cams, w.depth_num, w.depth_start, w.depth_interval)[:, :D].contiguous()
out = torch.empty((D, w.height, w.width, w.channels), device=dev)
for _ in range(3):
    cost_volume(feats[0], feats[1:], T, out=out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
ev[0].record()
for i in range(a.iters):
    cost_volume(feats[0], feats[1:], T, out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(a.iters))
nbytes = out.numel() * 4 + feats.numel() * 4
print("%s D=%d: median %.1f us  min %.1f  max %.1f   (%.0f GB/s algorithmic)  checksum %.6e"
      % (a.workload, D, ts[len(ts) // 2], ts[0], ts[-1], nbytes / ts[len(ts) // 2] / 1e3, float(out.double().sum())))
