"""The recurrent sweep (ConvGRU + winner-take-all) alone at workload c3 (N=5, D=256, 400x300): ms per depth map.
    python tools/gru_time.py [--iters 5]      MVS_LIB_PATH=<another build> for A/B of compile-time settings."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S                                   # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = S.make_workload("c3")
gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
gw = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev)


def run(n):
    for _ in range(n):
        plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
        plan.run_gru(feats, dv)


run(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.iters)
torch.cuda.synchronize()
print("c3 sweep: %.2f ms per depth map (lib %s)" % ((time.perf_counter() - t0) / a.iters * 1e3, os.environ.get("MVS_LIB_PATH", "default")), flush=True)
