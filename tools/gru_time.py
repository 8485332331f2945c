"""The recurrent sweep (ConvGRU + winner-take-all) alone at workload c3 (N=5, D=256, 400x300): ms per depth map.
    python tools/gru_time.py [--iters 5] [--views 1 2 4]      MVS_LIB_PATH=<another build> for A/B of compile-time settings.
--views B: B independent reference views per sweep (mvs_gru_wta_batch_f32)."""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S                              # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--views", type=int, nargs="+", default=[1])
ap.add_argument("--form", type=int, default=0, help="0 / 3 fused two-launch sweep (default), 1 wavefront with hoisted x-part, 2 wavefront with full 48-channel kernels")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_lib.check(_lib.load().mvs_gru_set_formulation(a.form), "mvs_gru_set_formulation")
w = S.make_workload("c3")
gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
gw = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
cams = torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
for B in a.views:
    feats = [torch.as_tensor(S.make_features(w.view_num, w.height, w.width, w.channels, seed=v)).to(dev) for v in range(B)]
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev, views=B)

    def run(n):
        for _ in range(n):
            for v in range(B):
                plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False, view=v)
            plan.run_gru_batch(feats, [dv] * B)

    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.iters)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("form %d: c3 sweep, %d view(s) per launch: %.2f ms per sweep = %.2f ms per depth map = %.1f depth maps/s (host enqueue %.2f ms per sweep; lib %s)"
          % (a.form, B, el / a.iters * 1e3, el / a.iters / B * 1e3, a.iters * B / el, th / a.iters * 1e3, os.environ.get("MVS_LIB_PATH", "default")), flush=True)
    del plan, feats
    torch.cuda.empty_cache()
pc, us = ctypes.c_int(-9), (ctypes.c_float * 8)()
_lib.check(_lib.load().mvs_gru_stream_layout(_lib.stream_ptr(), ctypes.byref(pc), us), "mvs_gru_stream_layout")
print("stream layout: caller on candidate pipe %d; calibration chains (us) high %s low %s" % (
    pc.value, ["%.0f" % x for x in us[:4]], ["%.0f" % x for x in us[4:]]), flush=True)
