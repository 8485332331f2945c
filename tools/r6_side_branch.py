"""Round 6 experiment: RegNetUS0's 3dconv1_1 on a side stream of the caller's stream set, beside 3dconv2_0 and the low-resolution
chain (MVS_HOOK_REGNET_SIDE_BRANCH), against the default (fused 3dconv1_1 + 2_0 in line).  Workload M, 200-step repetitions."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "M"
w = S.make_workload(name)
weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1, random_affine=True), device=dev)
feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
end = w.depth_start + (w.depth_num - 1) * w.depth_interval
run = lambda: plan.run_depth(feats, cams, w.depth_start, w.depth_interval, end, False)
for _ in range(50): run()
torch.cuda.synchronize()
ref = plan.depth.clone()
tok = _lib.gru_prepare()
print("stream set prepared:", tok is not None)
for side in (0, 1, 0, 1):
    _lib.set_test_hook("regnet_side_branch", side)
    for _ in range(30): run()
    torch.cuda.synchronize()
    rates = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): run()
        torch.cuda.synchronize()
        rates.append(200 / (time.perf_counter() - t0))
    d = plan.depth
    print("side branch %d: %s depth maps/s; max |depth - default| / depth = %.2e" % (
        side, " ".join("%.1f" % r for r in rates), float(((d - ref).abs() / ref).max())), flush=True)
_lib.set_test_hook("regnet_side_branch", 0)
_lib.gru_unref(tok)
