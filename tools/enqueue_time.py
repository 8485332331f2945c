"""Host time to enqueue one depth map (DepthPlan.run_depth without a sync) against the drained time per depth map: is `value` GPU-bound?"""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mvsnet_amd import synthetic as S, _lib
from mvsnet_amd.model import DepthPlan, MVSNetWeights
dev = "cuda"
w = S.make_workload("M")
for impl in ("auto", "bf16x3"):
    _lib.set_conv_impl(impl)
    weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1, random_affine=True), device=dev)
    feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
    for _ in range(5): plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
    torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n): plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host enqueue %.3f ms per depth map (no sync), total with drain %.3f ms per depth map" % (impl, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
