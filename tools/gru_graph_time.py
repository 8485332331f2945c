"""The fused ConvGRU sweep at c3 replayed from a hipGraph (torch.cuda.CUDAGraph) against eager launches: ms per sweep.
    python tools/gru_graph_time.py [--views 1 4]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S                              # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, nargs="+", default=[1, 4])
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
w = S.make_workload("c3")
gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
gw = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
cams = torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
for B in a.views:
    feats = [torch.as_tensor(S.make_features(w.view_num, w.height, w.width, w.channels, seed=v)).to(dev) for v in range(B)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev, views=B)
        for v in range(B):
            plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False, view=v)
        d0, p0 = plan.run_gru_batch(feats, [dv] * B)
        d0, p0 = d0.clone(), p0.clone()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            plan.run_gru_batch(feats, [dv] * B)
        s.synchronize()
        eager = (time.perf_counter() - t0) / a.iters
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            plan.run_gru_batch(feats, [dv] * B)
        g.replay(); s.synchronize()
        same = bool((plan.depth_v[:B] == d0).all())
        t0 = time.perf_counter()
        for _ in range(a.iters):
            g.replay()
        s.synchronize()
        graph = (time.perf_counter() - t0) / a.iters
    print("c3 sweep, %d view(s): eager %.2f ms, hipGraph replay %.2f ms per sweep (replayed depth identical to eager: %s)" % (B, eager * 1e3, graph * 1e3, same), flush=True)
    del plan, g
