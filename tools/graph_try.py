"""Does a DepthPlan step replay from a hipGraph (torch.cuda.graph capture of the ctypes launches), and what does
it buy on a launch-bound configuration?  python tools/graph_try.py [workload]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, model as M

name = sys.argv[1] if len(sys.argv) > 1 else "c1"
wl = S.make_workload(name)
weights = M.MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1), device="cuda")
feats = torch.as_tensor(wl.features).cuda(); cams = torch.as_tensor(wl.cams).cuda()
plan = M.DepthPlan(wl.view_num, wl.depth_num, wl.height, wl.width, 32, weights, "3DCNN")
def step():
    plan.set_cameras(cams, wl.depth_start, wl.depth_interval, wl.depth_end, False)
    plan.run_3dcnn(feats, wl.depth_start, wl.depth_interval)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50): step()
    torch.cuda.synchronize()
    eager = (time.time() - t0) / 50
    ref = plan.depth.clone()
    print("eager %.1f us per depth map (%.0f maps/s)" % (eager * 1e6, 1 / eager), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    gr = (time.time() - t0) / 50
    print("graph %.1f us per depth map (%.0f maps/s), same output: %s" % (gr * 1e6, 1 / gr, bool(torch.equal(ref, plan.depth))))
