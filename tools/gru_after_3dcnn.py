"""Why does the recurrent sweep take 44 ms after the 3D-CNN path ran in the same process, and 23 ms alone?  (round-2 bisect)"""
import os, sys, time, torch
sys.path.insert(0, '.')
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mode = sys.argv[1]

def gru_ms():
    w = S.make_workload("c3")
    gw = MVSNetWeights.from_numpy("normal", gru=S.make_gru_params("normal", seed=2, in_channels=w.channels), device=dev)
    feats = torch.as_tensor(w.features).to(dev); cams = torch.as_tensor(w.cams).to(dev)
    dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev)
    def run(n):
        for _ in range(n):
            plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
            plan.run_gru(feats, dv)
    run(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(5); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3

def cnn(name, steps):
    w = S.make_workload(name)
    weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1), device=dev)
    feats = torch.as_tensor(w.features).to(dev); cams = torch.as_tensor(w.cams).to(dev)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
    for _ in range(steps):
        plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
        plan.run_3dcnn(feats, w.depth_start, w.depth_interval)
    torch.cuda.synchronize()

if mode == "cnn_first":
    cnn("M", 20)
elif mode == "streams_first":
    ss = [torch.cuda.Stream() for _ in range(4)]
    for s in ss:
        with torch.cuda.stream(s):
            torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
elif mode == "c2_first":
    cnn("c2", 5)
elif mode == "events_first":
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(100)]
    for e in ev: e.record()
    torch.cuda.synchronize()
print(mode, "%.2f ms per depth map" % gru_ms(), flush=True)
