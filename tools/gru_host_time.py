"""Host enqueue time vs device time of the recurrent sweep (is the sweep launch-bound?).  Measured r01:
enqueue 15.5 ms, total 24.4 ms at c3 -- device-bound.  (Capturing the multi-stream sweep into a hipGraph through
torch.cuda.graph crashed the process on ROCm 7.2; not pursued.)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, model as M

wl = S.make_workload("c3")
weights = M.MVSNetWeights.from_numpy("normal", gru=S.make_gru_params("normal"), device="cuda")
feats = torch.as_tensor(wl.features).cuda(); cams = torch.as_tensor(wl.cams).cuda()
plan = M.DepthPlan(wl.view_num, wl.depth_num, wl.height, wl.width, 32, weights, "GRU")
plan.set_cameras(cams, wl.depth_start, wl.depth_interval, wl.depth_end, False)
dv = [wl.depth_start + i * wl.depth_interval for i in range(wl.depth_num)]
for _ in range(2): plan.run_gru(feats, dv)
torch.cuda.synchronize()
t0 = time.time(); plan.run_gru(feats, dv); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
print("enqueue %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
