import time, torch, sys
sys.path.insert(0, '.')
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
w = S.make_workload("c3")
import os
gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=bool(os.environ.get("RA")))
gw = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
feats = torch.as_tensor(w.features).to(dev); cams = torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev)
def run(stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
            plan.run_gru(feats, dv)
for name, st in (("default stream", torch.cuda.default_stream()), ("side stream", torch.cuda.Stream()), ("default again", torch.cuda.default_stream())):
    run(st, 2); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(st, 5); torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 5 * 1e3, "ms per depth map", flush=True)
