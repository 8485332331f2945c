"""Hot path forward+backward only (features -> depth -> gradients), for profiling: no torch towers."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S, train as T, backward as B
from mvsnet_amd.homography_warping import homography_transforms

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=3); ap.add_argument("--depth", type=int, default=192)
ap.add_argument("--height", type=int, default=120); ap.add_argument("--width", type=int, default=160)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
N, D, H, W = a.views, a.depth, a.height, a.width
cams = S.make_cams(N, H, W, D)
start, interval = float(cams[0, 1, 3, 0]), float(cams[0, 1, 3, 1])
params = S.make_regnet_params("normal", seed=1)
p = {k: {kk: torch.as_tensor(vv).cuda().requires_grad_(True) for kk, vv in v.items()} for k, v in params.items()}
feats = torch.as_tensor(S.make_features(N, H, W, 32)).cuda().requires_grad_(True)
t8 = homography_transforms(torch.as_tensor(cams).cuda(), D, start, interval)
g = torch.ones(H, W, device="cuda")
def hot():
    d, _ = B.plane_sweep_depth(feats, t8, start, interval, p)
    d.backward(g)
for _ in range(2): hot()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(a.iters): hot()
torch.cuda.synchronize()
print({"hot_path_fwd_bwd_ms": round((time.time() - t0) / a.iters * 1e3, 3), "config": vars(a)})
