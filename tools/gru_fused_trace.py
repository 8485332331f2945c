"""Device-side time line of the fused ConvGRU sweep (mvs_gru_fused_trace): where a launch's time goes.
    python tools/gru_fused_trace.py [--views 1]
Per phase (G = gates launch, C = output launch), over the steady-state launches of one sweep, in microseconds:
  gap      previous launch's last workgroup exit -> this launch's first workgroup entry
  ramp     first entry -> median entry of the launch's workgroups
  prologue entry -> first tile staged (weights in LDS, LayerNorm affines, first tile's loads + staging)
  tile     first tile staged -> first tile done ; per-tile average over the rest of the loop
  tail     loop done -> exit (sum reduction, float64 atomics, the stamp itself)
  span     first entry -> last exit of the launch"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S                              # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
w = S.make_workload("c3")
gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
gw = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
cams = torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
B = a.views
feats = [torch.as_tensor(S.make_features(w.view_num, w.height, w.width, w.channels, seed=v)).to(dev) for v in range(B)]
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev, views=B)
for v in range(B):
    plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False, view=v)
plan.run_gru_batch(feats, [dv] * B)
torch.cuda.synchronize()
cap = 2 * (w.depth_num + 3) * 256 + 16
buf = torch.zeros(1 + 8 * cap, dtype=torch.int64, device=dev)
lib = _lib.load()
_lib.check(lib.mvs_gru_fused_trace(_lib.ptr(buf), cap), "trace")
plan.run_gru_batch(feats, [dv] * B)
torch.cuda.synchronize()
_lib.check(lib.mvs_gru_fused_trace(None, 0), "trace")
h = buf.cpu().numpy()
n = int(h[0]); rec = h[1:1 + 8 * min(n, cap)].reshape(-1, 8)
launch = rec[:, 0] >> 32; phase = (rec[:, 0] >> 16) & 0xffff
ids = np.unique(launch)
per = {}
for L in ids:
    r = rec[launch == L]
    per[L] = dict(phase=int(r[0, 0] >> 16 & 0xffff), first=r[:, 1].min(), last=r[:, 5].max(), r=r)
us = lambda t: t / 100.0
for ph, name in ((0, "G"), (1, "C")):
    rows = []
    for L in ids:
        if per[L]["phase"] != ph or L < 20 or L > ids.max() - 20 or (L - 1) not in per:
            continue
        p, q = per[L], per[L - 1]
        r = p["r"]
        tiles = np.maximum(r[:, 6] & 0xffff, 1)
        pa, pb, pc = (r[:, 6] >> 16) & 0xffff, (r[:, 6] >> 32) & 0xffff, (r[:, 6] >> 48) & 0xffff
        rest = np.where(tiles > 1, (r[:, 4] - r[:, 3]) / np.maximum(tiles - 1, 1), np.nan)
        rows.append([us(p["first"] - q["last"]), us(np.median(r[:, 1]) - p["first"]), us(np.median(r[:, 2] - r[:, 1])),
                     us(np.median(r[:, 3] - r[:, 2])), us(np.nanmedian(rest)), us(np.median(r[:, 5] - r[:, 4])),
                     us(p["last"] - p["first"]), tiles.max(), us(np.median(r[:, 5]) - p["first"]), us(np.percentile(r[:, 1], 95) - p["first"]),
                     us(np.median(pa)), us(np.median(pb)), us(np.median(pc)), np.median(r[:, 7] / np.maximum(r[:, 5] - r[:, 1], 1)) * 100.0])
    m = np.median(np.array(rows), axis=0)
    print("%s launch, %d view(s): gap %.1f | ramp (median entry) %.1f, 95%% entry %.1f | prologue %.1f | first tile %.1f | later tiles %.2f each (max %d tiles) | tail %.1f | "
          "median exit at %.1f | span %.1f us\n      prologue, since entry: first barrier passed (affines) %.1f, first tile staged %.1f, weights in LDS + second tile requested %.1f; shader clock during the launch %.0f MHz" % (name, B, m[0], m[1], m[9], m[2], m[3], m[4], int(m[7]), m[5], m[8], m[6], m[10], m[11], m[12], m[13]))
