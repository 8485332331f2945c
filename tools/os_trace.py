"""Device-side phase timing of the low-resolution block kernels (conv3d_os.hip) inside a depth map at the metric workload:
variant build with the MVS_TL stamps of tools/gru_trace.patch + the same marks in os_block (see tools/README.md), then
    MVS_LIB_PATH=mvsnet_amd/variants/lib_tlos.so python tools/os_trace.py
kernel id = KIND * 1000 + Cin (0 stride 1, 1 stride 2, 2 transposed)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S                    # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights         # noqa: E402

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.load()
w = S.make_workload("M")
rp = S.make_regnet_params("normal", seed=1)
weights = MVSNetWeights.from_numpy("normal", regnet=rp, device=dev)
feats = torch.as_tensor(w.features).to(dev); cams = torch.as_tensor(w.cams).to(dev)
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
end = w.depth_start + (w.depth_num - 1) * w.depth_interval
for _ in range(5):
    plan.run_depth(feats, cams, w.depth_start, w.depth_interval, end, False)
torch.cuda.synchronize()
CAP = 1 << 20
b = torch.zeros(1 + 8 * CAP, dtype=torch.int64, device=dev)
fn = lib.mvs_tl_set_os; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(ctypes.c_void_p(b.data_ptr())) == 0
NMAPS = 10
for _ in range(NMAPS):
    plan.run_depth(feats, cams, w.depth_start, w.depth_interval, end, False)
torch.cuda.synchronize()
h = b.cpu().numpy(); n = int(h[0]); r = h[1:1 + 8 * n].reshape(n, 8)
fn(ctypes.c_void_p(0))
names = {32: "3dconv2_1 blocks (fillers)", 1032: "3dconv3_0", 64: "3dconv3_1", 2064: "3dconv4_0", 2032: "3dconv5_0"}
# launches: cluster ALL records by time (the four launches of a depth map follow each other)
x = r[np.argsort(r[:, 1])]
launches = []; s = 0; e = x[0, 2]
for i in range(1, len(x)):
    if x[i, 1] > e:
        launches.append(x[s:i]); s = i; e = x[i, 2]
    else:
        e = max(e, x[i, 2])
launches.append(x[s:])
print("%d workgroup records, %d launches in %d depth maps" % (n, len(launches), NMAPS))
per = len(launches) // NMAPS
for k in range(per):
    ls = launches[k::per][1:]                       # the k-th launch of every depth map but the first
    kinds = sorted(set(int(v) for v in ls[0][:, 0]))
    dur = np.mean([(l[:, 2].max() - l[:, 1].min()) / 100.0 for l in ls])
    gap = np.mean([(launches[k + per * (m + 1)][:, 1].min() - launches[k - 1 + per * (m + 1)][:, 2].max()) / 100.0 for m in range(len(ls))]) if k else float("nan")
    print("\nlaunch %d: %s  | %d workgroups, first start -> last end %.1f us, idle before it %.1f us"
          % (k, " + ".join(names.get(v, str(v)) for v in kinds), len(ls[0]), dur, gap))
    for kid in kinds:
        m = np.concatenate([l[l[:, 0] == kid] for l in ls])
        t0 = np.concatenate([np.full((l[:, 0] == kid).sum(), l[:, 1].min()) for l in ls])
        ph = np.stack([m[:, 1] - t0, m[:, 3] - m[:, 1], m[:, 4] - m[:, 3], m[:, 5] - m[:, 4], m[:, 6] - m[:, 5], m[:, 2] - m[:, 6]], 1) / 100.0
        print("   %-28s start after launch begin %5.1f (90 %%: %5.1f) | loads issued %4.1f | staged + barrier %4.1f | K loop %5.1f | stores %4.1f | sums + atomics %4.1f | lifetime %5.1f us"
              % ((names.get(kid, str(kid)), ph[:, 0].mean(), np.percentile(ph[:, 0], 90)) + tuple(ph[:, 1:].mean(0)) + ((m[:, 2] - m[:, 1]).mean() / 100.0,)))
