"""Phase split of the block kernels (conv3d_os.hip) inside a metric-size depth map, from s_memtime stamps of a DIAGNOSTIC build
(`git apply tools/os_prof.patch`, rebuild; the product source carries no stamps).  Ticks are s_memtime at its constant rate:
round 2 read 3dconv3_1 as load-issue 3.1 k, sums + float64 affine 3.9 k, LDS write + barrier 2.0 k, K loop 33 k (= 864 MFMAs at the
issue rate of the ~2.0 GHz the chip holds under MFMA load), stores 0.8 k, BatchNorm sums 8.7 k ticks."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights
lib = _lib.load()
lib.mvs_os_prof_dump.argtypes = [ctypes.c_void_p]
w = S.make_workload("M")
weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1), device="cuda")
feats, cams = torch.as_tensor(w.features).cuda(), torch.as_tensor(w.cams).cuda()
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", "cuda")
for _ in range(5):
    plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end)
torch.cuda.synchronize()
for label, sel, mf in (("3_1 (64->64 s1)", 0 * 1000 + 64, 864), ("2_1 (32->32 s1)", 0 * 1000 + 32, 432), ("3_0 (32->64 s2)", 1000 + 32, 432),
                       ("4_0 (deconv 64->32)", 2000 + 64, 432), ("5_0 (deconv 32->16)", 2000 + 32, 432)):
    lib.mvs_os_prof_select(sel)
    plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 1024, np.int64)
    lib.mvs_os_prof_dump(buf.ctypes.data)
    t = buf.reshape(8, 1024)
    nb = int((t[0] > 0).sum())
    t = t[:, :nb].astype(np.float64)
    base = t[0].min()
    d = lambda a, b: np.median(t[b] - t[a])
    print("%s: blocks stamped %d; start spread %.0f cyc; median cycles: load-issue %.0f, affine %.0f, lds-write+barrier %.0f, K loop %.0f (%.1f per MFMA), stores %.0f, stats %.0f; first start -> last end %.0f"
          % (label, nb, t[0].max() - base, d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(3, 4) / mf, d(4, 5), d(5, 6), t[6].max() - base))
