"""the fused 3dconv1_1 + 2_0 launch (conv3d_s1_kernel<16,16,8,...,FUSE2>) with two waves per SIMD.  At the metric
size the launcher picks 16 planes per workgroup = 240 workgroups = one per CU; the test hook MVS_HOOK_FUSE2_PLANES forces 8 (480
workgroups, two per CU), 12, 24, 32.  Per setting: depth maps/s of the whole hot path (200 steps, three repetitions) and the
launch's own time from the library's per-layer event brackets (mvs_profile_layers).   python tools/r6_fuse2_planes.py"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights
dev = torch.device("cuda", 0)
w = S.make_workload("M")
weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1, random_affine=True), device=dev)
feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
end = w.depth_start + (w.depth_num - 1) * w.depth_interval
lib = _lib.load()
run = lambda: plan.run_depth(feats, cams, w.depth_start, w.depth_interval, end, False)
for _ in range(50): run()
torch.cuda.synchronize()
ref = plan.depth.clone()
LAYERS = ["3dconv1_0", "3dconv2_0", "3dconv3_0", "3dconv0_1", "3dconv1_1", "3dconv2_1", "3dconv3_1", "3dconv4_0", "3dconv5_0", "3dconv6_0", "3dconv6_2"]
for planes in (0, 8, 12, 16, 24, 32, 0):
    _lib.set_test_hook("fuse2_planes", planes)
    for _ in range(20): run()
    torch.cuda.synchronize()
    rates = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): run()
        torch.cuda.synchronize()
        rates.append(200 / (time.perf_counter() - t0))
    same = bool(torch.equal(plan.depth, ref))
    ms, n = (ctypes.c_double * 11)(), ctypes.c_int(0)
    lib.mvs_profile_layers(1)
    for _ in range(20): run()
    torch.cuda.synchronize()
    lib.mvs_profile_layers_ms(ms, ctypes.byref(n)); lib.mvs_profile_layers(0)
    print("planes per workgroup %2d (%s): %s depth maps/s; fused 3dconv1_1 + 2_0 launch %.1f us; same depth map: %s" % (
        planes, "launcher's choice" if planes == 0 else "%d workgroups" % (40 * -(-96 // planes)), " ".join("%.1f" % r for r in rates),
        ms[LAYERS.index("3dconv1_1")] * 1e3, same), flush=True)
_lib.set_test_hook("fuse2_planes", 0)
