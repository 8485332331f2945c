import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights
dev = torch.device("cuda", 0)
weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1, random_affine=True), device=dev)
lib = _lib.load()
LAYERS = ["3dconv1_0", "3dconv2_0", "3dconv3_0", "3dconv0_1", "3dconv1_1", "3dconv2_1", "3dconv3_1", "3dconv4_0", "3dconv5_0", "3dconv6_0", "3dconv6_2"]
res = {}
for name in ("M", "c2"):
    w = S.make_workload(name)
    feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
    end = w.depth_start + (w.depth_num - 1) * w.depth_interval
    run = lambda: plan.run_depth(feats, cams, w.depth_start, w.depth_interval, end, False)
    for _ in range(20): run()
    torch.cuda.synchronize()
    ms, n = (ctypes.c_double * 11)(), ctypes.c_int(0)
    st, sn = (ctypes.c_double * 3)(), ctypes.c_int(0)
    lib.mvs_profile_layers(1)
    for _ in range(20): run()
    torch.cuda.synchronize()
    lib.mvs_profile_layers_ms(ms, ctypes.byref(n)); lib.mvs_profile_layers(0)
    lib.mvs_profile_stages(1)
    for _ in range(20): run()
    torch.cuda.synchronize()
    lib.mvs_profile_stages_ms(st, ctypes.byref(sn)); lib.mvs_profile_stages(0)
    res[name] = ([ms[i] * 1e3 for i in range(11)], [st[i] * 1e3 for i in range(3)], w.depth_num * w.height * w.width)
    del plan
r = res["c2"][2] / res["M"][2]
print("voxel ratio c2 / M = %.3f" % r)
for i, l in enumerate(LAYERS):
    a, b = res["M"][0][i], res["c2"][0][i]
    if a > 0: print("%-10s M %7.1f us   c2 %7.1f us   ratio %.2f  (%.2f of linear)" % (l, a, b, b / a, b / a / r))
for i, l in enumerate(("warp+variance", "RegNetUS0 stack", "soft-argmin")):
    a, b = res["M"][1][i], res["c2"][1][i]
    print("%-16s M %7.1f us   c2 %7.1f us   ratio %.2f  (%.2f of linear)" % (l, a, b, b / a, b / a / r))
