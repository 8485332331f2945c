"""Round-3 bisect of "the c3 recurrent sweep takes 44 ms at the tail of bench.py and 23 ms in a fresh process".

    python tools/gru_bisect.py <pre-step>[,<pre-step>...] [--caller null|stream]

Each pre-step replays one thing bench.py's main() does before it reaches `config_c3_gru`; the sweep is then timed in the
same process.  Run every combination in a process of its own (tools/gru_bisect.sh) and keep the log under profiles/.

pre-steps:
  none        nothing (the pristine-process figure)
  nullcnn     20 metric depth maps on the null stream
  streamcnn   20 metric depth maps on a torch.cuda.Stream that stays alive (bench.py's `streams[0]`)
  two         bench.py's two-stream pass: two more torch streams (alive), 48 depth maps
  twofree     the same, streams and plans released afterwards
  c2          configs[1] (288x216, D=192; 1.5 GB volume), plan released + empty_cache()
  tevents     80 timing-enabled torch events recorded on a side stream
  layers      mvs_profile_layers on -> 20 depth maps -> off
  dominant    mvs_profile_dominant on -> 20 depth maps -> off
  streams4    four idle torch streams that stay alive (one tiny kernel each)
  streams8    eight
  towers      UNetDS2GN towers on the HIP library, 5 views 512x640, 8 times (the production order towers -> sweep)
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S                                   # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("pre")
ap.add_argument("--caller", default="null", choices=["null", "stream"])
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.load()
keep = []          # objects bench.py's main() still holds when it reaches the recurrent configuration


def metric_plan():
    w = S.make_workload("M")
    weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1), device=dev)
    feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
    plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
    return w, weights, feats, cams, plan


def run_metric(n, stream=None, plans=None):
    w, weights, feats, cams, plan = metric_plan()
    if stream is None:
        for _ in range(n):
            plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
    else:
        with torch.cuda.stream(stream):
            for _ in range(n):
                plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
    torch.cuda.synchronize()
    return w, weights, feats, cams, plan


def pre(step):
    if step == "none":
        return
    if step == "nullcnn":
        run_metric(20)
    elif step == "streamcnn":
        s = torch.cuda.Stream(device=dev)
        keep.append((s, run_metric(20, s)))
    elif step in ("two", "twofree"):
        w, weights, feats, cams, plan = metric_plan()
        p2 = [plan, DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)]
        s2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for i in range(48):
            with torch.cuda.stream(s2[i % 2]):
                p2[i % 2].run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
        torch.cuda.synchronize()
        if step == "two":
            keep.append((p2, s2))
        else:
            del p2, s2, plan
            torch.cuda.empty_cache()
    elif step == "c2":
        w = S.make_workload("c2")
        weights = MVSNetWeights.from_numpy("normal", regnet=S.make_regnet_params("normal", seed=1, random_affine=True), device=dev)
        feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
        plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, weights, "3DCNN", dev)
        for _ in range(13):
            plan.run_depth(feats, cams, w.depth_start, w.depth_interval, w.depth_end, False)
        torch.cuda.synchronize()
        del plan
        torch.cuda.empty_cache()
    elif step == "tevents":
        s = torch.cuda.Stream(device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(80)]
        with torch.cuda.stream(s):
            for e in ev:
                torch.zeros(8, device=dev).add_(1)
                e.record()
        torch.cuda.synchronize()
        keep.append((s, ev))
    elif step in ("layers", "dominant"):
        fn = lib.mvs_profile_layers if step == "layers" else lib.mvs_profile_dominant
        _lib.check(fn(1), step)
        run_metric(20)
        if step == "layers":
            ms, n = (ctypes.c_double * 11)(), ctypes.c_int(0)
            _lib.check(lib.mvs_profile_layers_ms(ms, ctypes.byref(n)), step)
        else:
            ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
            _lib.check(lib.mvs_profile_dominant_ms(ctypes.byref(ms), ctypes.byref(n)), step)
        _lib.check(fn(0), step)
    elif step in ("streams4", "streams8"):
        ss = [torch.cuda.Stream(device=dev) for _ in range(int(step[7:]))]
        for s in ss:
            with torch.cuda.stream(s):
                torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
        keep.append(ss)
    elif step in ("homog", "cost", "soft", "memset", "torchbig"):      # ONE kind of work on a torch stream that stays alive
        s = torch.cuda.Stream(device=dev)
        w, weights, feats, cams, plan = metric_plan()
        from mvsnet_amd.model import cost_volume, softargmin_prob
        with torch.cuda.stream(s):
            for _ in range(20):
                if step == "homog":
                    plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
                elif step == "cost":
                    cost_volume(feats[0], feats[1:], plan.transforms, 0, plan.D, "mem", out=plan.cost)
                elif step == "soft":
                    softargmin_prob(plan.reg, w.depth_start, w.depth_interval, False, plan.depth, plan.prob)
                elif step == "memset":
                    rt = ctypes.CDLL("libamdhip64.so")
                    rt.hipMemsetAsync(ctypes.c_void_p(plan.reg.data_ptr()), 0, ctypes.c_size_t(plan.reg.numel() * 4), ctypes.c_void_p(s.cuda_stream))
                else:
                    a_ = torch.randn(2048, 2048, device=dev); (a_ @ a_).sum()
        torch.cuda.synchronize()
        keep.append((s, plan))
    elif step == "flags":
        rt = ctypes.CDLL("libamdhip64.so")
        s = torch.cuda.Stream(device=dev)
        fl, pr = ctypes.c_uint(99), ctypes.c_int(99)
        rt.hipStreamGetFlags(ctypes.c_void_p(s.cuda_stream), ctypes.byref(fl))
        rt.hipStreamGetPriority(ctypes.c_void_p(s.cuda_stream), ctypes.byref(pr))
        lo, hi = ctypes.c_int(0), ctypes.c_int(0)
        rt.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
        print("torch stream: flags %d (hipStreamNonBlocking = 1), priority %d; device priority range least %d greatest %d" % (fl.value, pr.value, lo.value, hi.value), flush=True)
        keep.append(s)
    elif step == "towers":
        from mvsnet_amd.feature_net_hip import HipUNetDS2GN
        net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev)
        imgs = torch.as_tensor(S.make_images(5, 512, 640, seed=0)).to(dev)
        for _ in range(8):
            f = net(imgs)
        torch.cuda.synchronize()
        keep.append((net, imgs, f))
    else:
        raise SystemExit("unknown pre-step " + step)


for st in a.pre.split(","):
    pre(st)

w = S.make_workload("c3")
gw = MVSNetWeights.from_numpy("normal", gru=S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True), device=dev)
feats, cams = torch.as_tensor(w.features).to(dev), torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev)
caller = torch.cuda.current_stream() if a.caller == "null" else torch.cuda.Stream(device=dev)


def run(n):
    with torch.cuda.stream(caller):
        for _ in range(n):
            plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False)
            plan.run_gru(feats, dv)


run(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.iters)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.iters * 1e3
print("pre=%-28s caller=%-6s GPU_MAX_HW_QUEUES=%-4s %7.2f ms per depth map (host enqueue %6.2f ms)  checksum %.6f" % (
    a.pre, a.caller, os.environ.get("GPU_MAX_HW_QUEUES", "-"), ms, t_host / a.iters * 1e3, float(plan.depth.double().sum())), flush=True)
