"""Captures the round-4 four-stream wavefront (formulation 1) of the library itself into a hipGraph -- the case that crashed the
host in round 4 (profiles/r04_gru_wavefront_capture_segfault.log).  MVS_GRU_CAPTURE_WAVEFRONT=1 lifts the library's
"one stream under capture" rule.      python tools/capture_wavefront_lib.py [torch|raw] [planes]
  torch: torch.cuda.graph (global capture mode, torch's private pool)      raw: hipStreamBeginCapture / EndCapture through ctypes"""
import ctypes as C, faulthandler, os, sys
faulthandler.enable()
os.environ["MVS_GRU_CAPTURE_WAVEFRONT"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values
mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
base = S.make_workload("c3")
Hh, Ww = 52, 72
gp = S.make_gru_params("normal", seed=2, in_channels=base.channels, random_affine=True)
weights = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
feats = torch.as_tensor(S.make_features(base.view_num, base.height, base.width, base.channels, seed=11)[:, :Hh, :Ww]).to(dev)
end = base.depth_start + (D - 1) * base.depth_interval
dv = wta_depth_values(D, base.depth_start, end, False)
lib = _lib.load()
_lib.check(lib.mvs_gru_set_formulation(1), "form")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    plan = DepthPlan(base.view_num, D, Hh, Ww, base.channels, weights, "GRU", dev)
    plan.set_cameras(torch.as_tensor(base.cams).to(dev), base.depth_start, base.depth_interval, end, False)
    d, p = plan.run_gru(feats, dv)
    s.synchronize()
    eager = d.clone()
    print("eager wavefront done; capturing (%s) ..." % mode, flush=True)
    if mode == "torch":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            plan.run_gru(feats, dv)
        print("capture_end returned", flush=True)
        plan.depth.zero_(); g.replay(); s.synchronize()
    else:
        hip = C.CDLL("libamdhip64.so")
        st = C.c_void_p(s.cuda_stream)
        graph, ex = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamBeginCapture(st, 0) == 0
        plan.run_gru(feats, dv)
        rc = hip.hipStreamEndCapture(st, C.byref(graph))
        print("hipStreamEndCapture rc", rc, flush=True)
        assert hip.hipGraphInstantiate(C.byref(ex), graph, None, None, 0) == 0
        plan.depth.zero_()
        assert hip.hipGraphLaunch(ex, st) == 0
        s.synchronize()
    print("replayed; depth identical to the eager wavefront:", bool(torch.equal(plan.depth, eager)), flush=True)
