"""Round 6 experiment: GroupNorm is per SAMPLE, so the V images of a tower pass are independent 32-layer chains.  Here every image
runs its chain on a stream of its own (V = 1 launches into its own buffers) and the whole pass -- V x 32 launches, fork / join --
is captured into ONE hipGraph: the fixed part of a layer (launch gap, prologue, tail: ~9 us of ~30) of one image overlaps with the
matrix work of the others.  Against the batched in-line pass (one launch per layer for all images).
    python tools/r6_unet_views.py [V H W]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S
from mvsnet_amd.feature_net_hip import HipUNetDS2GN
V, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 512, 640)
dev = torch.device("cuda", 0)
net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev, side_streams=0)
img = torch.randn(V, H, W, 3, device=dev)


def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3


ref = net(img).clone()
ms, host = timed(lambda: net(img))
print("batched, in line, eager:               %.3f ms per pass (host enqueue %.3f ms)" % (ms, host))
net2 = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev, side_streams=2)
ms, host = timed(lambda: net2(img))
print("batched, two side streams, eager:      %.3f ms per pass (host enqueue %.3f ms)" % (ms, host))

streams = [torch.cuda.Stream(dev) for _ in range(V)]
for groups in ([[v] for v in range(V)], [[0, 1, 2], [3, 4]] if V == 5 else None):
    if groups is None:
        continue
    outs = [None] * len(groups)

    def per_view():
        main = torch.cuda.current_stream(dev)
        for gi, g in enumerate(groups):
            s = streams[gi]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs[gi] = net._run(img[g[0]:g[-1] + 1], [], slot=1 + gi)
        for gi in range(len(groups)):
            main.wait_stream(streams[gi])
    per_view(); torch.cuda.synchronize()
    got = torch.cat(outs)
    print("%d chains of %s images: max |difference| to the batched pass %.2e" % (len(groups), [len(g) for g in groups], float((got - ref).abs().max())))
    ms, host = timed(per_view, 20)
    print("  eager:                                %.3f ms per pass (host enqueue %.3f ms)" % (ms, host))
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            per_view()
        torch.cuda.synchronize()
        ms, host = timed(g.replay)
        got = torch.cat(outs)
        print("  replayed from one hipGraph:           %.3f ms per pass (host %.3f ms); max |difference| %.2e" % (ms, host, float((got - ref).abs().max())))
    except Exception as e:
        print("  graph capture failed:", repr(e)[:200])
