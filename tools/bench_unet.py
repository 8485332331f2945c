"""Developer benchmark: UNetDS2GN feature extractor, HIP library vs PyTorch/MIOpen (5 views of 512x640)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import synthetic as S
from mvsnet_amd.feature_net import UNetDS2GN
from mvsnet_amd.feature_net_hip import HipUNetDS2GN

V, H, W = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (5, 512, 640)))
dev = torch.device("cuda", 0)
params = S.make_unet_params("normal", seed=3)
img = torch.randn(V, H, W, 3, device=dev)
for name, net in (("hip", HipUNetDS2GN(params, dev)), ("torch", UNetDS2GN(params, dev))):
    for _ in range(3):
        out = net(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = net(img)
    torch.cuda.synchronize()
    print("%-6s %.3f ms per %d views  (%s)" % (name, (time.perf_counter() - t0) / 10 * 1e3, V, tuple(out.shape)))
