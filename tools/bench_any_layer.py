"""Times ONE 3x3x3 convolution layer alone on an idle GPU: python tools/bench_any_layer.py D H W Cin Cout [stride] [iters].
(A/B switches are environment variables read by the library, e.g. MVS_NO_BLK64=1.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import model as M


def main():
    D, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
    stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 200
    x = torch.randn(D, H, W, Cin, device="cuda")
    w = torch.randn(3, 3, 3, Cin, Cout, device="cuda") * 0.05
    stats = torch.zeros(2, Cout, device="cuda", dtype=torch.float64)
    sc, sh = torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    for _ in range(10):
        M.conv3d(x, w, stride, x_affine=(sc, sh), stats=stats)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        M.conv3d(x, w, stride, x_affine=(sc, sh), stats=stats)
    e1.record()
    torch.cuda.synchronize()
    print("%dx%dx%d %d->%d s%d: %.1f us per launch" % (D, H, W, Cin, Cout, stride, e0.elapsed_time(e1) / iters * 1e3))


if __name__ == "__main__":
    main()
