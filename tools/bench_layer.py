"""Developer micro-benchmark: one RegNetUS0 layer (or the fused cost-volume pair) at workload-M size.

    python tools/bench_layer.py pair|c8|s2|deconv6|out [--iters 20]

Used under rocprofv3 (--kernel-trace --stats, or --pmc ...) to study a single kernel."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import model as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dhw", default="192,128,160")
    ap.add_argument("--no-stats", action="store_true", help="do not accumulate BatchNorm sums (measures the atomics' tail)")
    ap.add_argument("--data", default="randn", help="randn | const (low toggle rate: clocks stay high)")
    a = ap.parse_args()
    D, H, W = (int(v) for v in a.dhw.split(","))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    r = (lambda *s: torch.randn(*s, generator=g).to(dev)) if a.data == "randn" else (lambda *s: torch.full(s, 0.001).to(dev))
    if a.what == "pair":
        x, w1, w2 = r(D, H, W, 32), r(3, 3, 3, 32, 8) * 0.03, r(3, 3, 3, 32, 16) * 0.03
        s1 = torch.zeros(2, 8, dtype=torch.float64, device=dev); s2 = torch.zeros(2, 16, dtype=torch.float64, device=dev)
        fn = (lambda: M.conv3d_pair(x, w1, w2, None, None)) if a.no_stats else (lambda: M.conv3d_pair(x, w1, w2, s1, s2))
        flops = 2.0 * 27 * 32 * (D * H * W * 8 + D * H * W / 8 * 16)
    elif a.what == "c8":
        x, w1 = r(D, H, W, 32), r(3, 3, 3, 32, 8) * 0.03
        s1 = torch.zeros(2, 8, dtype=torch.float64, device=dev)
        fn = lambda: M.conv3d(x, w1, 1, stats=s1)
        flops = 2.0 * 27 * 32 * D * H * W * 8
    elif a.what == "s2":
        x, w2 = r(D, H, W, 32), r(3, 3, 3, 32, 16) * 0.03
        fn = lambda: M.conv3d(x, w2, 2)
        flops = 2.0 * 27 * 32 * D * H * W / 8 * 16
    elif a.what == "deconv6":
        x, sk, w = r(D // 2, H // 2, W // 2, 16), r(D // 2, H // 2, W // 2, 16), r(3, 3, 3, 8, 16) * 0.05
        one, zero = torch.ones(16, device=dev), torch.zeros(16, device=dev)
        s = torch.zeros(2, 8, dtype=torch.float64, device=dev)
        fn = lambda: M.conv3d(x, w, 1, (one, zero), sk, (one, zero), None if a.no_stats else s, transpose=True)
        flops = 2.0 * 27 * 16 * 8 * D * H * W / 8
    elif a.what == "out":
        x, sk, w = r(D, H, W, 8), r(D, H, W, 8), r(3, 3, 3, 8, 1) * 0.07
        one, zero = torch.ones(8, device=dev), torch.zeros(8, device=dev)
        fn = lambda: M.conv3d(x, w, 1, (one, zero), sk, (one, zero))
        flops = 2.0 * 27 * 8 * D * H * W
    else:
        raise SystemExit("unknown layer " + a.what)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print("%s: %.1f us  %.1f TFLOP/s" % (a.what, ms * 1e3, flops / ms / 1e9))


if __name__ == "__main__":
    main()
