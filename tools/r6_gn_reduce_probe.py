"""What bounds mvs_gn_bwd_reduce_f32: microseconds per launch (torch events around 50 back-to-back launches, so ~9 us of launch
gap are in every number) over (V, hw, C), beside the element-wise mvs_gn_bwd_apply(_tot)_f32 on the same tensors."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); P = _lib.ptr
def timed(fn, n=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("%4s %8s %4s | %8s %8s %8s" % ("V", "hw", "C", "reduce", "apply", "apply+tot"))
for V, hw, C in ((3, 1200, 128), (1, 1200, 128), (3, 1200, 8), (3, 4800, 64), (3, 19200, 32), (3, 76800, 16), (3, 307200, 8), (1, 307200, 8), (3, 64, 8)):
    x = torch.randn(V, hw, C, device=dev); g = torch.randn(V, hw, C, device=dev)
    stats = torch.zeros(V, 2, C, dtype=torch.float64, device=dev)
    _lib.check(lib.mvs_gn_stats_f32(P(x), V, hw, C, P(stats), _lib.stream_ptr()), "stats")
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    sums = torch.zeros(lib.mvs_gn_bwd_sums_doubles(V, C), dtype=torch.float64, device=dev)
    tot = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    dx = torch.empty_like(x)
    st = _lib.stream_ptr()
    r = timed(lambda: lib.mvs_gn_bwd_reduce_f32(P(x), P(stats), P(gamma), P(beta), 1e-5, 1, P(g), V, hw, C, P(sums), st))
    a = timed(lambda: lib.mvs_gn_bwd_apply_f32(P(x), P(stats), P(gamma), P(beta), 1e-5, 1, P(g), P(sums), V, hw, C, P(dx), st))
    t = timed(lambda: lib.mvs_gn_bwd_apply_tot_f32(P(x), P(stats), P(gamma), P(beta), 1e-5, 1, P(g), P(sums), P(tot), V, hw, C, P(dx), st))
    print("%4d %8d %4d | %8.1f %8.1f %8.1f" % (V, hw, C, r, a, t), flush=True)
