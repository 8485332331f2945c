"""Device-side time line of ONE persistent tower launch (tools/r6_unet_p_trace.patch built with -DMVS_PTRACE into
mvsnet_amd/variants/lib_ptrace.so): wall-clock stamps (100 MHz) of workgroup entry, weights staged, and per tile: top barrier
passed / staged / MFMAs done / stores issued; exit.
    MVS_LIB_PATH=mvsnet_amd/variants/lib_ptrace.so python tools/r6_unet_p_trace.py V H W C1 C2 Cout [grid]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib as L
V, H, W, C1, C2, Cout = (int(a) for a in sys.argv[1:7])
grid = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda", 0)
lib = L.load()
raw = ctypes.CDLL(L.LIB_PATH)
raw.mvs_unet_ptrace.argtypes = [ctypes.c_void_p, ctypes.c_int]
SL = lib.mvs_gn_stat_slots()
x1 = torch.randn(V, H, W, C1, device=dev); x2 = torch.randn(V, H, W, C2, device=dev) if C2 else None
w = torch.randn(3, 3, C1 + C2, Cout, device=dev) * 0.1
wp = torch.empty(lib.mvs_conv2d_prepared_floats(3, C1, C2, Cout), device=dev)
L.check(lib.mvs_conv2d_prepare_f32(L.ptr(w), 3, C1, C2, Cout, L.ptr(wp), L.stream_ptr()))
s1 = torch.rand(V, C1 // 8, SL, 2, dtype=torch.float64, device=dev) * H * W
s1[..., 1] += s1[..., 0] ** 2 / (H * W * 8 / SL) * SL
s2 = s1[:, :max(C2 // 8, 1)].clone() if C2 else None
g1 = torch.ones(C1, device=dev); b1 = torch.zeros(C1, device=dev)
g2 = torch.ones(max(C2, 1), device=dev); b2 = torch.zeros(max(C2, 1), device=dev)
y = torch.empty(V, H, W, Cout, device=dev); so = torch.zeros(V, Cout // 8, SL, 2, dtype=torch.float64, device=dev)
if grid:
    L.set_test_hook("unet_grid", grid)
def run():
    L.check(lib.mvs_conv2d_gn_f32(L.ptr(x1), L.ptr(s1), L.ptr(g1), L.ptr(b1), C1, 1, L.ptr(x2), L.ptr(s2), L.ptr(g2) if C2 else None,
                                  L.ptr(b2) if C2 else None, C2, 0, L.ptr(wp), V, H, W, Cout, 3, 1, L.ptr(y), L.ptr(so), L.stream_ptr()))
for _ in range(5):
    run()
torch.cuda.synchronize()
cap = 4096
buf = torch.zeros(1 + 32 * cap, dtype=torch.int64, device=dev)
raw.mvs_unet_ptrace(ctypes.c_void_p(buf.data_ptr()), cap)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
raw.mvs_unet_ptrace(None, 0)
h = buf.cpu().numpy(); n = int(h[0]); r = h[1:1 + 32 * n].reshape(n, 32)
t0 = r[:, 0].min()
us = lambda t: (t - t0) / 100.0
print("launch: %d workgroups, event time %.1f us, first entry -> last exit %.1f us" % (n, e0.elapsed_time(e1) * 1e3, us(r[:, 30].max())))
print("entry: median %.1f us, last %.1f us | prologue after entry (medians): loads requested + weights written %.2f, GroupNorm table %.2f, first tile staged %.2f us" % (
    np.median(us(r[:, 0])), us(r[:, 0]).max(), np.median(r[:, 28] - r[:, 0]) / 100.0, np.median(r[:, 29] - r[:, 0]) / 100.0, np.median(r[:, 1] - r[:, 0]) / 100.0))
hw = r[:, 31] >> 32                        # HW_ID register: CU / SE / XCC fields identify the compute unit
cu = hw & 0xfff0                           # drop the wave / SIMD bits, keep CU, SH, SE (and what lies above)
ntile = ((r[:, 2:26].reshape(n, 8, 3)[:, :, 0] > 0).sum(1))
print("tiles per workgroup (first 8 traced): min %d max %d" % (ntile.min(), ntile.max()))
import collections
per_cu = collections.Counter()
for i in range(n):
    per_cu[int(hw[i] >> 4)] += int(ntile[i])
vals = sorted(per_cu.values())
print("distinct HW_ID (bits 4..) values: %d; tiles per value: min %d median %d max %d" % (len(per_cu), vals[0], vals[len(vals) // 2], vals[-1]))
for k in range(int(ntile.max())):
    m = ntile > k
    q = r[m][:, 2 + 3 * k: 5 + 3 * k]
    print("tile %d (%4d wgs): MFMAs with everything in their shadow %.2f | copy + barrier %.2f us (medians); start at %.1f us" % (
        k, m.sum(), np.median(q[:, 1] - q[:, 0]) / 100.0, np.median(q[:, 2] - q[:, 1]) / 100.0, np.median(us(q[:, 0]))))
print("exit median %.1f us, last %.1f us" % (np.median(us(r[:, 30])), us(r[:, 30]).max()))
