"""Device-side timeline of the recurrent sweep (config c3) WITHOUT a profiler in the way: a variant build of the library
(tools/gru_trace.patch applied, -DMVS_TL) makes every workgroup of the cell-1 kernels (chain + x-part) and of the small-cell
kernels append (kernel id, first tick, last tick) of the 100 MHz wall clock to a buffer; this script runs sweeps, groups the
records into launches and prints what the chain's stream did.
    MVS_LIB_PATH=mvsnet_amd/variants/lib_tl.so python tools/gru_trace.py [--views 1] [--plane 120] [--planes 4]
kernel ids: cell 1 per plane: MODE*10 + (32 output channels ? 1 : 0)  -> 1 / 21 = gate convolution (plain / previous blend
folded), 10 = candidate; +100 = the batched x-part launches.  200.. = cell 2, 250.. = cell 3 (+ MODE*10, +1 matrix form)."""
import argparse, ctypes, os, sys, collections
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd import _lib, synthetic as S                              # noqa: E402
from mvsnet_amd.model import DepthPlan, MVSNetWeights, wta_depth_values  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=1)
ap.add_argument("--plane", type=int, default=120)
ap.add_argument("--planes", type=int, default=4)
ap.add_argument("--form", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
lib = _lib.load()
_lib.check(lib.mvs_gru_set_formulation(a.form), "mvs_gru_set_formulation")
w = S.make_workload("c3")
gp = S.make_gru_params("normal", seed=2, in_channels=w.channels, random_affine=True)
gw = MVSNetWeights.from_numpy("normal", gru=gp, device=dev)
cams = torch.as_tensor(w.cams).to(dev)
dv = wta_depth_values(w.depth_num, w.depth_start, w.depth_end, False)
B = a.views
feats = [torch.as_tensor(S.make_features(w.view_num, w.height, w.width, w.channels, seed=v)).to(dev) for v in range(B)]
plan = DepthPlan(w.view_num, w.depth_num, w.height, w.width, w.channels, gw, "GRU", dev, views=B)
for v in range(B):
    plan.set_cameras(cams, w.depth_start, w.depth_interval, w.depth_end, False, view=v)
for _ in range(2):
    plan.run_gru_batch(feats, [dv] * B)
torch.cuda.synchronize()
CAP = 1 << 20
bufs = {}
for name in ("mfma", "small"):
    b = torch.zeros(1 + 8 * CAP, dtype=torch.int64, device=dev)
    fn = getattr(lib, "mvs_tl_set_" + name); fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    assert fn(ctypes.c_void_p(b.data_ptr())) == 0
    bufs[name] = b
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
plan.run_gru_batch(feats, [dv] * B)
torch.cuda.synchronize()
print("sweep with tracing: %.2f ms (%d view(s))" % ((time.perf_counter() - t0) * 1e3, B))
recs = []
for name, b in bufs.items():
    h = b.cpu().numpy()
    n = min(int(h[0]), CAP)      # (the small-cell buffer fills up at four views per sweep: their statistics then cover the first planes only)
    r = h[1:1 + 8 * n].reshape(n, 8)
    recs.append(r)
    getattr(lib, "mvs_tl_set_" + name)(ctypes.c_void_p(0))
r = np.concatenate(recs)
tmin = r[:, 1].min()
# launches: per kernel id, records sorted by start; a new launch begins when a workgroup starts after every earlier one ended
launches = []
for kid in np.unique(r[:, 0]):
    x = r[r[:, 0] == kid]; x = x[np.argsort(x[:, 1])]
    s, e, n = x[0, 1], x[0, 2], 1
    for t0_, t1_ in x[1:, 1:3]:
        if t0_ > e:
            launches.append((int(kid), s, e, n)); s, e, n = t0_, t1_, 1
        else:
            e = max(e, t1_); n += 1
    launches.append((int(kid), s, e, n))
us = lambda t: (t - tmin) / 100.0
chain = sorted([l for l in launches if l[0] < 100], key=lambda l: l[1])
gates = [l for l in chain if l[0] in (1, 21)]
print("%d launches; chain launches %d (gate convolutions %d)" % (len(launches), len(chain), len(gates)))
span = us(max(l[2] for l in launches)) - us(min(l[1] for l in launches))
print("traced span %.2f ms" % (span / 1e3))
busy = sum(l[2] - l[1] for l in chain) / 100.0
gaps = [(chain[i + 1][1] - chain[i][2]) / 100.0 for i in range(len(chain) - 1)]
print("chain kernels: busy %.2f ms = %.0f %% of the span; gaps between consecutive chain kernels: mean %.1f us, median %.1f, sum %.2f ms"
      % (busy / 1e3, 100 * busy / span, np.mean(gaps), np.median(gaps), sum(gaps) / 1e3))
for kid in sorted(set(l[0] for l in launches)):
    d = [(l[2] - l[1]) / 100.0 for l in launches if l[0] == kid]
    print("  kernel %3d: %4d launches, duration mean %6.1f us  median %6.1f  min %6.1f  max %6.1f   workgroups/launch %d"
          % (kid, len(d), np.mean(d), np.median(d), min(d), max(d), int(np.median([l[3] for l in launches if l[0] == kid]))))
np.savez_compressed("gpurun_out/gru_trace_B%d.npz" % B, records=r, tmin=tmin)
# what slows the chain's kernels: their duration against what else was running, and how late their workgroups start
xs = [l for l in launches if 100 <= l[0] < 200]
def overlap(l, others):
    return sum(max(0, min(l[2], o[2]) - max(l[1], o[1])) for o in others) / float(max(1, l[2] - l[1]))
for kid in (21, 1, 10):
    ls = [l for l in chain if l[0] == kid]
    if not ls:
        continue
    ov = np.array([overlap(l, xs) for l in ls]); du = np.array([(l[2] - l[1]) / 100.0 for l in ls])
    hi, lo = du[ov > 0.8], du[ov < 0.2]
    print("  chain kernel %2d: %3d launches beside an x-part launch: mean %5.1f us;  %3d launches without: mean %5.1f us"
          % (kid, len(hi), hi.mean() if len(hi) else 0, len(lo), lo.mean() if len(lo) else 0))
    # start skew of the workgroups inside a launch
    x = r[r[:, 0] == kid]
    sk, life = [], []
    for l in ls[:: max(1, len(ls) // 64)]:
        m = x[(x[:, 1] >= l[1]) & (x[:, 2] <= l[2])]
        sk.append(np.percentile(m[:, 1] - l[1], [50, 90, 100]) / 100.0); life.append(np.median(m[:, 2] - m[:, 1]) / 100.0)
    sk = np.array(sk)
    print("      workgroup start after the launch's first one: median %.1f us, 90 %% %.1f us, last %.1f us; median workgroup lifetime %.1f us"
          % (sk[:, 0].mean(), sk[:, 1].mean(), sk[:, 2].mean(), np.mean(life)))
# phases inside a chain workgroup (first tile): start -> LayerNorm table read (weights requested, first tile's loads in flight)
# -> first tile staged -> its sweep done -> its store done / barrier; then the rest (second tile, statistics)
for kid in (21, 1, 10):
    x = r[(r[:, 0] == kid) & (r[:, 3] > 0)]
    if len(x) == 0:
        continue
    ph = np.stack([x[:, 3] - x[:, 1], x[:, 4] - x[:, 3], x[:, 5] - x[:, 4], x[:, 6] - x[:, 5], x[:, 2] - x[:, 6]], 1) / 100.0
    print("  chain kernel %2d phases (us, mean over %d workgroups): to affines %.1f | stage tile 1 %.1f | sweep %.1f | store + barrier %.1f | rest %.1f"
          % ((kid, len(x)) + tuple(ph.mean(0))))
# gap histogram by position in the group of 4 planes
PG = 4
bypos = collections.defaultdict(list)
for i in range(len(gates) - 1):
    nxt = gates[i + 1]
    prev = [l for l in chain if l[2] <= nxt[1] and l[1] >= gates[i][1]]
    if prev:
        bypos[(i + 1) % PG].append((nxt[1] - max(l[2] for l in prev)) / 100.0)
print("idle time on the chain before the gate convolution of plane d, by d mod %d: " % PG +
      "  ".join("%d: %.1f us" % (k, np.mean(v)) for k, v in sorted(bypos.items())))
# window
g0 = gates[a.plane][1]; g1 = gates[a.plane + a.planes][1]
print("\nwindow planes %d..%d: %.1f us per plane" % (a.plane, a.plane + a.planes - 1, (g1 - g0) / 100.0 / a.planes))
for l in sorted([l for l in launches if l[2] > g0 and l[1] < g1], key=lambda l: l[1]):
    lane = 0 if l[0] < 100 else 1 if l[0] < 200 else 2 if l[0] < 250 else 3
    print("   %s%8.1f -> %8.1f  dur %6.1f  kernel %3d  (%d workgroups)" % ("        " * lane, (l[1] - g0) / 100.0, (l[2] - g0) / 100.0, (l[2] - l[1]) / 100.0, l[0], l[3]))
