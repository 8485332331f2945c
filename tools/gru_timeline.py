"""Timeline of the recurrent sweep from a rocprofv3 --kernel-trace CSV (tools/gru_prof.sh / profiles/collect_rNN.sh):
per hardware queue, what ran when, for a window of planes in the middle of the LAST sweep of the run.
    python tools/gru_timeline.py <kernel_trace.csv> [first_plane] [planes]
At four views per sweep the profiler slows the sweep by ~5 % (host-side interception), at one view by ~35 % (the sweep becomes
enqueue-bound): read the one-view timeline as a distorted one."""
import csv, sys, collections

f = sys.argv[1]
p0 = int(sys.argv[2]) if len(sys.argv) > 2 else 120
npl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    r["n"] = n.split("(")[0][:46]
rows.sort(key=lambda r: r["s"])
# the last sweep = after the last wta_finish / zero kernel gap: take the last 1/k of the gate-convolution launches of cell 1
gates = [r for r in rows if "conv2d_cat_mfma_kernel" in r["n"] and r["n"].rstrip(">").split(", ")[3] in ("0", "2") and "true>" not in r["n"].split(", ")[-1]]
D = 256
sweeps = len(gates) // D
last = gates[-D:]
t_begin, t_end = last[0]["s"], rows[-1]["e"]
print("%d gate-convolution launches of cell 1 = %d sweeps; last sweep spans %.2f ms" % (len(gates), sweeps, (t_end - t_begin) / 1e6))
w0, w1 = last[p0]["s"], last[p0 + npl]["s"]
print("window: planes %d..%d of the last sweep, %.1f us (%.1f us per plane)" % (p0, p0 + npl - 1, (w1 - w0) / 1e3, (w1 - w0) / 1e3 / npl))
win = [r for r in rows if r["e"] > w0 and r["s"] < w1]
byq = collections.defaultdict(list)
for r in win:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(min(r["e"], w1) - max(r["s"], w0) for r in rs)
    print("\nqueue %s: %d launches, busy %.0f %% of the window" % (q, len(rs), 100.0 * busy / (w1 - w0)))
    prev_e = None
    for r in rs:
        gap = (r["s"] - prev_e) / 1e3 if prev_e is not None else 0.0
        print("   %8.1f -> %8.1f us  dur %6.1f  gap %6.1f  %s  grid %s" % ((r["s"] - w0) / 1e3, (r["e"] - w0) / 1e3, (r["e"] - r["s"]) / 1e3, gap, r["n"], r["Grid_Size_X"]))
        prev_e = r["e"]
# concurrency histogram over the whole last sweep
ev = []
for r in rows:
    if r["e"] > t_begin and r["s"] < t_end:
        ev.append((max(r["s"], t_begin), 1)); ev.append((min(r["e"], t_end), -1))
ev.sort()
hist = collections.Counter(); cur = 0; t = t_begin
for tt, d in ev:
    hist[cur] += tt - t; t = tt; cur += d
tot = float(sum(hist.values()))
print("\nkernels in flight over the last sweep: " + "  ".join("%d: %.0f %%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
