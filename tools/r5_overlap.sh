#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_overlap; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python tools/gru_time.py --views 4 --iters 2 > $O/log.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05_overlap/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
fused = [(s, e) for s, e, n, q in ks if "gru_fused" in n]
cost = [(s, e, q) for s, e, n, q in ks if "cost_volume_sweep" in n]
print("queues:", sorted(set(q for _, _, n, q in ks if "gru_fused" in n)), "fused |", sorted(set(q for _, _, q in cost)), "cost")
tot = 0; ov = 0
import bisect
starts = [s for s, e in fused]
for s, e, q in cost[8:]:
    tot += e - s
    i = max(0, bisect.bisect_left(starts, s) - 2)
    while i < len(fused) and fused[i][0] < e:
        ov += max(0, min(e, fused[i][1]) - max(s, fused[i][0])); i += 1
print("cost-slice launches: %d, total %.1f ms, of which overlapped with a fused launch %.1f ms; mean duration %.1f us" % (len(cost) - 8, tot / 1e6, ov / 1e6, tot / max(1, len(cost) - 8) / 1e3))
PY
grep "c3 sweep" $O/log.txt
