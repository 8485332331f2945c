#!/bin/bash
# round 5: per-kernel time of the fused sweep under rocprofv3 (B = 1, 4) and its SQ counters at B = 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05; mkdir -p $OUT
for B in 1 4; do
  rm -rf $OUT/gru_stats_B$B
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gru_stats_B$B -- python tools/gru_time.py --views $B --iters 3 > $OUT/gru_stats_B$B.log 2>&1
  f=$(ls $OUT/gru_stats_B$B/*/*kernel_stats.csv | head -1)
  python - "$f" $OUT/r05_gru_kernel_stats_B$B.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
  grep "c3 sweep" $OUT/gru_stats_B$B.log
  head -8 $OUT/r05_gru_kernel_stats_B$B.csv
done
bash tools/gru_pmc.sh > $OUT/r05_gru_pmc_B4.txt 2>&1
cat $OUT/r05_gru_pmc_B4.txt
