#!/bin/bash
# rocprofv3 kernel stats of the recurrent sweep at c3 for B = 1 and B = 4 views per launch -> gpurun_out/r04_gru_stats_B*.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for B in ${VIEWS:-1 4}; do
  O=gpurun_out/gruprof_B$B
  rm -rf $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python tools/gru_time.py --views $B --iters 3 > gpurun_out/gruprof_B$B.log 2>&1
  f=$(ls $O/*/*kernel_stats.csv | head -1)
  cp $f gpurun_out/r04_gru_kernel_stats_B$B.csv
  grep "c3 sweep" gpurun_out/gruprof_B$B.log
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r04_gru_kernel_stats_B$B.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("B=$B total kernel time %.1f ms over 5 sweeps (2 warm-up + 3)"%(tot/1e6))
for r in rows[:16]:
    n=r["Name"]; n=n.replace("(anonymous namespace)::","").replace("void ","")[:70]
    print("%-70s calls %5s avg %8.1f us  total %7.2f ms  min %7.1f max %8.1f"%(n,r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3))
PY
  rm -rf $O
done
