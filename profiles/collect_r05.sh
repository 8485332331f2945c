#!/bin/bash
# Round-5 profile collection; run on the GPU box:  gpurun -- 'bash profiles/collect_r05.sh'
# Kernel-trace/stats and PMC counters are collected in SEPARATE rocprofv3 runs (the pool refuses
# --pmc combined with trace domains other than --kernel-trace).  The program itself follows `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_$n.log 2>&1 || echo "PMC pass $n failed"
done
python profiles/summarize.py $OUT gpurun_out/r05/r05
# the recurrent sweep (config c3) alone, one and four reference views per launch: per-kernel time inside a sweep
for B in 1 4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gru_stats_B$B -- python tools/gru_time.py --views $B --iters 3 > $OUT/gru_stats_B$B.log 2>&1
  f=$(ls $OUT/gru_stats_B$B/*/*kernel_stats.csv | head -1)
  python - "$f" gpurun_out/r05/r05_gru_kernel_stats_B$B.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
done
ls -la gpurun_out/r05/r05*
# SQ counters of the 4-view recurrent sweep, per kernel and plane
bash tools/gru_pmc.sh > gpurun_out/r05/r05_gru_pmc_B4.txt 2>&1
# the default bench record of the round (what the driver runs)
timeout -k 10 600 python bench.py > gpurun_out/r05/r05_bench_default.json 2> gpurun_out/r05/r05_bench_default.err
ls -la gpurun_out/r05/r05*
# round 5: device-side time line of the fused recurrent sweep, the probes behind its design, the capture reproducer
for B in 1 4; do timeout -k 10 200 python tools/gru_fused_trace.py --views $B 2>&1 | grep -A1 launch; done > gpurun_out/r05/r05_gru_fused_trace.txt
timeout -k 5 100 ./tools/bin/coissue_probe > gpurun_out/r05/r05_coissue_probe.txt 2>&1
timeout -k 5 200 ./tools/bin/store_hazard_probe > gpurun_out/r05/r05_store_hazard_probe.txt 2>&1
timeout -k 10 300 python tools/gru_graph_time.py 2>&1 | grep "c3 sweep" > gpurun_out/r05/r05_gru_graph_replay.txt
ls -la gpurun_out/r05/
