#!/bin/bash
# Round-2 profile collection; run on the GPU box:  gpurun -- 'bash profiles/collect_r02.sh'
# Kernel-trace/stats and PMC counters are collected in SEPARATE rocprofv3 runs (the pool refuses
# --pmc combined with trace domains other than --kernel-trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r02
mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_$n.log 2>&1 || echo "PMC pass $n failed"
done
python profiles/summarize.py $OUT gpurun_out/r02/r02   # then locally: MVS_PROFILE_DEPTH_MAPS=4 python profiles/summarize.py gpurun_out/r02 profiles/r02
# the recurrent sweep (config c3) alone: per-kernel time inside a depth map
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gru_stats -- python bench.py --regularization GRU --workload c3 --steps 4 > $OUT/gru_stats.log 2>&1
