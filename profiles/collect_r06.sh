#!/bin/bash
# Round-6 profile collection; run on the GPU box:  gpurun -- 'bash profiles/collect_r06.sh'
# Kernel-trace/stats and PMC counters are collected in SEPARATE rocprofv3 runs; the program itself follows `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$n -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_$n.log 2>&1 || echo "PMC pass $n failed"
done
python profiles/summarize.py $OUT gpurun_out/r06/r06
# the towers: per-launch counters (in line) -> gpurun_out/r06_unet/table.txt
UNET_SIDE=0 bash tools/r6_unet_counters.sh > $OUT/unet_counters.log 2>&1
cp gpurun_out/r06_unet/table.txt $OUT/r06_unet_counters_after.txt
# the default bench record of the round (what the driver runs)
timeout -k 10 900 python bench.py > $OUT/r06_bench_default.json 2> $OUT/r06_bench_default.err
ls -la $OUT/r06*
