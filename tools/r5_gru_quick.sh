#!/bin/bash
# round 5: quick loop for the fused sweep -- toy / ragged parity, the c3 fixture, batch equality at the mid size, then timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gru" > gpurun_out/r5_gru_parity.log 2>&1 || { tail -40 gpurun_out/r5_gru_parity.log; exit 1; }
tail -1 gpurun_out/r5_gru_parity.log
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -s -k "gru_sweep_matches_the_fixture and False or mid or inverse" > gpurun_out/r5_gru_full.log 2>&1 || { tail -60 gpurun_out/r5_gru_full.log; exit 1; }
grep -E "plane agreement|margin|differing|passed|failed|wavefront" gpurun_out/r5_gru_full.log
timeout -k 10 300 python tools/gru_time.py --views 1 4 --iters 5 ${GRU_TIME_ARGS} > gpurun_out/r5_gru_time_fused.log 2>&1 || { tail -20 gpurun_out/r5_gru_time_fused.log; exit 1; }
grep "c3 sweep" gpurun_out/r5_gru_time_fused.log
