#!/bin/bash
# round 5: first run of the fused sweep -- parity subset, then timing (one call, stops at the first failure)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gru" > gpurun_out/r5_gru_parity.log 2>&1 || { tail -40 gpurun_out/r5_gru_parity.log; exit 1; }
tail -3 gpurun_out/r5_gru_parity.log
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -s -k "gru" > gpurun_out/r5_gru_full.log 2>&1 || { tail -60 gpurun_out/r5_gru_full.log; exit 1; }
grep -E "plane agreement|margin|differing|passed|failed|wavefront" gpurun_out/r5_gru_full.log
timeout -k 10 300 python tools/gru_time.py --views 1 2 4 --iters 5 > gpurun_out/r5_gru_time_fused.log 2>&1 || { tail -20 gpurun_out/r5_gru_time_fused.log; exit 1; }
cat gpurun_out/r5_gru_time_fused.log
timeout -k 10 300 python tools/gru_time.py --views 1 4 --iters 5 --form 2 > gpurun_out/r5_gru_time_wave.log 2>&1
cat gpurun_out/r5_gru_time_wave.log
