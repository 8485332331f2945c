#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 200 python tools/gru_time.py --views 1 4 --iters 4 2>&1 | grep "c3 sweep" > gpurun_out/r4_t1aff.log
cat gpurun_out/r4_t1aff.log
timeout -k 10 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -x -q -m gpu -k "gru or convgru or wta" > gpurun_out/r4_t3.log 2>&1
tail -3 gpurun_out/r4_t3.log
