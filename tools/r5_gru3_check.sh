#!/bin/bash
# round 5: the three-group fused ConvGRU kernel (MVS_GRU_THREE_GROUPS=1): parity, then timing against the default and the two-group kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export MVS_GRU_THREE_GROUPS=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gru or wta" > gpurun_out/gru3_tests.log 2>&1 || { tail -30 gpurun_out/gru3_tests.log; exit 1; }
tail -2 gpurun_out/gru3_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "gru" > gpurun_out/gru3_tests_full.log 2>&1 || { tail -30 gpurun_out/gru3_tests_full.log; exit 1; }
tail -2 gpurun_out/gru3_tests_full.log
unset MVS_GRU_THREE_GROUPS
for rep in 1 2; do
  for L in one two three; do
    unset MVS_GRU_TWO_GROUPS MVS_GRU_THREE_GROUPS
    if [ $L = two ]; then export MVS_GRU_TWO_GROUPS=1; fi
    if [ $L = three ]; then export MVS_GRU_THREE_GROUPS=1; fi
    timeout -k 10 200 python tools/gru_time.py --views 1 4 --iters 4 2>&1 | grep "c3 sweep" | sed "s/^/$L group(s): /" | cut -c1-125
  done
done
