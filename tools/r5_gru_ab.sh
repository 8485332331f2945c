#!/bin/bash
# round 5: parity of the recurrent sweep with the working tree's library, then same-box timing against mvsnet_amd/variants/lib_<base>.so
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gru or wta" > gpurun_out/gru_ab_tests.log 2>&1 || { tail -30 gpurun_out/gru_ab_tests.log; exit 1; }
tail -2 gpurun_out/gru_ab_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "gru" > gpurun_out/gru_ab_tests_full.log 2>&1 || { tail -30 gpurun_out/gru_ab_tests_full.log; exit 1; }
tail -2 gpurun_out/gru_ab_tests_full.log
for rep in 1 2 3; do
  for L in ${1:-g0} new; do
    if [ $L = new ]; then unset MVS_LIB_PATH; else export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$L.so; fi
    timeout -k 10 200 python tools/gru_time.py --views 1 4 --iters 4 2>&1 | grep "c3 sweep" | sed "s/^/$L: /" | cut -c1-110
  done
done
