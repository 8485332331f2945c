#!/bin/bash
# round 5: the one-block warp + variance sweep -- parity tests, then same-box A/B against mvsnet_amd/variants/lib_base.so
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lab.py -x -q -m gpu -k "cost_volume or warp" > gpurun_out/cv_tests.log 2>&1 || { tail -30 gpurun_out/cv_tests.log; exit 1; }
tail -3 gpurun_out/cv_tests.log
for rep in 1 2 3; do
  for L in base new; do
    if [ $L = base ]; then export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_base.so; else unset MVS_LIB_PATH; fi
    timeout -k 10 120 python tools/cv_time.py M --iters 30 2>&1 | tail -1 | sed "s/^/$L: /"
  done
done
