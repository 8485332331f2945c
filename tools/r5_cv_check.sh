#!/bin/bash
# round 5: the warp + variance sweep -- parity tests, then same-box A/B of the whole depth map against mvsnet_amd/variants/lib_<base>.so
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lab.py -x -q -m gpu -k "cost_volume or warp" > gpurun_out/cv_tests.log 2>&1 || { tail -30 gpurun_out/cv_tests.log; exit 1; }
tail -3 gpurun_out/cv_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/cv_tests_full.log 2>&1 || { tail -30 gpurun_out/cv_tests_full.log; exit 1; }
tail -3 gpurun_out/cv_tests_full.log
bash tools/m_ab.sh ${1:-cv1} default 2>&1 | cut -c1-60
