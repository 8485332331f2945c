#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== new" > gpurun_out/r4_cv.log
timeout -k 10 200 python tools/cv_time.py M >> gpurun_out/r4_cv.log 2>&1
timeout -k 10 200 python tools/cv_time.py c2 >> gpurun_out/r4_cv.log 2>&1
echo "== old" >> gpurun_out/r4_cv.log
MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_cvold.so timeout -k 10 200 python tools/cv_time.py M >> gpurun_out/r4_cv.log 2>&1
MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_cvold.so timeout -k 10 200 python tools/cv_time.py c2 >> gpurun_out/r4_cv.log 2>&1
cat gpurun_out/r4_cv.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "cost_volume or 3dcnn or warp or gru_sweep" > gpurun_out/r4_t5.log 2>&1
tail -3 gpurun_out/r4_t5.log
