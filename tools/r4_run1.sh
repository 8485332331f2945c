#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_full_size.py -x -q -m gpu -s > gpurun_out/r4_t2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4_t2.log
tail -5 gpurun_out/r4_t2.log
