#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python tools/prof_session.py > gpurun_out/r4_prof_session1.log 2>&1
tail -42 gpurun_out/r4_prof_session1.log
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r4_t4.log 2>&1
tail -4 gpurun_out/r4_t4.log
