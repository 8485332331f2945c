#!/bin/bash
# full GPU suite + default bench (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/full_tests.log
tail -3 gpurun_out/full_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc $?"
tail -c 600 gpurun_out/bench_default.err
