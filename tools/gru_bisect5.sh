#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03_gru_bisect5.log
: > $L
run() { timeout -k 10 240 python tools/gru_bisect.py "$@" 2>&1 | grep -E "^pre=|Error|error" >> $L || echo "FAILED: $*" >> $L; }
run none; run streamcnn; run two
run none --iters 1; run streamcnn --iters 1
cat $L
