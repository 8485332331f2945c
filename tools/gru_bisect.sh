#!/bin/bash
# Every bisect mode of tools/gru_bisect.py in a process of its own; log -> gpurun_out/r03_gru_bisect.log (copy to profiles/).
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03_gru_bisect.log
: > $L
run() { timeout -k 10 240 python tools/gru_bisect.py "$@" 2>&1 | grep -E "^pre=|Error|error" >> $L || echo "FAILED: $*" >> $L; }
for m in none nullcnn streamcnn two twofree c2 tevents layers dominant streams4 streams8 towers \
         streamcnn,two streamcnn,two,c2 nullcnn,streamcnn,dominant,layers,two,c2; do
    run $m
done
run none --caller stream
run streamcnn,two --caller stream
for q in 1 2 8 16; do
    GPU_MAX_HW_QUEUES=$q run none
    GPU_MAX_HW_QUEUES=$q run streamcnn,two
done
cat $L
