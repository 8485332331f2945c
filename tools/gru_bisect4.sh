#!/bin/bash
# After the fix (four private streams of one priority class, queues created back to back): the modes that were slow.
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03_gru_bisect_after_fix.log
: > $L
run() { timeout -k 10 240 python tools/gru_bisect.py "$@" 2>&1 | grep -E "^pre=|Error|error" >> $L || echo "FAILED: $*" >> $L; }
for m in none nullcnn streamcnn two streams4 streams8 towers streamcnn,two nullcnn,streamcnn,dominant,layers,two,c2; do run $m; done
run none --caller stream
run streamcnn,two --caller stream
cat $L
