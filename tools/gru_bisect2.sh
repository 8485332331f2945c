#!/bin/bash
# NOTE (round 6): MVS_GRU_ONE_STREAM is now mvs_set_test_hook(MVS_HOOK_GRU_ONE_STREAM); kept as the record of the round-3 bisect
# Second bisect pass: which single kind of work on a side stream makes the null-stream sweep slow, and what the runtime logs.
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03_gru_bisect2.log
: > $L
run() { timeout -k 10 240 python tools/gru_bisect.py "$@" 2>&1 | grep -E "^pre=|^torch stream|Error|error" >> $L || echo "FAILED: $*" >> $L; }
for m in flags homog cost soft memset torchbig; do run $m; done
MVS_GRU_ONE_STREAM=1 run none
MVS_GRU_ONE_STREAM=1 run streamcnn
echo "--- runtime log message histogram, one sweep each (AMD_LOG_LEVEL=4)" >> $L
for m in none streamcnn; do
  AMD_LOG_LEVEL=4 timeout -k 10 300 python tools/gru_bisect.py $m --iters 1 2> /tmp/amdlog_$m.txt | grep -E "^pre=" >> $L
  echo "== $m: $(wc -l < /tmp/amdlog_$m.txt) log lines; the last sweep's tail (300k lines):" >> $L
  tail -n 300000 /tmp/amdlog_$m.txt | sed -E 's/^:[0-9]+:[^:]*:[0-9 ]+: *[0-9]+ *us: *(\[[^]]*\])?//; s/0x[0-9a-f]+/X/g; s/[0-9]+/N/g' | sort | uniq -c | sort -rn | head -40 >> $L
done
cat $L
