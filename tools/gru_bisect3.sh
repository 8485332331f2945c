#!/bin/bash
# Third pass: which hardware queue every stream of the sweep is bound to, fast case against slow cases (AMD_LOG_LEVEL=4).
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r03_gru_bisect3.log
: > $L
for m in none streamcnn two streams4; do
  AMD_LOG_LEVEL=4 timeout -k 10 300 python tools/gru_bisect.py $m --iters 1 2> /tmp/amdlog_$m.txt | grep -E "^pre=" >> $L
  echo "== $m: software queue -> hardware queue of every dispatch / barrier packet of the run (count SWq HWq id type)" >> $L
  grep -oE "SWq=0x[0-9a-f]+, HWq=0x[0-9a-f]+, id=[0-9]+, (Dispatch|BarrierValue|BarrierAND)" /tmp/amdlog_$m.txt | sort | uniq -c | sort -k2,2 -k5,5 >> $L
  echo "== $m: queue creation lines" >> $L
  grep -iE "acquire.*queue|created.*queue|hsa_queue_create|queue.*priority|Selected queue" /tmp/amdlog_$m.txt | sed -E 's/^:[0-9]+:[^:]*:[0-9 ]+: *[0-9]+ *us: *//' | sort | uniq -c | sort -rn | head -30 >> $L
done
cat $L
