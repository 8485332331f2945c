#!/bin/bash
# Round 6: warp + variance at workload c2 (288 x 216 feature maps) against M: duration, HBM bytes (FETCH_SIZE x 2, WRITE_SIZE), texture-path
# and vector counters -- why the launch costs 1.19 x its voxel ratio at c2.   gpurun -- 'bash tools/r6_cv_c2.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for wl in M c2; do
  O=gpurun_out/r06_cv_$wl; rm -rf $O; mkdir -p $O
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
           "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -- python tools/cv_time.py $wl --iters 5 > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
  done
  python - $wl <<'PY'
import csv, glob, collections, sys
wl = sys.argv[1]
acc = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/r06_cv_%s/p*/**/*counter_collection.csv" % wl, recursive=True):
    for r in csv.DictReader(open(f)):
        if "cost_volume_sweep" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r06_cv_%s/p1/**/*kernel_trace.csv" % wl, recursive=True):
    for r in csv.DictReader(open(f)):
        if "cost_volume_sweep" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("== %s: %.1f us under the profiler; read %.1f MB (FETCH_SIZE x 2), written %.1f MB" % (wl, sum(dur[-5:]) / 5, m.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6, m.get("WRITE_SIZE", 0) * 1024 / 1e6))
for k in sorted(m):
    print("   %-32s %.4g" % (k, m[k]))
PY
done
