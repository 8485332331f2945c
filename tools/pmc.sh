#!/bin/bash
# usage (on the GPU box): bash tools/pmc.sh <tag> <layer> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; layer=$2; shift 2
mkdir -p gpurun_out/$tag
i=0
for c in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$tag/p$i -- python tools/bench_layer.py $layer --iters 3 > gpurun_out/$tag/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$tag/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "conv" not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %.4g" % (c, sum(v) / len(v)))
PY
