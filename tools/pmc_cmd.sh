#!/bin/bash
# usage (on the GPU box): bash tools/pmc_cmd.sh <tag> <kernel-substring> "<counters pass 1>" ... -- <python args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; pat=$2; shift 2
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
mkdir -p gpurun_out/$tag
i=0
for c in "${passes[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/$tag/p$i -- python "$@" > gpurun_out/$tag/p$i.log 2>&1 || echo "pass $i failed"
done
python - "$tag" "$pat" <<'PY'
import csv, glob, collections, sys
tag, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/%s/p*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if pat not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
