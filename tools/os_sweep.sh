#!/bin/bash
# Sweep of the block-kernel tiling variants (conv3d_os.hip) at the metric workload: per-layer in-pipeline times from
# bench.py's roofline_kernels rows.  Run on the GPU box:  gpurun -- 'bash tools/os_sweep.sh > gpurun_out/os_sweep.log'
run() {
  echo "== $*"
  env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c '
import json,sys
for line in sys.stdin:
    if line.startswith("{"):
        d=json.loads(line)
        rows={k["kernel"].split(" ")[0]:k["ms"]*1e3 for k in d["roofline_kernels"] if k["kernel"].startswith("3dconv")}
        print("  %.1f maps/s  chain %.1f us | " % (d["value"], d.get("low_resolution_chain_us",0)) + "  ".join("%s %.1f" % (k[6:],v) for k,v in rows.items()))
'
}
run MVS_X=0
run MVS_NO_OS=1
for l in 31 21 30 40 50; do run MVS_OS_$l=-1; done
run MVS_OS_20=0
run MVS_OS_11=0
run MVS_SIDE_STREAM=1
