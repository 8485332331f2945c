#!/bin/bash
# rocprofv3 kernel trace of the UNetDS2GN towers on the HIP library (5 views of 512x640): per-launch durations in layer order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/unetprof${TAG:-}; rm -rf $O; export UNETPROF_DIR=$O
cat > /tmp/unet_only.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mvsnet_amd import synthetic as S
from mvsnet_amd.feature_net_hip import HipUNetDS2GN
dev = torch.device("cuda", 0)
net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev)
img = torch.randn(5, 512, 640, 3, device=dev)
for _ in range(6):
    out = net(img)
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python /tmp/unet_only.py > gpurun_out/unetprof${TAG:-}.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob(__import__("os").environ.get("UNETPROF_DIR","gpurun_out/unetprof") + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "conv2d_gn_kernel" in r["Kernel_Name"] or "deconv2d" in r["Kernel_Name"] and "layout" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // 6
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for i, r in enumerate(last):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
    print("%2d %-48s grid %-8s lds %-6s start %8.1f us  dur %6.1f us" % (i, n, r.get("Grid_Size_X", r.get("Grid_Size", "?")) , r.get("LDS_Block_Size", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, d))
print("launches %d, sum of durations %.1f us, span %.1f us" % (per, tot, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
PY
