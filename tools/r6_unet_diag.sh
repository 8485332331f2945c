#!/bin/bash
# Round 6: timing-only diagnosis builds of the tower kernels (tools/r6_unet_diag.patch, -DDIAG_*: results wrong on purpose),
# per-launch durations side by side.  usage: bash tools/r6_unet_diag.sh default dNOGN dNOSTAT hook:unet_persistent=0 hook:unet_grid=768 ...
# (a name = mvsnet_amd/variants/lib_<name>.so; hook:<name>=<value>[,...] = the product library with test hooks set)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cat > /tmp/unet_only.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mvsnet_amd import synthetic as S
from mvsnet_amd.feature_net_hip import HipUNetDS2GN
dev = torch.device("cuda", 0)
from mvsnet_amd import _lib
for kv in filter(None, os.environ.get("UNET_HOOKS", "").split(",")):      # e.g. UNET_HOOKS=unet_persistent=0,unet_grid=768
    k, v = kv.split("="); _lib.set_test_hook(k, int(v))
net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev, **({"side_streams": int(os.environ["UNET_SIDE"])} if "UNET_SIDE" in os.environ else {}))
img = torch.randn(5, 512, 640, 3, device=dev)
for _ in range(6):
    out = net(img)
torch.cuda.synchronize()
PY
for v in "$@"; do
  unset MVS_LIB_PATH UNET_HOOKS UNET_SIDE
  case $v in
    default) ;;
    side:*) export UNET_SIDE=${v#side:} ;;
    hook:*) export UNET_HOOKS=${v#hook:} ;;
    *) export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$v.so ;;
  esac
  O=gpurun_out/r06_diag/$(echo $v | tr ":=," "___"); rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -- python /tmp/unet_only.py > $O.log 2>&1 || { echo "$v failed"; tail -5 $O.log; }
done
python - "$@" <<'PY'
import csv, glob, sys
names = sys.argv[1:]
cols = {}
spans = {}
for v in names:
    f = glob.glob("gpurun_out/r06_diag/%s/**/*kernel_trace.csv" % v.replace(":", "_").replace("=", "_").replace(",", "_"), recursive=True)
    if not f: continue
    rows = [r for r in csv.DictReader(open(f[0])) if ("conv2d_gn" in r["Kernel_Name"] or "conv2d_p_" in r["Kernel_Name"] or "unet_" in r["Kernel_Name"]) and "layout" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = 32                                  # launches of one pass (the last one is taken)
    cols[v] = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]) for r in rows[-per:]]
    spans[v] = (max(int(r["End_Timestamp"]) for r in rows[-per:]) - min(int(r["Start_Timestamp"]) for r in rows[-per:])) / 1e3
print("%-3s %-44s" % ("i", "kernel") + "".join("%10s" % v[-10:] for v in cols))
n = max(len(c) for c in cols.values())
for i in range(n):
    k = next(iter(cols.values()))
    kn = k[i][1].replace("(anonymous namespace)::", "").replace("void ", "").replace("(Conv2dArgs)", "")[:44] if i < len(k) else ""
    print("%-3d %-44s" % (i, kn) + "".join("%10.1f" % (c[i][0] if i < len(c) else 0) for c in cols.values()))
print("%-48s" % "sum" + "".join("%10.1f" % sum(x[0] for x in c) for c in cols.values()))
print("%-48s" % "span (first start -> last end)" + "".join("%10.1f" % spans[v] for v in cols))
PY
