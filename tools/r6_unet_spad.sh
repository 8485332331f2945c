#!/bin/bash
# Round 6: LDS bank-conflict share and duration of the persistent tower launches for several slab pitches (variants built with
# tools/build_variant.sh spad<N> unet2d_p.hip after editing `S = CIN + 4` in PGeom to CIN + N: pitch = Cin + N floats).   gpurun -- 'bash tools/r6_unet_spad.sh'
cat > /tmp/unet_only.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mvsnet_amd import synthetic as S
from mvsnet_amd.feature_net_hip import HipUNetDS2GN
dev = torch.device("cuda", 0)
net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev, side_streams=0)
img = torch.randn(5, 512, 640, 3, device=dev)
for _ in range(3):
    out = net(img)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in default spad8 spad12 spad20; do
  if [ $v = default ]; then unset MVS_LIB_PATH; else export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$v.so; fi
  O=gpurun_out/r06_spad/$v; rm -rf $O; mkdir -p $O
  UNET_SIDE=0 UNET_PASSES=3 timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -- python /tmp/unet_only.py > $O.log 2>&1
  python - $O $v <<'PY'
import csv, glob, sys, collections
O, v = sys.argv[1], sys.argv[2]
f = glob.glob(O + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(dict)
name = {}
for r in csv.DictReader(open(f)):
    if "conv2d_p_" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"]); name[int(r["Dispatch_Id"])] = r["Kernel_Name"]
ids = sorted(per)[-13:]
tr = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {int(r["Dispatch_Id"]): (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr))}
print(v, " ".join("%s:%.2f/%.0fus" % (name[i].split("<")[1][:14].replace(" ", ""), per[i].get("SQ_LDS_BANK_CONFLICT", 0) / max(per[i].get("SQ_LDS_IDX_ACTIVE", 1), 1), dur.get(i, 0)) for i in ids))
PY
done
