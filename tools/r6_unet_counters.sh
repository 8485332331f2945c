#!/bin/bash
# Round 6: per-LAUNCH time and counters of the UNetDS2GN towers (5 views of 512x640), in layer order.
# One kernel-trace pass and separate --pmc passes (the pool refuses --pmc with other trace domains).
#   gpurun -- 'bash tools/r6_unet_counters.sh'      (TAG=_x MVS_LIB_PATH=... for variants)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_unet${TAG:-}; rm -rf $O; mkdir -p $O
cat > /tmp/unet_only.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from mvsnet_amd import synthetic as S
from mvsnet_amd.feature_net_hip import HipUNetDS2GN
dev = torch.device("cuda", 0)
net = HipUNetDS2GN(S.make_unet_params("normal", seed=3), dev, **({"side_streams": int(os.environ["UNET_SIDE"])} if "UNET_SIDE" in os.environ else {}))
img = torch.randn(5, 512, 640, 3, device=dev)
for _ in range(int(os.environ.get("UNET_PASSES", "6"))):
    out = net(img)
torch.cuda.synchronize()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python /tmp/unet_only.py > $O/trace.log 2>&1 || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
export UNET_PASSES=3
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -- python /tmp/unet_only.py > $O/p$i.log 2>&1 || echo "pmc pass $i failed"
done
python tools/r6_unet_table.py $O > $O/table.txt 2>&1
cat $O/table.txt
