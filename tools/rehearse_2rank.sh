#!/bin/bash
# Two-rank rehearsal of the multi-process paths on a ONE-GPU box (both ranks share cuda:0; gloo instead of RCCL, which
# needs one GPU per rank).  The real 1/2/4/8-GPU curve is the driver's to measure on an 8-GPU node; this only proves
# that the sharded entry points run, bind their device and agree on the rendezvous.
#   gpurun -- 'bash tools/rehearse_2rank.sh > gpurun_out/rehearse_2rank.log 2>&1'
set -e
export MVS_DIST_BACKEND=gloo MVS_ALLOW_SHARED_GPU=1
echo "== bench.py --gpus 2 (gloo, shared GPU)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra | grep '^{' | python -c '
import json,sys
d=json.loads(sys.stdin.read()); print("value", d["value"], "n_gpus", d["n_gpus"], "per rank", d["per_rank_depth_maps_per_s"])'
echo "== mvsnet_amd.inference sharded by reference view over 2 ranks"
SESS=$(mktemp -d)/sess
python - "$SESS" <<'PY'
import sys
sys.path.insert(0, ".")
from tests._helpers import make_session
make_session(sys.argv[1], n_images=4, h=96, w=128)
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
    -m mvsnet_amd.inference --input_dir "$SESS" --view_num 3 --max_d 8 --width 64 --height 64 --base_image_size 8 2>&1 | grep -i "finished\|error" | tail -4
ls "$SESS/depths_mvsnet" | grep -c "_init.pfm"
