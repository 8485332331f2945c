#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in default pg8 pg2 c3same; do
  if [ $v = default ]; then unset MVS_LIB_PATH; else export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$v.so; fi
  echo "== $v" >> gpurun_out/r4_ab1.log
  timeout -k 10 200 python tools/gru_time.py --views 1 4 --iters 4 2>&1 | grep "c3 sweep" >> gpurun_out/r4_ab1.log
done
cat gpurun_out/r4_ab1.log
