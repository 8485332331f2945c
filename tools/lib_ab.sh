#!/bin/bash
# same-box A/B of library builds on the recurrent sweep: tools/lib_ab.sh <variant names...>  (default = the product library)
cd "$GRAFT_REPO_ROOT" || exit 1
rm -f gpurun_out/lib_ab.log
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset MVS_LIB_PATH; else export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/lib_ab.log
  timeout -k 10 200 python tools/gru_time.py --views 1 4 --iters 4 2>&1 | grep "c3 sweep" | sed 's/(host.*//' >> gpurun_out/lib_ab.log
done
done
cat gpurun_out/lib_ab.log
