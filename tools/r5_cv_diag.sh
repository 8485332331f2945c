#!/bin/bash
# which unit bounds the warp + variance sweep: timing-only builds (wrong results) without tap reloads / with half the blend
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for L in default cvnoreload cvhalfblend; do
  if [ $L = default ]; then unset MVS_LIB_PATH; else export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$L.so; fi
  rm -rf gpurun_out/cvd_$L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cvd_$L -- python tools/cv_time.py M --iters 30 > gpurun_out/cvd_$L.log 2>&1 || { echo "$L failed"; tail -3 gpurun_out/cvd_$L.log; }
  f=$(find gpurun_out/cvd_$L -name "*kernel_stats.csv" | head -1)
  echo "$L: $(grep cost_volume_sweep $f | cut -d, -f1-6 | cut -c1-40,90-200)"
done
