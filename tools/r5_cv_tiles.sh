#!/bin/bash
# NOTE (round 6): MVS_CV_TILE_ROWS_LOG2 is no longer read from the environment; the shape is forced through mvs_set_test_hook(MVS_HOOK_CV_TILE_ROWS_LOG2) -- kept as the record of the round-5 measurement
# round 5: wave tile shape of the warp + variance sweep (rows = 1, 2, 4, 8 of the wave's 8 pixels): parity tests at every shape, then
# the whole depth map, same box, three repetitions
cd "$GRAFT_REPO_ROOT" || exit 1
for t in 3 2 0 -1; do
  MVS_CV_TILE_ROWS_LOG2=$t timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lab.py -x -q -m gpu -k "cost_volume or warp" > gpurun_out/cv_tests_$t.log 2>&1 || { tail -30 gpurun_out/cv_tests_$t.log; exit 1; }
  echo "rows_log2=$t: $(tail -1 gpurun_out/cv_tests_$t.log)"
done
for rep in 1 2 3; do
  for t in 0 3 -1; do
    MVS_CV_TILE_ROWS_LOG2=$t timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/cvt_line.json || exit 1
    python -c "
import json; d = json.load(open('gpurun_out/cvt_line.json'))
w = [r for r in d.get('roofline_kernels', []) if 'warp' in r['kernel']]
print('rows_log2=$t rep $rep: %.1f depth maps/s  %.4f ms | warp+variance %s us' % (d['value'], d['ms_per_step'], ', '.join('%.1f' % (r['ms'] * 1e3) for r in w)))"
  done
done
