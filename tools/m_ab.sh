#!/bin/bash
# same-box A/B of library builds on the metric workload: tools/m_ab.sh <variant names...>  (default = the product library)
# three repetitions of `bench.py --no-extra --no-cpu-baseline --steps 200`; prints depth maps/s and the per-layer microseconds
cd "$GRAFT_REPO_ROOT" || exit 1
rm -f gpurun_out/m_ab.log
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = default ]; then unset MVS_LIB_PATH; else export MVS_LIB_PATH=$PWD/mvsnet_amd/variants/lib_$v.so; fi
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/m_ab_line.json || exit 1
  python - "$v" "$rep" <<'PY' >> gpurun_out/m_ab.log
import json, sys
d = json.load(open("gpurun_out/m_ab_line.json"))
rows = {r["kernel"].split(" (")[0]: r["ms"] * 1e3 for r in d.get("roofline_kernels", []) if r["kernel"].startswith("3dconv")}
print("%-10s rep %s: %.1f depth maps/s  %.4f ms  | " % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"]) +
      "  ".join("%s %.1f" % (k.replace("3dconv", ""), v) for k, v in rows.items()))
PY
done
done
cat gpurun_out/m_ab.log
