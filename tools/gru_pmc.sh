#!/bin/bash
# SQ counters of the recurrent sweep at 4 views per launch, summed per kernel over one process (2 warm-up + 2 timed sweeps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/gru_pmc; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- python tools/gru_time.py --views 4 --iters 2 > $O/p1.log 2>&1 || echo "pass failed"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/gru_pmc/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
sweeps = 4.0
tot = collections.defaultdict(float)
lines = []
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_ACTIVE_INST_ANY", 0)):
    if n[k] < 50: continue
    # per-SIMD microseconds at 2.4 GHz if the instruction-active time of the waves of a SIMD did not overlap: quad-cycles * 4 / 1024 SIMDs / 2400
    us = lambda c: d.get(c, 0) * 4 / 1024 / 2400 / sweeps / 256     # per plane (256 planes per sweep)
    lines.append("%-60s launches %6d | per plane: active_any %6.1f us  valu %6.1f us  mfma_busy %6.1f us | valu instr/plane %7.0f k  wait_any %6.1f  wait_inst %6.1f" % (
        k, n[k], us("SQ_ACTIVE_INST_ANY"), us("SQ_ACTIVE_INST_VALU"), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / 2400 / sweeps / 256,
        d.get("SQ_INSTS_VALU", 0) / sweeps / 256 / 1e3, us("SQ_WAIT_ANY"), us("SQ_WAIT_INST_ANY")))
    for c in d: tot[c] += d[c]
print("\n".join(lines))
print("TOTAL per plane: active_any %.1f us, valu %.1f us, mfma_busy %.1f us (sum over kernels, per SIMD, 2.4 GHz)" % (
    tot["SQ_ACTIVE_INST_ANY"] * 4 / 1024 / 2400 / sweeps / 256, tot["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / 2400 / sweeps / 256, tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2400 / sweeps / 256))
PY
grep "c3 sweep" $O/p1.log
