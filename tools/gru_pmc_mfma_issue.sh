#!/bin/bash
# How much of SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the 4-view recurrent sweep is the ISSUE of matrix instructions (they are
# vector-ALU-class instructions: one issue quad-cycle each, while the matrix pipe is busy for 8 or 32 clocks behind them)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/gru_pmc2; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/p1 -- python tools/gru_time.py --views 4 --iters 2 ${GRU_FORM:+--form $GRU_FORM} > $O/p1.log 2>&1 || echo "pass failed: $(tail -2 $O/p1.log)"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/gru_pmc2/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
sweeps, planes = 4.0, 256
tot = collections.defaultdict(float)
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if n[k] < 50: continue
    us = lambda x: x * 4 / 1024 / 2400 / sweeps / planes
    print("%-50s per plane: vector-ALU-class instructions %7.0f k, of them matrix %7.0f k | active %5.1f us = matrix issue %5.1f us + other vector %5.1f us | matrix pipe busy %5.1f us" % (
        k, d["SQ_INSTS_VALU"] / sweeps / planes / 1e3, d.get("SQ_INSTS_MFMA", 0) / sweeps / planes / 1e3, us(d["SQ_ACTIVE_INST_VALU"]),
        us(d.get("SQ_INSTS_MFMA", 0)), us(d["SQ_ACTIVE_INST_VALU"] - d.get("SQ_INSTS_MFMA", 0)), d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2400 / sweeps / planes))
    for c in d: tot[c] += d[c]
us = lambda x: x * 4 / 1024 / 2400 / sweeps / planes
print("TOTAL per plane: vector active %.1f us = matrix-instruction issue %.1f us + other vector instructions %.1f us; matrix pipe busy %.1f us" % (
    us(tot["SQ_ACTIVE_INST_VALU"]), us(tot["SQ_INSTS_MFMA"]), us(tot["SQ_ACTIVE_INST_VALU"] - tot["SQ_INSTS_MFMA"]), tot["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2400 / sweeps / planes))
PY
