"""Per-launch table of the UNetDS2GN towers from tools/r6_unet_counters.sh: duration, algorithmic GFLOP and
fraction of the fp32 MFMA peak, matrix-pipe busy share, wait shares, LDS conflict share, HBM bytes against
the layer's algorithmic bytes.  Launches are matched to layers by their order inside the LAST pass."""
import collections
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsnet_amd.feature_net import UNET_LAYERS

O = sys.argv[1]
V, H, W = 5, 512, 640
PEAK_TF, HBM_TBS = 157.3, 8.0


def is_tower(name):
    return ("conv2d_gn" in name or "deconv2d_gn" in name or "conv2d_p_" in name or "unet_" in name) and "layout" not in name


def last_pass(rows, per=None):
    rows = [r for r in rows if is_tower(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r.get("Start_Timestamp", r.get("Dispatch_Id", 0))))
    return rows


# layer shapes
shape = {"data": (H, W, 4)}
layers = []
for name, kind, srcs, k, mult, stride in UNET_LAYERS:
    h, w, _ = shape[srcs[0]]
    cin = sum(shape[s][2] for s in srcs)
    cout = 8 * mult
    if kind == "dg":
        ho, wo = 2 * h, 2 * w
        macs = ho * wo * 2.25 * cin * cout
    else:
        ho, wo = -(-h // stride), -(-w // stride)
        macs = ho * wo * k * k * (3 if srcs == ("data",) else cin) * cout
    shape[name] = (ho, wo, cout)
    byts = (h * w * cin + ho * wo * cout) * 4
    layers.append((name, V * macs * 2 / 1e9, V * byts / 1e6, "%dx%d %d->%d k%d s%d" % (ho, wo, cin, cout, k, stride)))

tr = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = last_pass(list(csv.DictReader(open(tr))))
nl = int(os.environ.get("UNET_LAUNCHES", len(layers)))
passes = len(rows) // nl
rows = rows[-nl:]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t0 = int(rows[0]["Start_Timestamp"])
span = (int(rows[-1]["End_Timestamp"]) - t0) / 1e3

# counters: per dispatch, keyed by order among tower launches
ctr = collections.defaultdict(dict)
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if is_tower(r["Kernel_Name"]):
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-nl:]
    for i, d in enumerate(ids):
        ctr[i].update(per[d])

print("# UNetDS2GN towers, %d views of %dx%d: per launch of the last of %d passes (rocprofv3 kernel trace + separate --pmc passes)" % (V, H, W, passes))
print("# busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x SQ_BUSY_CYCLES-equivalent); shares of wave cycles: wait = SQ_WAIT_ANY, stall = SQ_WAIT_INST_ANY, act = SQ_ACTIVE_INST_ANY")
print("%-3s %-10s %-28s %-34s %7s %7s %6s | %6s %5s %5s %5s %6s | %7s %7s %7s %5s" % (
    "i", "layer", "shape", "kernel", "us", "GFLOP", "frac", "mfma%", "wait", "stall", "act", "ldscf", "alg MB", "rd MB", "wr MB", "x"))
tot_us = tot_gf = 0.0
for i, (r, d) in enumerate(zip(rows, dur)):
    name, gf, mb, shp = layers[i] if nl == len(layers) else ("?", 0.0, 0.0, "")
    c = ctr.get(i, {})
    kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("(Conv2dArgs)", "")[:34]
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (guide): busy share of the kernel = cycles / (1024 SIMDs x duration x clock)
    clk = 2.4e3   # cycles per us (upper bound; the clock drops under load)
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * d * clk) * 100 if d else 0
    rd = c.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6   # KB -> MB, x2 gfx950 wide-read correction (MI355X_MICROARCH.md:298)
    wr = c.get("WRITE_SIZE", 0) * 1024 / 1e6
    cf = c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1)
    print("%-3d %-10s %-28s %-34s %7.1f %7.2f %6.2f | %6.1f %5.2f %5.2f %5.2f %6.2f | %7.1f %7.1f %7.1f %5.2f" % (
        i, name, shp, kn, d, gf, gf / d / PEAK_TF * 1e3 if d else 0, mf, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
        c.get("SQ_ACTIVE_INST_ANY", 0) / wc, cf, mb, rd, wr, (rd + wr) / mb if mb else 0))
    tot_us += d; tot_gf += gf
print("launches %d, sum of durations %.1f us, span %.1f us, %.1f GFLOP = %.2f of fp32 MFMA peak" % (nl, tot_us, span, tot_gf, tot_gf / tot_us / PEAK_TF * 1e3))
