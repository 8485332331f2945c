"""Per-kernel time inside ONE iteration of a profiled loop: python tools/prof_step.py <results.db> <marker substring> [which]
The marker is a kernel launched once per iteration (e.g. rmsprop); the window is between marker `which`-1 and `which`
(default: the last two)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2]
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t and "info" in t][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
marks = [i for i, r in enumerate(rows) if marker in r[0]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 1
lo, hi = marks[which - 1], marks[which]
win = rows[lo + 1:hi + 1]
agg = {}
for name, s, e in win:
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += e - s
busy = sum(v[1] for v in agg.values())
print("window %.2f ms wall, %.2f ms of kernel time, %d launches" % ((win[-1][2] - win[0][1]) / 1e6, busy / 1e6, len(win)))
for name, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print("%-100s calls %5d  total %8.1f us  avg %7.1f us" % (name[:100], cnt, t / 1e3, t / 1e3 / cnt))
