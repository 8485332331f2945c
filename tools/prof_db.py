"""Per-kernel summary of a rocprofv3 rocpd database: python tools/prof_db.py <results.db> [divisor] [filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
flt = sys.argv[3] if len(sys.argv) > 3 else "%"
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t and "info" in t][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
     "where s.kernel_name like ? group by s.kernel_name order by 4 desc" % (kd, ks))
rows = list(cur.execute(q, (flt,)))
tot = sum(r[3] for r in rows)
for r in rows[:45]:
    print("%-92s calls %5d avg %9.1f us  per-unit %8.1f us" % (r[0][:92], r[1], r[2] / 1e3, r[3] / 1e3 / div))
print("total per unit: %.1f us" % (tot / 1e3 / div))
