"""Print per-kernel averages of PMC counters from a rocprofv3 rocpd sqlite db. usage: pmc_summary.py <db> [name-filter]"""
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
rows = cur.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
namecol = "kernel_name" if "kernel_name" in ci else [c for c in cols if "name" in c][0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    kn = r[ci[namecol]]
    if flt and flt not in kn: continue
    short = re.sub(r"\(.*", "", kn)[:70]
    agg[short][r[ci["counter_name"]]].append(r[ci["value"]])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%3d avg=%.4g" % (c, len(v), sum(v) / len(v)))
print("columns:", cols)
