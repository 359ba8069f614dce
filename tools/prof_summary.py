"""Summarise a rocprofv3 rocpd sqlite db: per-kernel stats and, for the last forward in the trace,
each conv dispatch in launch order (grid, duration).  usage: prof_summary.py <db> [launches_per_forward]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count from kernels order by start").fetchall()
def short(n):
    m = re.search(r"(conv_h2[a-z]?|conv_g64|conv_w1_one|conv_w1)_kernel<([^>]*)>", n)
    if m: return "%s<%s>" % (m.group(1), m.group(2).replace(" ", "").replace("false", "f").replace("true", "t"))
    n = re.sub(r"\(.*", "", n); return n.split("::")[-1][:40]
agg = {}
for r in rows:
    a = agg.setdefault(short(r[0]), [0, 0.0]); a[0] += 1; a[1] += r[3]
tot = sum(v[1] for v in agg.values())
print("%-36s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-36s %8d %12.1f %10.2f %6.2f" % (k, v[0], v[1] / 1e3, v[1] / 1e3 / v[0], 100 * v[1] / tot))
# last forward = the dispatches after the second-to-last head kernel up to the last one (every forward ends with the head)
ALL = "--all" in sys.argv
idx = [i for i, r in enumerate(rows) if "head_conv" in r[0] and "pack" not in r[0]]
if len(idx) >= 2:
    fw = rows[idx[-2] + 1: idx[-1] + 1]
    print("\nlast forward: %d dispatches, span %.3f ms, busy %.3f ms" % (len(fw), (fw[-1][2] - fw[0][1]) / 1e6, sum(r[3] for r in fw) / 1e6))
    for r in fw:
        if ALL or ("conv_" in r[0] and "pack" not in r[0]):
            print("  t=%8.1f  %-34s grid=%6d wg=%d lds=%d vgpr=%d agpr=%d  %9.1f us" % ((r[1] - fw[0][1]) / 1e3, short(r[0]), r[4] // r[5], r[5], r[6], r[7], r[8], r[3] / 1e3))
