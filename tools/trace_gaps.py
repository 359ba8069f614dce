"""Summarise a rocprofv3 kernel trace (rocpd sqlite): total span, busy time and idle gaps of the LAST `n` dispatches -- how launch-bound a
short forward is.  usage: trace_gaps.py <db> [n]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n:]
span = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum(r[2] - r[1] for r in rows) / 1e3
gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print(f"{len(rows)} dispatches: span {span:.1f} us, busy {busy:.1f} us ({100 * busy / span:.1f} %), positive gaps {len(pos)}: sum {sum(pos):.1f} us, median {sorted(pos)[len(pos) // 2] if pos else 0:.2f} us")
