"""Per-kernel average durations from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -> DIR/NAME_results.db).
    python tools/kernel_times.py gpurun_out/prof/x_results.db [substring ...]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"
    pats = sys.argv[2:]
    for name, c, avg, tot in db.execute(q):
        if not pats or any(p in name for p in pats):
            print(f"{name[:90]:90s} {c:6d} {avg:10.1f} us  {tot:12.1f} us")


if __name__ == "__main__":
    main()
