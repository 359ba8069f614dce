"""Tabulate the three rocprofv3 --pmc passes of tools/profile_round.sh.
usage: pmc_table.py <prof_dir>   (expects <prof_dir>/{fetch,write,sq}/**/*.db)
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE under-reports wide coalesced reads by
2x (MI355X_MICROARCH.md, HBM section): fetch_MB_corrected = 2 x raw.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); the wait/active columns are fractions of SQ_WAVE_CYCLES.
clock_GHz = (GRBM_GUI_ACTIVE / 8 XCDs) / the kernel's average duration in the kernel-trace pass of the same command (<prof_dir>/trace):
the effective shader clock while the kernel runs (DVFS: MI355X_MICROARCH.md, "DVFS give-back").  Printed only for kernels of >= 30 us: below
that GRBM_GUI_ACTIVE also counts the dispatch's ramp and drain and the quotient is meaningless (3 - 7 "GHz" on 5 us kernels: VERDICT r4)."""
import collections, glob, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from src_digest import digest

def load(dbdir):
    dbs = glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True)
    if not dbs: return {}
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [d[1] for d in cur.execute("pragma table_info('counters_collection')")]
    ci = {c: i for i, c in enumerate(cols)}
    namecol = "kernel_name" if "kernel_name" in ci else [c for c in cols if "name" in c][0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in cur.execute("select * from counters_collection"):
        agg[short(r[ci[namecol]])][r[ci["counter_name"]]].append(r[ci["value"]])
    return agg

def short(n):
    m = re.search(r"(conv_h2[a-z]?|conv_g64|conv_w1_one|conv_w1)_kernel(?:<([^>]*)>)?", n)
    if m: return m.group(1) + ("<" + m.group(2).replace(" ", "").replace("false", "f").replace("true", "t") + ">" if m.group(2) else "")
    return re.sub(r"\(.*", "", n).split("::")[-1][:34]

def durations(dbdir):
    """kernel -> average duration in ns from the kernel-trace pass"""
    dbs = glob.glob(os.path.join(dbdir, "**", "*.db"), recursive=True)
    if not dbs: return {}
    agg = collections.defaultdict(list)
    try:
        for name, dur in sqlite3.connect(dbs[0]).cursor().execute("select name, duration from kernels"):
            agg[short(name)].append(dur)
    except sqlite3.Error:
        return {}
    return {k: sum(v) / len(v) for k, v in agg.items()}

d = sys.argv[1]
fetch, write, sq = load(os.path.join(d, "fetch")), load(os.path.join(d, "write")), load(os.path.join(d, "sq"))
sq2 = load(os.path.join(d, "sq2"))
dur = durations(os.path.join(d, "trace"))
print("# source_digest: %s" % digest())
avg = lambda v: sum(v) / len(v) if v else 0.0
order = sorted(fetch, key=lambda k: -sum(fetch[k].get("FETCH_SIZE", [0])))
print("# per launch; bytes from separate --pmc FETCH_SIZE / WRITE_SIZE passes")
print("%-36s %8s %14s %20s %12s" % ("kernel", "launches", "fetch_MB(raw)", "fetch_MB_corrected", "write_MB"))
for k in order:
    f = avg(fetch[k].get("FETCH_SIZE", [])) / 1024.0
    w = avg(write.get(k, {}).get("WRITE_SIZE", [])) / 1024.0
    print("%-36s %8d %14.2f %20.2f %12.2f" % (k, len(fetch[k].get("FETCH_SIZE", [])), f, 2 * f, w))
print()
print("%-36s %8s %10s %10s %10s %10s %12s %10s %10s" % ("kernel", "launches", "mfma_util", "wait_any", "wait_inst", "active", "lds_conflict", "trace_us", "clock_GHz"))
for k in sorted(sq, key=lambda k: -sum(sq[k].get("GRBM_GUI_ACTIVE", [0]))):
    c = sq[k]; g = avg(c.get("GRBM_GUI_ACTIVE", [])); wc = avg(c.get("SQ_WAVE_CYCLES", [])) or 1.0
    if g <= 0: continue
    t = dur.get(k, 0.0)
    print("%-36s %8d %10.3f %10.3f %10.3f %10.3f %12.0f %10.1f %10s" % (k, len(c.get("GRBM_GUI_ACTIVE", [])), avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / (1024.0 * g / 8.0),
          avg(c.get("SQ_WAIT_ANY", [])) / wc, avg(c.get("SQ_WAIT_INST_ANY", [])) / wc, avg(c.get("SQ_ACTIVE_INST_ANY", [])) / wc, avg(c.get("SQ_LDS_BANK_CONFLICT", [])),
          t / 1e3, ("%.2f" % ((g / 8.0) / t)) if t >= 30e3 else "-"))

if sq2:
    # instruction mix per launch (the second SQ pass): conv_w1's K loop is bound by instruction ISSUE -- the SIMD hides ~5 issue slots behind an
    # MFMA (MI355X_MICROARCH.md) -- so the non-MFMA instructions per MFMA are the figure to watch, not only mfma_util
    print()
    print("%-36s %8s %12s %12s %12s %12s %12s %14s" % ("kernel", "launches", "insts_mfma", "insts_valu", "insts_salu", "insts_lds", "insts_vmem", "non-mfma/mfma"))
    for k in sorted(sq2, key=lambda k: -sum(sq2[k].get("SQ_INSTS_MFMA", [0]))):
        c = sq2[k]; m = avg(c.get("SQ_INSTS_MFMA", []))
        if m <= 0: continue
        v = avg(sq.get(k, {}).get("SQ_INSTS_VALU", [])); sa = avg(c.get("SQ_INSTS_SALU", [])); l = avg(c.get("SQ_INSTS_LDS", [])); vm = avg(c.get("SQ_INSTS_VMEM", []))
        print("%-36s %8d %12.0f %12.0f %12.0f %12.0f %12.0f %14.2f" % (k, len(c.get("SQ_INSTS_MFMA", [])), m, v - m if v > m else v, sa, l, vm, ((v - m if v > m else v) + sa + l + vm) / m))
