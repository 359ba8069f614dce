"""CPU (cross-compile): instruction mix of every MFMA loop of a translation unit's kernels -- MFMA / VALU / SALU / LDS / VMEM / waitcnt /
s_nop per loop iteration.  The MFMA kernels here are bound by instruction ISSUE as much as by the matrix pipe (the SIMD hides ~5 issue
slots behind a v_mfma_f32_32x32x16, MI355X_MICROARCH.md): non-MFMA instructions per MFMA is the number to drive down.
    python tools/isa_mix.py conv_h2_launch.cpp [substring of the demangled kernel name]"""
import collections, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_check as ic


def mix(lines):
    c = collections.Counter()
    for x in lines:
        t = x.strip()
        if not t or t.startswith((";", ".")) or re.match(r"^\.?L", t):
            continue
        i = t.split()[0]
        c["mfma" if i.startswith("v_mfma") else "valu" if i.startswith("v_") else "waitcnt" if i.startswith("s_waitcnt") else "nop" if i.startswith("s_nop")
          else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem" if i.startswith(("buffer_", "global_", "scratch_")) else "other"] += 1
    return c


def main():
    unit = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    s = open(ic.compile_asm(unit=unit)).read()
    for m in re.finditer(r"^(_ZN5tsnet\w+):\s*;", s, re.M):
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("tsnet::", "").replace("(ConvArgs)", "").replace("void ", "")
        if pat not in d:
            continue
        L = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
        labels = {mm.group(1): k for k, l in enumerate(L) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = []
        for k, l in enumerate(L):
            mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                loops.append((labels[mm.group(1)], k))
        inner = [(a, b) for a, b in loops if not any((c, e) != (a, b) and a <= c and e <= b for c, e in loops)]
        rows = [(mix(L[a:b + 1]), b - a) for a, b in inner]
        rows = [(c, n) for c, n in rows if c["mfma"] >= 8 or (c["vmem"] >= 4 and c["lds"] >= 8)]
        for c, n in sorted(rows, key=lambda t: -t[0]["mfma"]):
            tot = sum(c.values())
            print(f"{d[:70]:70s} loop of {tot:4d}: mfma {c['mfma']:3d} valu {c['valu']:3d} salu {c['salu']:3d} lds {c['lds']:3d} vmem {c['vmem']:3d} waitcnt {c['waitcnt']:2d} nop {c['nop']:2d}"
                  + (f"   non-mfma/mfma {(tot - c['mfma']) / c['mfma']:.2f}" if c["mfma"] else "   (no MFMA: a producer / staging loop)"))


if __name__ == "__main__":
    main()
