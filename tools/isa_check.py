"""Compile the patch-convolution kernels for gfx950 with -save-temps and report, per kernel, the VGPR count and -- inside the main loop
(the backward branch spanning the most MFMAs) -- the MFMA / total instruction counts, scratch (spill) operations and how the compiler
interleaved MFMAs with the rest.  The 168-VGPR (three workgroups per CU) and 256-VGPR tiles sit at their register limit: a harmless-looking
source change can push the allocator into spilling inside the loop (-3 % of the headline for 11 scratch ops), and only the ISA shows it.
    python tools/isa_check.py [substring of the demangled kernel name] [--tools]
Exit status 1 if a kernel on the forward's hot path spills in its loop (tests/test_isa.py)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (kernel-name substring, scratch operations tolerated in the loop): the ResnetBlock / decoder / FuseNet tiles of the headline forward
HOT = [("conv_h2_kernel<4, 64, 2, 2, 3, true, 0, 0>", 0), ("conv_h2_kernel<4, 64, 2, 2, 3, false, 0, 0>", 0),
       ("conv_h2_kernel<4, 128, 2, 2, 3, false, 0, 0>", 0), ("conv_h2_kernel<4, 128, 2, 2, 3, true, 0, 0>", 4),
       ("conv_h2_kernel<4, 64, 2, 2, 1, true, 0, 0>", 0), ("conv_h2_kernel<4, 64, 2, 2, 1, false, 0, 0>", 0),
       ("conv_h2_kernel<4, 64, 2, 2, 3, true, 0, 24>", 0), ("conv_h2_kernel<4, 32, 4, 1, 3, true, 0, 24>", 0),
       # the bf16 modes' 4 x 128 tile, four waves side by side (round 6): 168 VGPRs = the three workgroups per CU it is scheduled for
       ("conv_h2_kernel<4, 128, 1, 4, 1, true, 0, 0>", 0), ("conv_h2_kernel<4, 128, 1, 4, 1, false, 0, 0>", 0)]
# the encoder front end of the headline forward: the stride-2 patch kernel from 128 channels on, the 7 x 7 stem.  The IN + ReLU form of the
# stride-2 tile holds 2 spill operations per slab pair at its 168 VGPRs (three workgroups per CU): measured, tolerated, pinned so that it cannot grow unseen.
HOT_FRONT = [("conv_h2d_kernel<128, 4, 3, true, 2, false>", 2), ("conv_h2d_kernel<128, 4, 3, false, 2, false>", 0), ("conv_h2s_kernel<3>", 0)]
# the deep schedule of single-frame launches (round 6): 250 - 253 VGPRs, two workgroups per CU, no spill
HOT_FRONT_DEEP = [("conv_h2d_kernel<128, 4, 3, true, 2, true>", 0), ("conv_h2d_kernel<128, 4, 3, false, 2, true>", 0)]
# ... and of the general kernel's unit (conv_h2r_launch.cpp): the 64 -> 128 stride-2 layer and the two 1 x 1 convolutions
HOT_GENERAL = [("conv_h2r_kernel<3, 128, 2, 2, 3, true, false>", 0), ("conv_h2r_kernel<1, 64, 2, 2, 3, false, false>", 0)]
# ... and of its 64-deep-step form (conv_g64_launch.cpp, round 6), which now carries those layers in the forward: the 64-row tile, 1 x 1 on a raw input,
# 3 x 3 stride 2 under a fused InstanceNorm + ReLU (no spill with either operand form: the bf16 loop held 4 spill operations per step until its
# bf16-storage input path was rewritten to read 8-byte quads)
HOT_G64 = [("conv_g64_kernel<1, 64, 3, false>", 0), ("conv_g64_kernel<3, 64, 3, true>", 0), ("conv_g64_kernel<3, 64, 1, true>", 0), ("conv_g64_kernel<1, 64, 1, false>", 0)]
HOT_S32 = [("conv_h2s32_kernel<3>", 0), ("conv_h2s32_kernel<1>", 0)]      # the pose model's stems (same unit)


def loops_of(body):
    L = body.split("\n")
    labels = {m.group(1): k for k, l in enumerate(L) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    out = []
    for k, l in enumerate(L):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            b = L[labels[m.group(1)]:k + 1]
            out.append((sum("v_mfma" in x for x in b), b))
    return sorted(out, key=lambda t: -t[0])


def analyse(asm_path, pattern=""):
    s = open(asm_path).read()
    rows = []
    for m in re.finditer(r"^(_ZN5tsnet\w+):\s*;", s, re.M):
        nm = m.group(1)
        d = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
        d = d.replace("tsnet::", "").replace("(ConvArgs)", "").replace("(tsnet::ConvArgs)", "").replace("void ", "")
        if pattern not in d:
            continue
        body = s[m.start():s.index(".Lfunc_end", m.start())]
        lp = loops_of(body)
        if not lp or not lp[0][0]:
            continue
        n, b = lp[0]
        ins = [x.split()[0] for x in b if x.strip() and not x.strip().startswith((";", "."))]
        runs, cur, cnt = [], None, 0
        for x in ins:
            t = "M" if x.startswith("v_mfma") else "o"
            if t == cur:
                cnt += 1
            else:
                if cur:
                    runs.append((cur, cnt))
                cur, cnt = t, 1
        runs.append((cur, cnt))
        mr = [c for t, c in runs if t == "M"]
        orr = [c for t, c in runs if t == "o"] or [0]
        vg = re.search(re.escape(nm) + r"\.num_vgpr, (\d+)", s)
        rows.append(dict(name=d, vgpr=int(vg.group(1)) if vg else -1, mfma=n, total=len(ins), scratch=sum(x.startswith("scratch_") for x in ins),
                         mfma_run_max=max(mr), other_run_max=max(orr)))
    return rows


_ASM = {}


def compile_all(units, tools=False):
    """compile several translation units at once (hipcc is single-threaded per unit: four units in the time of the slowest); compile_asm then returns from the cache"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(units)) as ex:
        list(ex.map(lambda u: compile_asm(tools, u), units))


def compile_asm(tools=False, unit="conv_h2_launch.cpp"):
    if (tools, unit) not in _ASM:
        _ASM[(tools, unit)] = _compile_asm(tools, unit)
    return _ASM[(tools, unit)]


def _compile_asm(tools, unit):
    tmp = tempfile.mkdtemp(prefix="tsnet_isa_")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from wacv23_tsnet_amd import build as B          # the library's own flags: the ISA looked at is the ISA that ships
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + B.UNIT_FLAGS.get(unit, []) + ["-save-temps", "-I" + os.path.join(ROOT, "include")]
    if tools:
        cmd.append("-DTSNET_TOOLS")
    cmd += ["-c", os.path.join(ROOT, "wacv23_tsnet_amd", "csrc", unit), "-o", os.path.join(tmp, "x.o")]
    subprocess.run(cmd, cwd=tmp, check=True, capture_output=True)
    return os.path.join(tmp, unit.replace(".cpp", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernel_resources(asm_path, pattern):
    """{demangled kernel name: (vgpr_count, private_segment_fixed_size)} from the .amdhsa metadata (whole kernel: any scratch at all shows here)"""
    s = open(asm_path).read()
    out = {}
    for m in re.finditer(r"\.name:\s+(_ZN5tsnet\w+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", s):
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pattern in d:
            out[d] = (int(m.group(3)), int(m.group(2)))
    return out


def scratch_in_inner_loops(asm_path, pattern):
    """{demangled kernel name: scratch (spill) instructions inside INNERMOST loops}: a spill in straight-line prologue / tile-boundary code costs
    a few instructions per tile; one inside a period loop is paid every period beside the MFMAs."""
    s = open(asm_path).read()
    out = {}
    for m in re.finditer(r"^(_ZN5tsnet\w+):\s*;", s, re.M):
        d = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pattern not in d:
            continue
        L = s[m.start():s.index(".Lfunc_end", m.start())].split("\n")
        labels = {mm.group(1): k for k, l in enumerate(L) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = []
        for k, l in enumerate(L):
            mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                loops.append((labels[mm.group(1)], k))
        inner = [(a, b) for a, b in loops if not any((c, e) != (a, b) and a <= c and e <= b for c, e in loops)]
        out[d] = sum(1 for a, b in inner for l in L[a:b + 1] if l.strip().startswith("scratch_"))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = analyse(compile_asm("--tools" in sys.argv), args[0] if args else "")
    bad = 0
    for r in sorted(rows, key=lambda r: r["name"]):
        lim = [l for n, l in HOT if n in r["name"]]
        flag = ""
        if lim and r["scratch"] > lim[0]:
            flag = "   <-- spills in the loop of a hot kernel"; bad += 1
        print(f"{r['name']:62s} vgpr={r['vgpr']:3d} loop: mfma={r['mfma']:3d} instr={r['total']:4d} scratch={r['scratch']:2d} "
              f"longest mfma run={r['mfma_run_max']:2d} longest other run={r['other_run_max']:3d}{flag}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
