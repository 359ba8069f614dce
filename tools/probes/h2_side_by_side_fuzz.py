"""GPU box: the bf16 4 x 128 patch tile with its four waves side by side (tile code 3128; the bf16 modes' own 3 x 3 tile since round 6)
against the 2 x 2 wave grid (128) over random shapes nobody named in the operator tests: every batch 1..8, heights 4..64, widths 32..128,
odd and even slab counts (Cin 16..512), 128..512 output channels, both pads, the fused InstanceNorm + ReLU on and off.  The two forms run the
same K order and chains: torch.equal, plus the bf16 operand-rounding class (2e-2) against fp64.  The launcher's own choice (tile 0) must
be one of the two.
    python tools/probes/h2_side_by_side_fuzz.py [cases] [seed]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import op_cases as oc
    from wacv23_tsnet_amd import _lib
    lib = _lib.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad, worst = [], 0.0
    for i in range(n):
        d = dict(N=rng.randint(1, 8), H=4 * rng.randint(1, 16), W=32 * rng.randint(1, 4), Cin=16 * rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 24, 32]),
                 Cout=128 * rng.randint(1, 4), reflect=rng.random() < 0.6, norm=rng.random() < 0.6)
        if d["reflect"] and d["H"] < 2:
            continue
        a = (d["N"], d["H"], d["W"], d["Cin"], d["Cout"], d["reflect"])
        ys = [oc.conv_h2_case(lib, "cuda", *a, norm=d["norm"], nprod=1, tile_n=t, seed=i, return_output=True) for t in (128, 3128, 0)]
        rel = oc.conv_h2_case(lib, "cuda", *a, norm=d["norm"], nprod=1, tile_n=3128, seed=i)
        ok = torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2]) and rel < 2e-2
        worst = max(worst, rel)
        print(f"[{i:3d}] {d}  rel {rel:.2e}  equal bits {torch.equal(ys[0], ys[1])} own choice equal {torch.equal(ys[1], ys[2])}", flush=True)
        if not ok:
            bad.append(d)
    print(f"{n} cases, wrong: {len(bad)}, worst rel {worst:.2e}")
    for d in bad:
        print("   WRONG", d)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
