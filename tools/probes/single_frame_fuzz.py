"""GPU box: the launch choices round 6 added for launches that do not fill the chip, over random shapes nobody named in the operator tests.
  * head_conv3 on 8- / 16- / 32-row tiles and the launcher's own choice: torch.equal between all four, the fp32 class against torch
    (frames of 8 .. 256 rows -- ragged against every tile height --, widths 32 .. 256, batches 1 .. 6, 16 / 32 / 64 channels, with and
    without the fused InstanceNorm + ReLU);
  * the stride-2 patch tile's deep schedule (tile code 12128) against the plain one (2128) and the launcher's choice: torch.equal, the
    fp16 x 2 class against fp64 (batches 1 .. 6, output heights 4 .. 64, widths 32 .. 128, 128 .. 512 input channels, 128 .. 512 output channels).
    python tools/probes/single_frame_fuzz.py [cases] [seed]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad, worst_h, worst_c = [], 0.0, 0.0
    for i in range(n):
        d = dict(N=rng.randint(1, 6), H=8 * rng.randint(1, 32), W=32 * rng.randint(1, 8), C=rng.choice([16, 32, 64]), norm=rng.random() < 0.7)
        ys = [oc.head_case(lib, "cuda", d["N"], d["H"], d["W"], d["C"], norm=d["norm"], seed=i, rows=r, return_output=True) for r in (8, 16, 32, 0)]
        err = oc.head_case(lib, "cuda", d["N"], d["H"], d["W"], d["C"], norm=d["norm"], seed=i)
        eq = all(torch.equal(ys[0], y) for y in ys[1:])
        worst_h = max(worst_h, err)
        print(f"[head {i:3d}] {d}  max|d| {err:.2e}  equal bits {eq}", flush=True)
        if not eq or err > 2e-5:
            bad.append(("head", d))
    for i in range(n):
        d = dict(N=rng.randint(1, 6), Ho=4 * rng.randint(1, 16), Wo=32 * rng.randint(1, 4), Cin=rng.choice([128, 256, 512]), Cout=128 * rng.randint(1, 4), norm=rng.random() < 0.7)
        a = (d["N"], 2 * d["Ho"], 2 * d["Wo"], d["Cin"], d["Cout"], 3)
        ys = [oc.conv_h2r_case(lib, "cuda", *a, norm=d["norm"], kernel=2, tile=t, seed=i, return_output=True) for t in (2128, 12128, 0)]
        rel = oc.conv_h2r_case(lib, "cuda", *a, norm=d["norm"], kernel=2, tile=12128, seed=i)
        eq = all(torch.equal(ys[0], y) for y in ys[1:])
        worst_c = max(worst_c, rel)
        print(f"[h2d  {i:3d}] {d}  rel {rel:.2e}  equal bits {eq}", flush=True)
        if not eq or rel > 2e-6:
            bad.append(("h2d", d))
    print(f"{2 * n} cases, wrong: {len(bad)}, worst head max|d| {worst_h:.2e}, worst stride-2 rel {worst_c:.2e}")
    for d in bad:
        print("   WRONG", d)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
