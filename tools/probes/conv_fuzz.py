"""CPU only: random convolution shapes through tsnet_op_conv2d on the emulation build (tests/conftest.py build_emu_lib), kernel = 0 -- the
forward's own choice of kernel, tile and chunk for the layer (engine.cpp run_conv) -- against an fp64 torch reference.  The fixed operator
tests name the forward's shapes; this walks the selection logic over shapes nobody named: ragged frames, channel counts at and off the
tiles' multiples, every batch 1..6, both pads, the fused InstanceNorm + ReLU on and off.  A refusal (rc != 0 with a message) is a loud
failure and is counted separately from a wrong result.
    python tools/probes/conv_fuzz.py [cases] [seed] [w1]      (w1: only the Winograd kernel's chunk cases)"""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REL = 2e-6          # the operator tests' gate for the fp16 x 2 form (tests/test_emu_ops.py)


def main():
    import conftest
    import op_cases as oc
    from wacv23_tsnet_amd import _lib
    lib = _lib.bind(ctypes.CDLL(conftest.build_emu_lib()))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad, refused, worst = [], [], 0.0
    t0 = time.time()
    for i in range(0 if "w1" in sys.argv else n):
        k, stride = rng.choice([(3, 1), (3, 1), (3, 1), (3, 2), (1, 1), (7, 1)])
        cin = rng.choice([8, 16, 24, 32, 48, 64, 80, 128, 256]) if k != 7 else rng.choice([8, 16, 32])
        cout = rng.choice([16, 32, 64, 128, 256]) if k != 7 else rng.choice([32, 64])
        N = rng.randint(1, 6)
        H = rng.choice([8, 12, 16, 20, 32])
        W = rng.choice([16, 24, 32, 64]) if k != 7 else rng.choice([32, 64])
        reflect = (k == 7) or (k == 3 and stride == 1 and rng.random() < 0.6)
        norm = rng.random() < 0.5
        pad = k // 2
        desc = dict(N=N, H=H, W=W, Cin=cin, Cout=cout, k=k, stride=stride, reflect=reflect, norm=norm)
        try:
            e = oc.conv_case(lib, "cpu", N, H, W, cin, cout, k, stride, pad, reflect, norm=norm, seed=100 + i)
        except AssertionError as ex:
            refused.append((desc, str(ex)[:160]))
            continue
        worst = max(worst, e)
        if not e < REL:
            bad.append((desc, e))
        print(f"[{i:3d}] {desc}  rel {e:.2e}", flush=True)
    # the Winograd kernel named explicitly (kernel = 3), chunk forced to 0 (the launcher's choice) / 1 (the one-tile kernel) / 2 / 3: shapes whose tile count
    # does not divide by the chunk are refused by the launcher; everything accepted must be right AND the same bits as the one-tile kernel's
    for i in range(n if "w1" in sys.argv else n // 2):
        cin = rng.choice([16, 32, 48, 64, 80, 96, 112, 128, 144, 256, 512])
        cout = rng.choice([32, 64, 128, 256])
        N, H, W = rng.randint(1, 6), rng.choice([8, 16, 32, 64]), rng.choice([32, 64])
        if rng.random() < 0.7:      # chunks need whole rows of the tile matrix per XCD (tiles of 8 x 32 pixels, a multiple of 8 of them) and >= 5 slabs
            N, H, W = rng.choice([(1, 64, 32), (2, 32, 32), (2, 64, 32), (3, 64, 32), (3, 32, 64), (4, 16, 32), (6, 32, 32), (4, 32, 32), (2, 32, 64),
                                  (6, 16, 64), (1, 64, 64), (5, 64, 32), (3, 16, 64)])
            cin = rng.choice([80, 96, 112, 128, 144, 256, 512])
        reflect, norm = rng.random() < 0.6, rng.random() < 0.5
        desc = dict(N=N, H=H, W=W, Cin=cin, Cout=cout, k="w1", reflect=reflect, norm=norm)
        outs = {}
        for c in (1, 0, 2, 3):
            try:
                outs[c] = oc.conv_w1_case(lib, "cpu", N, H, W, cin, cout, reflect, norm=norm, seed=500 + i, chunk=c, return_output=True)
            except AssertionError as ex:
                refused.append((dict(desc, chunk=c), str(ex)[:160]))
        if 1 in outs:
            e = oc.conv_w1_case(lib, "cpu", N, H, W, cin, cout, reflect, norm=norm, seed=500 + i, chunk=1)
            worst = max(worst, e)
            same = all(bool((o == outs[1]).all()) for o in outs.values())
            if not (e < REL and same):
                bad.append((desc, e if same else "chunks differ in bits"))
            print(f"[w1 {i:3d}] {desc}  rel {e:.2e}  chunks run {sorted(outs)} same bits {same}", flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s: worst rel {worst:.2e}; wrong: {len(bad)}; refused: {len(refused)}")
    for d, e in bad:
        print("WRONG", d, e)
    for d, m in refused:
        print("REFUSED", d, m)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
