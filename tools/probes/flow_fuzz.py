"""CPU only: random map sizes, channel counts, source counts and batches through the transformation branch (tsnet_op_flow_k / tsnet_op_flow /
tsnet_op_warp) on the emulation build against the oracle's transformation_branch -- the kernel choice (flow_kernel<1|2> below 2048 positions,
the persistent flow_kernel_p from there on), the slice count and the number of workgroups sharing a (source, target tile) are the launcher's.
Ragged maps (positions not a multiple of 32 or 64), every mask model, dominant sources planted or not.  Each case also runs twice and
must return the same bits (op_cases.flow_case / flow_k_case assert it).
    python tools/probes/flow_fuzz.py [cases] [seed]"""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GATE = 5e-5         # tests/test_emu_ops.py's gate on |d flow| (flows live in [-1, 1])


def main():
    import conftest
    import op_cases as oc
    from wacv23_tsnet_amd import _lib
    lib = _lib.bind(ctypes.CDLL(conftest.build_emu_lib()))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad, refused, worst = [], [], 0.0
    t0 = time.time()
    for i in range(n):
        h, w = rng.choice([(4, 4), (4, 6), (5, 7), (7, 9), (8, 8), (6, 12), (8, 16), (12, 20), (16, 16), (24, 24), (32, 32), (32, 64), (48, 48), (40, 52), (64, 64)])
        big = h * w >= 2048
        C = rng.choice([8, 16, 32]) if big else rng.choice([16, 32, 64, 128, 256, 512])
        B, K = rng.randint(1, 4), rng.randint(1, 5)
        if big:
            B, K = rng.randint(1, 3), rng.randint(1, 3)
        mask, spike = rng.choice(["bernoulli", "soft", "ones", "zeros"]), rng.random() < 0.5
        desc = dict(B=B, K=K, h=h, w=w, C=C, mask=mask, spike=spike)
        try:
            dk = oc.flow_k_case(lib, "cpu", B, K, h, w, C, mask, seed=700 + i, spike=spike)
            df, dw = oc.flow_case(lib, "cpu", B, h, w, C, mask, seed=700 + i, spike=spike) if not big else (0.0, 0.0)
        except AssertionError as ex:
            (refused if "tsnet" in str(ex) or str(ex) else bad).append((desc, str(ex)[:200] or "twice-equal assertion"))
            print(f"[{i:3d}] {desc}  ASSERT {str(ex)[:120]}", flush=True)
            continue
        worst = max(worst, dk, df)
        ok = dk < GATE and df < GATE and dw < 4e-3      # dw = flow error x feature gradient (the operator tests' gate)
        if not ok:
            bad.append((desc, (dk, df, dw)))
        print(f"[{i:3d}] {desc}  d_flow(K) {dk:.2e} d_flow {df:.2e} d_warp {dw:.2e}  {'ok' if ok else 'WRONG'}", flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s: worst |d flow| {worst:.2e}; wrong {len(bad)}; refused / asserted {len(refused)}")
    for r in bad:
        print("WRONG", *r)
    for r in refused:
        print("REFUSED", *r)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
