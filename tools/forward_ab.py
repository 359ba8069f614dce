"""GPU box: same-box, same-process A/B of the whole forward under a tools-build knob (tsnet_tools_set; the product library has no knobs).
Interleaved rounds, medians.  knob 0 = largest conv_w1 chunk the launcher may choose (3 = product behaviour, 1 = one tile per workgroup).
    python tools/forward_ab.py [--batch 4] [--n-blocks 0] [--rounds 7] [--iters 20] [--knob 0] [--values 3,1]
--lib2 PATH: A/B of two BUILDS instead (this tree's tools library against another .so of the same ABI, e.g. the previous commit's, built
into wacv23_tsnet_amd/lib/libtsnet_tools_prev.so): one engine per library, interleaved rounds in one process on one box."""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib, synth
from wacv23_tsnet_amd.engine import TSNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--n-blocks", type=int, default=0)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--knob", type=int, default=0)
ap.add_argument("--values", default="3,1")
ap.add_argument("--operands", default="fp32", help="fp32 | bf16 | bf16s (tsnet_cfg.operand_mode)")
ap.add_argument("--lib2", default=None)
ap.add_argument("--reps", type=int, default=6, help="engine instantiations per build (--lib2)")
a = ap.parse_args()
lib = _lib.load_tools()
vals = [int(v) for v in a.values.split(",")]
inp = synth.inputs(3, 2, a.batch, 256, 256, seed=1)
si, sl, sb, tl, tb = [[t.cuda() for t in x] if isinstance(x, list) else x.cuda() for x in inp]
if not a.lib2:
    eng = TSNetEngine(label_nc=2, n_blocks=a.n_blocks, n_downsampling=3, n_source=3, height=256, width=256, max_batch=a.batch, operands=a.operands, lib=lib)
    eng.load_state_dict(synth.state_dict(eng.param_shapes(), seed=0)); eng.finalize("cuda")
    step = lambda: eng.forward(si, sl, sb, tl, tb)[0]
outs, res = {}, {v: [] for v in vals}
for r in range(a.rounds + 1 if not a.lib2 else 0):
    for v in vals:
        lib.tsnet_tools_set(a.knob, v)
        for _ in range(3):
            o = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters):
            o = step()
        torch.cuda.synchronize()
        if r:                                   # round 0 warms up
            res[v].append((time.perf_counter() - t0) / a.iters * 1e3)
        outs[v] = o.clone()
if a.lib2:
    # Engines are re-created several times: where an engine's arena and weights land in HBM moves the forward by a few per cent (the same build
    # measured 5.07 and 5.34 ms in two engines of one process), so one instantiation per build says little -- medians over instantiations do.
    import ctypes
    lib2 = _lib.bind(ctypes.CDLL(a.lib2))
    names = ["this tree", os.path.basename(a.lib2)]
    libs2 = {names[0]: lib, names[1]: lib2}
    per = {k: [] for k in names}
    cls = {}
    for rep in range(a.reps):
        order = names if rep % 2 == 0 else names[::-1]
        es = {}
        for k in order:
            e = TSNetEngine(label_nc=2, n_blocks=a.n_blocks, n_downsampling=3, n_source=3, height=256, width=256, max_batch=a.batch, operands=a.operands, lib=libs2[k])
            e.load_state_dict(synth.state_dict(e.param_shapes(), seed=0)); e.finalize("cuda")
            es[k] = e
        r2 = {k: [] for k in names}
        for r in range(a.rounds + 1):
            for k in order:
                f = es[k]
                for _ in range(3):
                    f.forward(si, sl, sb, tl, tb)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(a.iters):
                    f.forward(si, sl, sb, tl, tb)
                torch.cuda.synchronize()
                if r:
                    r2[k].append((time.perf_counter() - t0) / a.iters * 1e3)
        for k in names:
            per[k].append(statistics.median(r2[k]))
            if rep == a.reps - 1:                # per-class kernel times of the last instantiation (hipEvent brackets: one lane, serialised)
                es[k].timing_enable(True)
                for _ in range(5):
                    es[k].forward(si, sl, sb, tl, tb)
                torch.cuda.synchronize()
                cls[k] = {n: v[0] / 5 for n, v in es[k].timing_read(reset=True).items() if v[1]}
                es[k].timing_enable(False)
            es[k].close()
        print(f"instantiation {rep}: " + "   ".join(f"{k}: {per[k][-1]:.3f} ms" for k in names), flush=True)
    for k in names:
        print(f"{k:28s}: median over {a.reps} instantiations {statistics.median(per[k]):.3f} ms (min {min(per[k]):.3f} max {max(per[k]):.3f})  {a.batch / statistics.median(per[k]) * 1e3:.1f} frames/s")
        print("   per class, one lane: " + "  ".join(f"{n}: {v:.3f}" for n, v in cls[k].items()) + f"   sum {sum(cls[k].values()):.3f}")
    sys.exit(0)
for v in vals:
    t = res[v]
    print(f"knob {a.knob} = {v}: median {statistics.median(t):.3f} ms  (min {min(t):.3f} max {max(t):.3f})  {a.batch / statistics.median(t) * 1e3:.1f} frames/s   same bits as value {vals[0]}: {torch.equal(outs[v], outs[vals[0]])}")
