"""GPU box: this tree's tools library against another build of the same ABI (--lib2, e.g. the previous commit's), same box, same process,
interleaved: (1) the ResnetBlock convolution alone -- 8 launches (burst), 24 launches (sustained), 24 launches on cold weights (a fresh copy of
the planes per launch, as consecutive layers of a forward read them); (2) the whole forward; (3) the forward's per-class kernel times
(hipEvent brackets: one lane, serialised).
    python tools/lib_ab.py --lib2 wacv23_tsnet_amd/lib/libtsnet_tools_prev.so [--rounds 5]"""
import argparse, ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib, synth
from wacv23_tsnet_amd.engine import TSNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--lib2", required=True)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
libs = {"this tree": _lib.load_tools(), os.path.basename(a.lib2): _lib.bind(C.CDLL(a.lib2))}
torch.zeros(1, device="cuda")
W1 = 32768
# (name, (N, H, W, Cin, Cout), norm | stats, variant, launches[, ksize, stride, pad, reflect])
OTHER = [("stem 7x7 8->64 (12 images)", (12, 256, 256, 8, 64), 2, 0, 8, 7, 1, 3, 1), ("stem 7x7 8->64 (4 images)", (4, 256, 256, 8, 64), 2, 0, 8, 7, 1, 3, 1),
         ("down1 64->128 s2 (12)", (12, 256, 256, 64, 128), 3, 0, 8, 3, 2, 1, 0), ("down2 128->256 s2 (12)", (12, 128, 128, 128, 256), 3, 0, 8, 3, 2, 1, 0),
         ("down3 256->512 s2 (12)", (12, 64, 64, 256, 512), 3, 0, 8, 3, 2, 1, 0), ("dec_up2 128->64 @256^2 (4)", (4, 256, 256, 128, 64), 2, 0, 8, 3, 1, 1, 1),
         ("1x1 1024->512 (4)", (4, 32, 32, 1024, 512), 0, 0, 8, 1, 1, 0, 0),
         ("bf16: res 512->512 (24 images)", (24, 32, 32, 512, 512), 3, 8192, 8, 3, 1, 1, 1), ("bf16: dec_up1 256->128 @128^2 (8)", (8, 128, 128, 256, 128), 2, 8192, 8, 3, 1, 1, 1),
         ("bf16: dec_up2 128->64 @256^2 (8)", (8, 256, 256, 128, 64), 2, 8192, 8, 3, 1, 1, 1), ("bf16: down2 128->256 s2 (24)", (24, 128, 128, 128, 256), 3, 8192, 8, 3, 2, 1, 0)]
cases = [("res IN+ReLU+stats, 8 launches", (12, 32, 32, 512, 512), 3, W1, 8), ("res IN+ReLU+stats, 24 launches", (12, 32, 32, 512, 512), 3, W1, 24),
         ("res IN+ReLU+stats, 24 launches, cold weights", (12, 32, 32, 512, 512), 3, W1 | (1 << 21), 24),
         ("res raw+stats, 24 launches, cold weights", (12, 32, 32, 512, 512), 2, W1 | (1 << 21), 24),
         ("fuse_c2 IN+ReLU+stats, 8 launches", (12, 32, 32, 1024, 1024), 3, W1, 8)]
for case in cases + OTHER:
    name, (N, H, W, Ci, Co), nrm, var, iters = case[:5]
    ks, st, pd, rf = case[5:] if len(case) > 5 else (3, 1, 1, 1)
    res = {k: [] for k in libs}
    for r in range(a.rounds):
        for k, lib in libs.items():
            ms = C.c_float()
            rc = lib.tsnet_bench_conv(N, H, W, Ci, Co, ks, st, pd, rf, nrm, var, iters, C.byref(ms), None)
            res[k].append(ms.value * 1e3 if rc == 0 else float("nan"))
    print(f"{name:50s} " + "   ".join(f"{k}: {statistics.median(v):7.1f} us" for k, v in res.items()), flush=True)
inp = synth.inputs(3, 2, 4, 256, 256, seed=1)
si, sl, sb, tl, tb = [[t.cuda() for t in x] if isinstance(x, list) else x.cuda() for x in inp]
eng = {}
for k, lib in libs.items():
    e = TSNetEngine(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3, height=256, width=256, max_batch=4, lib=lib)
    e.load_state_dict(synth.state_dict(e.param_shapes(), seed=0)); e.finalize("cuda")
    eng[k] = e
tm = {k: [] for k in eng}
for r in range(a.rounds + 1):
    for k, e in eng.items():
        for _ in range(3):
            e.forward(si, sl, sb, tl, tb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            e.forward(si, sl, sb, tl, tb)
        torch.cuda.synchronize()
        if r:
            tm[k].append((time.perf_counter() - t0) / 20 * 1e3)
print("forward B=4: " + "   ".join(f"{k}: {statistics.median(v):.3f} ms" for k, v in tm.items()))
for k, e in eng.items():
    e.timing_enable(True)
    for _ in range(5):
        e.forward(si, sl, sb, tl, tb)
    torch.cuda.synchronize()
    t = e.timing_read(reset=True)
    e.timing_enable(False)
    print(f"   {k:28s} " + "  ".join(f"{n}: {v[0] / 5:.3f}" for n, v in t.items() if v[1]))
