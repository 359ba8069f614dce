"""GPU-box diagnostic: time every conv shape of the cfg0 forward under every tile variant.
usage: conv_sweep.py [quick]   -> prints one line per (layer, variant): ms, TFLOP/s (real FLOPs)."""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda")
#        name       N   H    W   Cin  Cout k s p refl norm
LAYERS = [
    ("stem_img",   12, 256, 256,   8,   64, 7, 1, 3, 1, 0),
    ("down1",      12, 256, 256,  64,  128, 3, 2, 1, 0, 1),
    ("down2",      12, 128, 128, 128,  256, 3, 2, 1, 0, 1),
    ("down3",      12,  64,  64, 256,  512, 3, 2, 1, 0, 1),
    ("res_c1",     12,  32,  32, 512,  512, 3, 1, 1, 1, 0),
    ("res_c2",     12,  32,  32, 512,  512, 3, 1, 1, 1, 1),
    ("fuse_c2",    12,  32,  32, 1024, 1024, 3, 1, 1, 1, 1),
    ("fuse_1x1",    4,  32,  32, 1024, 512, 1, 1, 0, 0, 0),
    ("dec_up0",     4,  64,  64, 512,  256, 3, 1, 1, 1, 0),
    ("dec_up1",     4, 128, 128, 256,  128, 3, 1, 1, 1, 0),
    ("dec_up2",     4, 256, 256, 128,   64, 3, 1, 1, 1, 0),
    ("head",        4, 256, 256,  64,    3, 7, 1, 3, 1, 1),
    ("lbl_stem",    4, 256, 256,   8,   64, 7, 1, 3, 1, 0),
    ("lbl_down3",   4,  64,  64, 256,  512, 3, 2, 1, 0, 1),
]
TILES = {0: (128, 128), 1: (128, 64), 2: (64, 64), 3: (128, 32)}
DT = TILES
XT = {0: (128, 128), 1: (128, 128), 2: (128, 128), 3: (128, 128), 4: (128, 64), 5: (64, 64), 6: (64, 128), 7: (96, 128), 8: (128, 32), 9: (128, 128), 10: (128, 128), 11: (128, 128), 12: (128, 64), 13: (128, 128), 14: (128, 128), 15: (128, 64), 16: (128, 128), 17: (128, 64)}
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for (name, N, H, W, Cin, Cout, k, s, p, refl, norm) in LAYERS:
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    npad = (Cout + 127) // 128 * 128 if Cout >= 128 else (Cout + 31) // 32 * 32
    res = []
    if len(sys.argv) > 1 and sys.argv[1] == "h2":
        # conv_h2 (fp16x2, fused IN + ReLU when norm): variant 16384 + (1: 128-wide tiles) + (4: four products); next to x3q (X14 / X15)
        if not (k == 3 and s == 1): continue
        hres = []
        for v in (8192 + 14, 8192 + 15, 16384, 16384 + 1, 16384 + 4):
            for nrm in ((0, 1) if v & 16384 else (0,)):
                if (v & 16384) and (v & 1) and npad % 128: continue
                ms = C.c_float()
                rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, k, s, p, refl, nrm, v, 8, C.byref(ms), None)
                hres.append(f"{'H' if v & 16384 else 'X'}{v & 63}{'n' if nrm else ''}:" + (f"{ms.value:.3f}ms/{flops/ms.value/1e9:.0f}TF" if rc == 0 else "ERR " + lib.tsnet_op_last_error().decode()))
        print(f"{name:10s} M={N*Ho*Wo:7d} N={Cout:5d} K={Cin*k*k:5d} GF={flops/1e9:7.1f} | " + " ".join(hres), flush=True)
        continue
    for v in ([4096 + t for t in (0, 1)] + [8192 + t for t in (0, 4, 7, 14, 15, 16, 17)]):
        tl = XT if v & 8192 else (DT if v & 4096 else TILES)
        if npad % tl[v & 63][1] or (tl[v & 63][1] > 32 and Cout <= tl[v & 63][1] // 2):
            continue
        ms = C.c_float()
        iters = 5 if flops > 5e10 else 10
        rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, k, s, p, refl, 0 if v & (4096 | 8192) else norm, v, iters, C.byref(ms), None)
        if rc != 0:
            print(name, v, "ERR", lib.tsnet_op_last_error().decode()); continue
        res.append((v, ms.value, flops / ms.value / 1e9))
    if not res: continue
    best = max(r[2] for r in res)
    print(f"{name:10s} M={N*Ho*Wo:7d} N={Cout:5d} K={Cin*k*k:5d} GF={flops/1e9:7.1f} | " +
          " ".join(f"{('h' if v < 0 else ('%s%d' % ('X' if v & 8192 else 'D' if v & 4096 else 't', v & 63)))}:{ms:.3f}ms/{tf:.0f}TF{'*' if tf == best else ''}" for v, ms, tf in res), flush=True)
