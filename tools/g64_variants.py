"""GPU box: conv_g64 (64-deep K steps, csrc/conv_g64.hpp) against conv_h2r (16-deep steps) on the layers the general kernel class carries in
the forward -- interleaved rounds in ONE process, medians (tsnet_bench_conv: HIP events around `iters` back-to-back launches).
variant code: tile | general kernel << 12 | bf16 operands << 13.   tile 64 / 128 = conv_h2r with that width; 3064 / 3128 = conv_g64, 64 / 128 rows.
usage: g64_variants.py [rounds]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
GEN, BF = 1 << 12, 1 << 13
# name, (N, H, W, Cin, Cout, k, stride, pad, reflect), norm modes (1 = IN + ReLU on load, 2 = statistics of the output), variants
CASES = [
    ("pose stem 7x7 32->64, 12 images 256^2 (conv_h2s32 = the layer's own kernel)", (12, 256, 256, 32, 64, 7, 1, 3, 1), (2,), [("h2r 128x64", GEN | 64), ("h2s32 (own)", 0)]),
    ("pose stem, 4 images (target lane)", (4, 256, 256, 32, 64, 7, 1, 3, 1), (2,), [("h2r 128x64", GEN | 64), ("h2s32 (own)", 0)]),
    ("fuse_net.conv / dec.map_conv 1x1 1024->512, B=4", (4, 32, 32, 1024, 512, 1, 1, 0, 0), (0,), [("h2r 128x64", GEN | 64), ("g64 64x128", GEN | 3064), ("g64 128x128", GEN | 3128)]),
    ("same, one frame", (1, 32, 32, 1024, 512, 1, 1, 0, 0), (0,), [("h2r 128x64", GEN | 64), ("g64 64x128", GEN | 3064)]),
    ("down1 64->128 s2, 12 images 256^2", (12, 256, 256, 64, 128, 3, 2, 1, 0), (3,), [("h2r 128x128", GEN | 128), ("g64 64x128", GEN | 3064), ("g64 128x128", GEN | 3128)]),
    ("down1, 4 images (target lane)", (4, 256, 256, 64, 128, 3, 2, 1, 0), (3,), [("h2r 128x128", GEN | 128), ("g64 128x128", GEN | 3128)]),
    ("bf16 down1 64->128 s2, 24 images", (24, 256, 256, 64, 128, 3, 2, 1, 0), (3,), [("h2r 128x128", GEN | BF | 128), ("g64 64x128", GEN | BF | 3064), ("g64 128x128", GEN | BF | 3128)]),
    ("bf16 down2 128->256 s2, 24 images", (24, 128, 128, 128, 256, 3, 2, 1, 0), (3,), [("h2r 128x128", GEN | BF | 128), ("g64 64x128", GEN | BF | 3064), ("g64 128x128", GEN | BF | 3128)]),
    ("bf16 down3 256->512 s2, 24 images", (24, 64, 64, 256, 512, 3, 2, 1, 0), (3,), [("h2r 128x128", GEN | BF | 128), ("g64 64x128", GEN | BF | 3064), ("g64 128x128", GEN | BF | 3128)]),
    ("fp16x2 down2 128->256 s2, 12 images (forward: conv_h2d)", (12, 128, 128, 128, 256, 3, 2, 1, 0), (3,), [("h2d (own)", 0), ("g64 128x128", GEN | 3128)]),
    ("fp16x2 down3 256->512 s2, 12 images (forward: conv_h2d)", (12, 64, 64, 256, 512, 3, 2, 1, 0), (3,), [("h2d (own)", 0), ("g64 64x128", GEN | 3064), ("g64 128x128", GEN | 3128)]),
]
for name, shp, norms, variants in CASES:
    N, H, W, Cin, Cout, k, s, p, refl = shp
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    iters = 10
    res = {}
    for r in range(rounds + 1):
        for vn, v in variants:
            for nrm in norms:
                ms = C.c_float()
                rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, k, s, p, refl, nrm, v, iters, C.byref(ms), None)
                if r:
                    res.setdefault((vn, nrm), []).append(ms.value if rc == 0 else float("nan"))
                if rc != 0 and r == 0:
                    print("   ERR", vn, lib.tsnet_op_last_error().decode())
    print(f"{name}: M={N*Ho*Wo} N={Cout} K={Cin*k*k} {flops/1e9:.1f} GFLOP")
    for (vn, nrm), t in res.items():
        med = statistics.median(t)
        print(f"   {vn:16s} {'IN+ReLU' if nrm & 1 else 'raw    '}{'+stats' if nrm & 2 else '      '}  {med*1e3:8.1f} us  {flops/med/1e9:7.1f} TF   (min {min(t)*1e3:.1f} max {max(t)*1e3:.1f})", flush=True)
