"""GPU-box diagnostic: ablation of the bf16x3 conv K loops.  usage: x3_ablate.py [legacy|patch|q|h2]
legacy (conv_x3.hpp) mask bits: 1 no DMA, 2 no barrier/vmcnt, 4 no fold, 8 no ds_read, 16 no B DMAs, 32 no A DMAs, 64 contiguous A DMAs.
patch (conv_x3p.hpp x3p) mask bits: 1 no DMA, 2 no barrier/vmcnt, 4 no fold, 8 fragments read once.
q (conv_x3p.hpp x3q, 128x64 tiles) mask bits: 1 no patch staging, 2 no slab barrier, 4 no fold, 8 weight fragments loaded once,
16 A fragments read once, 32 plane 1 unused (timing proxy of a two-plane / three-product scheme).  (bits 4 and 16 let hipcc delete or reschedule the MFMAs -- only 1, 2, 8 are meaningful.)
h2 (conv_h2.hpp, 64-wide tiles, raw input) mask bits: 1 no patch staging, 2 weight fragments loaded once, 4 A fragments read once,
8 no fold, 16 no slab barrier.  Needs the tools build of the library (python -m wacv23_tsnet_amd.build --tools)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load_tools(); torch.zeros(1, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "legacy"
patch = mode == "patch"
masks = {"patch": (0, 1, 2, 4, 8, 12, 15, 0), "q": (0, 1, 2, 8, 32, 33, 40, 41, 0), "h2": (0, 1, 2, 4, 8, 16, 3, 6, 7, 15, 31, 0)}.get(mode, (0, 16, 32, 64, 1, 0))
tile = {"patch": 11, "q": 15}.get(mode, 0)
base = 16384 if mode == "h2" else 8192 + tile
for (name, N, H, W, Cin, Cout) in [("res", 12, 32, 32, 512, 512), ("res_x16", 16, 32, 32, 512, 512), ("fuse_c2", 12, 32, 32, 1024, 1024), ("dec_up1", 4, 128, 128, 256, 128)]:
    flops = 2.0 * N * H * W * Cout * Cin * 9
    out = []
    for abl in masks:
        ms = C.c_float()
        rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, 3, 1, 1, 1, 0, base + 65536 * abl, 6, C.byref(ms), None)
        out.append(f"abl{abl}:{ms.value:.3f}ms/{flops/ms.value/1e9:.0f}TF" if rc == 0 else "ERR " + lib.tsnet_op_last_error().decode())
    print(name, " ".join(out), flush=True)
