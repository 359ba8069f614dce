"""GPU-box diagnostic: ablation of the conv K loop (real / no global loads / no LDS stores+barriers)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load(); torch.zeros(1, device="cuda")
for (name, N, H, W, Cin, Cout, norm) in [("res_c1", 12, 32, 32, 512, 512, 0), ("res_c2", 12, 32, 32, 512, 512, 1), ("fuse_c2", 12, 32, 32, 1024, 1024, 1)]:
    flops = 2.0 * N * H * W * Cout * Cin * 9
    out = []
    for base in (0, 12):
        for abl in (0, 1, 2):
            ms = C.c_float()
            rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, 3, 1, 1, 1, norm, base + 16 * abl, 5, C.byref(ms), None)
            out.append(f"v{base}/abl{abl}:{ms.value:.3f}ms/{flops/ms.value/1e9:.0f}TF" if rc == 0 else "ERR " + lib.tsnet_op_last_error().decode())
    print(name, " ".join(out), flush=True)
