import ctypes as C, os, statistics, sys
sys.path.insert(0, "/root/repo")
import torch
from wacv23_tsnet_amd import _lib
libs = {"this": _lib.load_tools(), "r4": _lib.bind(C.CDLL("/root/repo/wacv23_tsnet_amd/lib/libtsnet_tools_r4.so"))}
torch.zeros(1, device="cuda")
for name, shp, nrm, var, iters in (("res 3 img IN+ReLU", (3,32,32,512,512), 1, 32768, 8), ("res 3 img raw", (3,32,32,512,512), 0, 32768, 8), ("res 3 img IN+ReLU cold", (3,32,32,512,512), 1, 32768 | (1<<21), 24),
                                  ("res 12 img IN+ReLU", (12,32,32,512,512), 1, 32768, 8), ("res 12 img raw cold", (12,32,32,512,512), 0, 32768 | (1<<21), 24), ("fuse_c2 3 img", (3,32,32,1024,1024), 1, 32768, 8), ("dec_up0 1 img", (1,64,64,512,256), 0, 32768, 8)):
    res = {k: [] for k in libs}
    for r in range(5):
        for k, lib in libs.items():
            ms = C.c_float()
            rc = lib.tsnet_bench_conv(*shp, 3, 1, 1, 1, nrm, var, iters, C.byref(ms), None)
            res[k].append(ms.value * 1e3 if rc == 0 else float("nan"))
    print(f"{name:28s} " + "  ".join(f"{k}: {statistics.median(v):7.1f} us" for k, v in res.items()), flush=True)
