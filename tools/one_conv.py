"""GPU-box diagnostic: run ONE conv shape/variant a few times (for rocprofv3 --pmc / --kernel-trace).
usage: one_conv.py N H W Cin Cout k s p refl norm variant iters"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load(); torch.zeros(1, device="cuda")
a = [int(x) for x in sys.argv[1:13]]
ms = C.c_float()
rc = lib.tsnet_bench_conv(*a, C.byref(ms), None)
print("rc", rc, "ms", ms.value, lib.tsnet_op_last_error().decode() if rc else "")
