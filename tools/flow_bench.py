"""GPU box: the transformation branch alone (tsnet_op_flow: L2-normalise + split into fp16 planes, correlation + softmax(100 .) + soft-argmax)
at a chosen feature size -- BASELINE.json configs[4] is 64 x 64 positions (P = 4096), C = 512, five sources.  Prints the time per call
(hipMalloc / sync included: use it under rocprofv3 for the kernel's own duration and counters, tools/profile_round.sh with TSNET_PROF_CMD).
    python tools/flow_bench.py [--hw 64] [--images 5] [--iters 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--images", type=int, default=5)
ap.add_argument("--channels", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
lib = _lib.load()
h = w = a.hw; B = a.images; C = a.channels; H = W = 8 * a.hw
g = torch.Generator().manual_seed(0)
tar = torch.randn((B, h, w, C), generator=g).cuda(); src = torch.randn((B, h, w, C), generator=g).cuda()
mt = (torch.rand((B, H, W), generator=g) > 0.5).float().cuda(); ms = (torch.rand((B, H, W), generator=g) > 0.5).float().cuda()
flow = torch.empty((B, h, w, 2), device="cuda")
for i in range(a.iters + 2):
    if i == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.tsnet_op_flow(tar.data_ptr(), src.data_ptr(), mt.data_ptr(), ms.data_ptr(), B, h, w, C, H, W, flow.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
torch.cuda.synchronize()
P = h * w
print(json.dumps({"positions": P, "channels": C, "image_pairs": B, "ms_per_call_with_malloc_and_sync": round((time.perf_counter() - t0) / a.iters * 1e3, 3),
                  "algorithmic_gflop": round(2.0 * B * P * P * C / 1e9, 1), "mfma_gflop_issued": round(6.0 * B * P * P * C / 1e9, 1),
                  "algorithmic_MB": round(2 * B * P * C * 4 / 2**20, 1)}))
