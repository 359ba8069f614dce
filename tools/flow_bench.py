"""GPU box: the transformation branch alone (tsnet_op_flow_k: L2-normalise + split into fp16 planes, correlation + softmax(100 .) +
soft-argmax) in the shape the model runs it -- B driving frames, K sources each.  BASELINE.json configs[4] per GPU: B = 1, K = 5, 64 x 64
positions (P = 4096), C = 512.  Prints the flow kernel's own time per launch (HIP events around `--iters` back-to-back launches) and the
kernel the plan picks; run it under rocprofv3 / tools/profile_round.sh (TSNET_PROF_CMD) for the counters.
    python tools/flow_bench.py [--hw 64] [--batch 1] [--sources 5] [--channels 512] [--iters 20]
    python tools/flow_bench.py --variants        # tools library: both kernels and the epilogue ablation, flows compared with the first"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--hw", type=int, default=64)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--sources", type=int, default=5)
ap.add_argument("--channels", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--variants", action="store_true")
a = ap.parse_args()
lib = _lib.load_tools() if a.variants else _lib.load()
h = w = a.hw; B = a.batch; K = a.sources; Cc = a.channels; H = W = 8 * a.hw
g = torch.Generator().manual_seed(0)
tar = torch.relu(torch.randn((B, h, w, Cc), generator=g)).cuda()
src = torch.randn((K * B, h, w, Cc), generator=g).cuda()
mt = (torch.rand((B, H, W), generator=g) > 0.5).float().cuda(); ms = (torch.rand((K * B, H, W), generator=g) > 0.5).float().cuda()
P = h * w


def run(variant):
    flow = torch.empty((K * B, h, w, 2), device="cuda")
    ms_out = C.c_float(0)
    rc = lib.tsnet_op_flow_k(tar.data_ptr(), src.data_ptr(), mt.data_ptr(), ms.data_ptr(), B, K, h, w, Cc, H, W, flow.data_ptr(), variant, a.iters + 1, C.byref(ms_out), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    torch.cuda.synchronize()
    return ms_out.value * 1e3, flow


work = {"positions": P, "channels": Cc, "batch": B, "sources": K, "plan_G": lib.tsnet_flow_plan(B, h, w, Cc),
        "algorithmic_gflop": round(2.0 * K * B * P * P * Cc / 1e9, 1), "mfma_gflop_issued": round(6.0 * K * B * P * P * Cc / 1e9, 1),
        "algorithmic_MB_planes": round((K + 1) * B * P * Cc * 4 / 1e6, 1)}
if not a.variants:
    us, _ = run(0)
    work["flow_kernel_us"] = round(us, 1)
    work["mfma_frac_of_2500TF"] = round(work["mfma_gflop_issued"] / us / 2500e-3, 3)
    print(json.dumps(work))
    sys.exit(0)

print(json.dumps(work))
variants = [("flow_kernel (a workgroup per source x 64 targets)", 1),
            ("flow_kernel_p (product at this shape)", 0),
            ("flow_kernel_p without the exp pass (ablation: garbage)", 2)]
ref = None
for name, env in variants:
    best = 1e9
    for _ in range(3):
        us, flow = run(env)
        best = min(best, us)
    if ref is None:
        ref = flow
    d = (flow - ref).abs().max().item()
    print(f"{name:62s} {best:8.1f} us   {work['mfma_gflop_issued'] / best / 1e-3:7.1f} TF issued   max|d flow| vs first {d:.2e}")
