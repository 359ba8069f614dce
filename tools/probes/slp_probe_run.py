"""GPU box (<= 10 GPU-minutes): is the wrong-result fault of the SLP-vectorized flow_kernel_p (packed fp32 arithmetic beside the MFMAs,
csrc/flow_persist.hpp) a hazard the compiler misses or something else?  For each probe build (tools/probes/slp_probe_build.py) the configs[4]
form (one driving frame, five sources, 64 x 64 positions, 512 channels) runs R times from fresh scratch memory; reported: positions whose
flow differs from the product build's (scalar arithmetic: exact, run-to-run identical), run-to-run differences, and which target columns."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from wacv23_tsnet_amd import _lib, prng
import op_cases as oc

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, K, h, w, Cc = 1, 5, 64, 64, 512
H, W, P = h * 8, w * 8, h * w
tar = F.relu(oc._rand(0, "tar", (B, Cc, h, w), -1, 1))
mt = prng.bernoulli(0, "mt", (B, H, W))
srcs = [oc._rand(1 + k, "src", (B, Cc, h, w), -1, 1) * 3 for k in range(K)]
mss = [prng.bernoulli(1 + k, "ms", (B, H, W)) for k in range(K)]
tard = oc.nhwc(tar).cuda(); srcd = torch.cat([oc.nhwc(s) for s in srcs], 0).contiguous().cuda()
mtd, msd = mt.cuda(), torch.cat(mss, 0).contiguous().cuda()

def run(lib):
    flow = torch.empty((K * B, h, w, 2), device="cuda")
    rc = lib.tsnet_op_flow_k(tard.data_ptr(), srcd.data_ptr(), mtd.data_ptr(), msd.data_ptr(), B, K, h, w, Cc, H, W, flow.data_ptr(), 0, 1, None, None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    torch.cuda.synchronize()
    return flow.cpu()

ref = run(_lib.load())
assert torch.equal(ref, run(_lib.load()))
print(f"product build (scalar fp32 arithmetic in the epilogue): {R} runs identical: {all(torch.equal(ref, run(_lib.load())) for _ in range(R))}")
for k in (0, 1, 2):
    path = os.path.join(ROOT, "wacv23_tsnet_amd", "lib", f"libtsnet_probe_slp{k}.so")
    if not os.path.exists(path):
        print(f"probe {k}: not built"); continue
    lib = _lib.bind(C.CDLL(path))
    outs = [run(lib) for _ in range(R)]
    bad = [int(((o - ref).abs().amax(dim=-1) > 1e-6).sum()) for o in outs]
    rr = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
    cols = torch.zeros(64, dtype=torch.long)
    for o in outs:
        d = ((o - ref).abs().amax(dim=-1) > 1e-6).view(K * B, P // 64, 64).sum(dim=(0, 1))
        cols += d
    print(f"probe {k}: positions differing from the scalar build per run {bad} of {K * B * P}; runs differing from the first: {rr} of {R - 1}; "
          f"worst |d flow| {max(float((o - ref).abs().max()) for o in outs):.3e}; differing positions by target column of the tile (0..63): "
          f"{[int(c) for c in cols]}")
