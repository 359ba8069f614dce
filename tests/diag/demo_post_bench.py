"""GPU box: demo post-processing -- device kernels vs the reference's host path (oracle) per frame."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/ (helpers)
import torch
from oracle import demo_oracle as DO
from wacv23_tsnet_amd import demo
from test_demo_post import _frames
B, H, W = 4, 256, 256
rec, ref = _frames(B, H, W, 7)
recd, refd = rec.cuda(), ref.cuda()
post = demo.DemoPostprocessor(refd)
for _ in range(5): out = post(recd)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 200
for _ in range(N): out = post(recd)
torch.cuda.synchronize(); dev_ms = (time.perf_counter() - t0) / N * 1e3
host = out.cpu()                                         # what crosses PCIe: 3 bytes per pixel
rm, rs = DO.ref_statistics(ref)
t0 = time.perf_counter()
for _ in range(5):
    for b in range(B):
        frame = recd[b:b + 1].cpu()                      # the reference copies the fp32 frame to the host first
        DO.postprocess_frame(frame, rm, rs)
cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
algo_bytes = B * H * W * (2 * 12 + 3)
print(json.dumps({"frames": B, "device_ms_per_batch": round(dev_ms, 4), "host_reference_ms_per_batch": round(cpu_ms, 3),
                  "algorithmic_MB": round(algo_bytes / 1e6, 2), "achieved_GB_s": round(algo_bytes / dev_ms / 1e6, 1)}))
