"""GPU box: what the side lane buys and what the kernel boundaries cost, un-traced.  Times (a) `forward` (two lanes: the driving-frame chain
and the transformation branch run beside the source encoder / the synthesis branch) and (b) `set_sources` + `forward_target` (one lane:
the same kernels, every one after the other, bit-identical results) at the bench workload.  rocprofv3's kernel trace serialises
dispatches, so the sum of the traced kernel durations (`busy` in profiles/round4_kernel_trace_stats.txt) is the kernel time of (b):
(b) - busy = what ~70 kernel boundaries cost on one lane; (b) - (a) = what the second lane hides.
    python tools/lane_overlap.py [--batch 4] [--iters 100]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import synth
from wacv23_tsnet_amd.engine import TSNetEngine
from wacv23_tsnet_amd.dist import build_replica

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
eng = TSNetEngine(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3, height=256, width=256, max_batch=a.batch)
build_replica(eng, synth.state_dict(eng.param_shapes(), seed=0), dev, src=0)
inp = [[t.to(dev) for t in x] if isinstance(x, list) else x.to(dev) for x in synth.inputs(3, 2, a.batch, 256, 256, seed=3)]
src_img, src_lbl, src_bbox, tar_lbl, tar_bbox = inp[:5]


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.iters * 1e3


def one_lane():
    eng.set_sources(src_img, src_lbl, src_bbox)
    return eng.forward_target(tar_lbl, tar_bbox)


two = timed(lambda: eng.forward(*inp))
one = timed(one_lane)
same = torch.equal(eng.forward(*inp)[0], one_lane()[0])
print(json.dumps({"batch": a.batch, "two_lanes_ms": round(two, 3), "one_lane_ms": round(one, 3), "second_lane_hides_ms": round(one - two, 3), "bit_identical": bool(same)}))
