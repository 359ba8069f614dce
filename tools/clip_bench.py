"""GPU box: clip-mode throughput: the K source frames are encoded once
(tsnet_set_sources), every driving frame then costs only tsnet_forward_target.  Same cfg1 workload as bench.py."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import synth
from wacv23_tsnet_amd.engine import TSNetEngine
B, H, W = 4, 256, 256
NB = int(sys.argv[sys.argv.index("--n-blocks") + 1]) if "--n-blocks" in sys.argv else 0
if "--batch" in sys.argv:
    B = int(sys.argv[sys.argv.index("--batch") + 1])
eng = TSNetEngine(label_nc=2, n_blocks=NB, n_downsampling=3, n_source=3, height=H, width=W, max_batch=B)
eng.load_state_dict(synth.state_dict(eng.param_shapes(), seed=0)); eng.finalize("cuda")
inp = synth.inputs(3, 2, B, H, W, seed=1)
si, sl, sb, tl, tb = [[t.cuda() for t in x] if isinstance(x, list) else x.cuda() for x in inp]
full, _ = eng.forward(si, sl, sb, tl, tb)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.set_sources(si, sl, sb)
torch.cuda.synchronize(); t_src = time.perf_counter() - t0
for _ in range(5): out, _ = eng.forward_target(tl, tb)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 30
for _ in range(N): out, _ = eng.forward_target(tl, tb)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
print(json.dumps({"batch": B, "n_blocks": NB, "set_sources_ms": round(t_src * 1e3, 3), "forward_target_ms": round(dt * 1e3, 3),
                  "clip_frames_per_s": round(B / dt, 1), "bit_identical_to_full_forward": bool(torch.equal(out, full))}))
