"""GPU box: run the forward a few times at a chosen per-GPU batch (full forward, or clip mode = set_sources once + forward_target) and
print its wall time -- the command tools/trace_cmd.sh wraps in rocprofv3 to get the per-dispatch table of a small-batch forward.
    python tools/forward_run.py --batch 1 [--clip] [--n-blocks 4] [--iters 20]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import synth
from wacv23_tsnet_amd.engine import TSNetEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--n-blocks", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--clip", action="store_true")
ap.add_argument("--size", type=int, default=256, help="frame height = width (512: BASELINE.json configs[4])")
ap.add_argument("--n-source", type=int, default=3)
ap.add_argument("--bf16", action="store_true", help="tsnet_cfg.operand_mode = 1 (bf16 convolution operands)")
ap.add_argument("--bf16s", action="store_true", help="tsnet_cfg.operand_mode = 2 (bf16 operands + bf16 storage of the large activations)")
a = ap.parse_args()
H = W = a.size
eng = TSNetEngine(label_nc=2, n_blocks=a.n_blocks, n_downsampling=3, n_source=a.n_source, height=H, width=W, max_batch=a.batch, operands="bf16s" if a.bf16s else ("bf16" if a.bf16 else "fp32"))
eng.load_state_dict(synth.state_dict(eng.param_shapes(), seed=0)); eng.finalize("cuda")
inp = synth.inputs(a.n_source, 2, a.batch, H, W, seed=1)
si, sl, sb, tl, tb = [[t.cuda() for t in x] if isinstance(x, list) else x.cuda() for x in inp]
if a.clip:
    eng.set_sources(si, sl, sb)
    step = lambda: eng.forward_target(tl, tb)
else:
    step = lambda: eng.forward(si, sl, sb, tl, tb)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
print(json.dumps({"batch": a.batch, "n_blocks": a.n_blocks, "clip": a.clip, "bf16": a.bf16, "size": a.size, "n_source": a.n_source, "ms": round(dt * 1e3, 3), "frames_per_s": round(a.batch / dt, 1)}))
