"""GPU box: cost of the training-mode extras at the cfg1 shapes, device vs the oracle's host path."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
import torch
from oracle import tsnet_oracle as O
from wacv23_tsnet_amd.engine import TSNetEngine
B, H, W = 4, 256, 256
cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3)
eng = TSNetEngine(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3, height=H, width=W, max_batch=B)
eng.load_state_dict(O.synth_state_dict(cfg, seed=0)); eng.finalize("cuda")
inp = O.synth_inputs(cfg, B, H, W, seed=1)
tar_img = O.synth_inputs(cfg, B, H, W, seed=1001)[0][0]
si, sl, sb, tl, tb = [[t.cuda() for t in x] if isinstance(x, list) else x.cuda() for x in inp]
ti = tar_img.cuda()
rec, flows = eng.forward(si, sl, sb, tl, tb, return_flow=True)
for _ in range(3): out = eng.train_extras(si, ti)
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 100
for _ in range(N): out = eng.train_extras(si, ti)
torch.cuda.synchronize(); dev_ms = (time.perf_counter() - t0) / N * 1e3
pg = eng.stage("pg", "cuda").permute(0, 3, 1, 2).contiguous().cpu(); sg = eng.stage("sg", "cuda").permute(0, 3, 1, 2).contiguous().cpu()
fl = [f.cpu() for f in flows]; simg = [x / 255.0 for x in inp[0]]; timg = tar_img / 255.0
torch.set_num_threads(16)
t0 = time.perf_counter()
for _ in range(3): ref = O.train_extras(simg, timg, fl, pg, sg)
cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
d = max((a.cpu() - b).abs().max().item() for a, b in zip(out[0], ref["warp_src_img_list"]))
print(json.dumps({"device_ms": round(dev_ms, 4), "host_oracle_ms_16_threads": round(cpu_ms, 2), "max_abs_delta_warp_img": d,
                  "loss_warp": [out[1].item(), float(ref["loss_warp"])], "loss_align": [out[2].item(), float(ref["loss_align"])]}))
