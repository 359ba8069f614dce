"""Diagnostic (GPU box): where does the HIP path's deviation from the fp32 oracle sit relative to the
oracle's own fp32 rounding noise?  Compares GPU fp32, oracle fp32 and oracle fp64 ("truth") at cfg0."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/ (helpers)
import torch
import helpers as Hh
from oracle import tsnet_oracle as O

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.set_num_threads(threads)
cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=3)
sd = O.synth_state_dict(cfg, seed=0)
inp = O.synth_inputs(cfg, B, 256, 256, seed=1)
t = time.time(); r32 = O.tsnet_forward(sd, cfg, *inp, want_stages=True); t32 = time.time() - t
sd64 = {k: v.double() for k, v in sd.items()}
inp64 = [[x.double() for x in a] if isinstance(a, list) else a.double() for a in inp]
t = time.time(); r64 = O.tsnet_forward(sd64, cfg, *inp64, want_stages=True); t64 = time.time() - t
eng = Hh.make_engine(cfg, sd, 256, 256, B, "cuda")
rec, flows = Hh.run_engine(eng, inp, "cuda")
def st(name):
    return Hh.nhwc_to_nchw(eng.stage(name, "cuda").cpu())
rows = {}
def cmp(name, g, a32, a64):
    g = g.double(); a32 = a32.double()
    rows[name] = dict(gpu_vs_f32=(g - a32).abs().max().item(), gpu_vs_f64=(g - a64).abs().max().item(), f32_vs_f64=(a32 - a64).abs().max().item(),
                      mean_gpu_vs_f64=(g - a64).abs().mean().item(), mean_f32_vs_f64=(a32 - a64).abs().mean().item(), absmax=a64.abs().max().item())
cmp("rec", rec, r32["rec_tar_img"], r64["rec_tar_img"])
for i in range(3):
    cmp(f"flow{i}", flows[i], r32["flows"][i], r64["flows"][i])
src = st("src_fea")
for i in range(3):
    cmp(f"src_fea{i}", src[i * B:(i + 1) * B], r32["stages"]["src_fea"][i], r64["stages"]["src_fea"][i])
for k in ("tar_fea", "pg", "sg", "dec_map"):
    cmp(k, st(k), r32["stages"][k], r64["stages"][k])
print(json.dumps(dict(threads=threads, t_f32=t32, t_f64=t64)))
for k, v in rows.items():
    print(f"{k:10s} " + " ".join(f"{a}={b:.3e}" for a, b in v.items()))
