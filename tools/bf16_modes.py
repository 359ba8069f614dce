"""GPU box: the two bf16 modes side by side -- operand_mode 1 (bf16 operands) and 2 (+ bf16 storage of the large activations) -- at the
shapes of BASELINE.json configs[2] (n_blocks = 4, bs = 8, 256^2) and configs[4] (512^2, K = 5, one pair per GPU): frames/s and, with
--report, the distances to the oracle that rounds at the same points (tests/helpers.bf16_mode_report).
usage: bf16_modes.py [--report]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from wacv23_tsnet_amd import synth
from wacv23_tsnet_amd.engine import TSNetEngine

CASES = {"cfg2": (dict(label_nc=2, n_blocks=4, n_downsampling=3, n_source=3), 8, 256, 256), "cfg4": (dict(label_nc=2, n_blocks=0, n_downsampling=3, n_source=5), 1, 512, 512)}
out = {}
for tag, (kw, B, H, W) in CASES.items():
    for mode in ("bf16", "bf16s"):
        eng = TSNetEngine(height=H, width=W, max_batch=B, operands=mode, **kw)
        eng.load_state_dict(synth.state_dict(eng.param_shapes(), seed=0)); eng.finalize("cuda")
        inp = [[t.cuda() for t in x] if isinstance(x, list) else x.cuda() for x in synth.inputs(kw["n_source"], 2, B, H, W, seed=3)]
        for _ in range(5): eng.forward(*inp)
        torch.cuda.synchronize(); n = 30; t0 = time.perf_counter()
        for _ in range(n): eng.forward(*inp)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        out[f"{tag}_{mode}"] = {"ms_per_step": round(dt * 1e3, 3), "frames_per_s": round(B / dt, 1)}
        eng.close(); del eng; torch.cuda.empty_cache()
print(json.dumps(out))
if "--report" in sys.argv:
    import helpers as Hh
    from oracle import tsnet_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    for tag, cfg, B, H, W, ws, isd, mask in (("cfg2", O.TSNetConfig(label_nc=2, n_blocks=4, n_source=3), 8, 256, 256, 21, 22, "box"),
                                             ("cfg2", O.TSNetConfig(label_nc=2, n_blocks=4, n_source=3), 8, 256, 256, 31, 32, "box"),
                                             ("cfg4", O.TSNetConfig(label_nc=2, n_blocks=0, n_source=5), 1, 512, 512, 25, 26, "bernoulli"),
                                             ("cfg4", O.TSNetConfig(label_nc=2, n_blocks=0, n_source=5), 1, 512, 512, 35, 36, "bernoulli")):
        sd = O.synth_state_dict(cfg, seed=ws, bias_std=0.02)
        inp = O.synth_inputs(cfg, B, H, W, seed=isd, mask_mode=mask)
        eng = Hh.make_engine(cfg, sd, H, W, B, "cuda", operands="bf16s")
        rec, _ = Hh.run_engine(eng, inp, "cuda")
        r = Hh.bf16_mode_report(eng, cfg, sd, inp, rec, B, "cuda", mode="bf16s")
        eng.close()
        print(f"[{tag} bf16s w{ws} i{isd}] " + " ".join(f"{k}={v:.3e}" for k, v in r.items()), flush=True)
