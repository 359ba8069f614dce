"""CPU only: random TS-Net configurations and frame sizes through the ENGINE on the emulation build (tests/conftest.py build_emu_lib) against
the oracle (oracle/tsnet_oracle.py) -- the checker, as in tests/test_emu_forward.py.  The fixed tests name a handful of configurations; this
walks the engine's own kernel / tile / chunk / lane choices over ones nobody named: label counts 2..25, 0..2 ResnetBlocks, 1..4 sources, batch
1..3, frames from 32 x 32 to 64 x 96 with sides that are not powers of two, the pose variant (its composite exists at 256 x 256 only and has its own test), three mask models.
A refusal at engine creation (ArgError with a message) is a loud failure and is listed, not counted as wrong.
With `bf16` / `bf16s` as third argument the engine runs tsnet_cfg.operand_mode 1 / 2 (BASELINE.json configs[2], [4]) and is held to what
tests/test_emu_forward.py holds those modes to (helpers.bf16_mode_report: stages against the oracle that rounds the same operands, the
transformation branch and the decoder on the engine's own features).  Every case also runs in clip mode (set_sources + forward_target: the
demo's call pattern) and must give the one-shot forward's bits.
    python tools/probes/forward_fuzz.py [cases] [seed] [bf16|bf16s]"""
import ctypes
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import conftest
    import helpers as Hh
    from oracle import tsnet_oracle as O
    from wacv23_tsnet_amd import _lib
    lib = _lib.bind(ctypes.CDLL(conftest.build_emu_lib()))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
    bad, refused = [], []
    t0 = time.time()
    for i in range(n):
        pose = rng.random() < 0.3
        kw = dict(label_nc=rng.choice([2, 3, 5, 8, 25]) if pose else rng.choice([2, 3, 5, 8]), n_blocks=rng.randint(0, 2), n_source=rng.randint(1, 4),
                  ngf=rng.choice([8, 16]), enc_blocks=rng.randint(0, 2), pose=pose,
                  use_mask=False)                   # the composite is written for 256 x 256 frames (TSNet_pose.py:416): tests/test_emu_forward.py test_pose_composite
        kw["fuse_ngf"] = 16 * kw["ngf"]             # FuseNet runs on cat(source, target) features: 2 x ngf x 2^3 channels (TSNet.py:227 at ngf = 64)
        B, H, W = rng.randint(1, 3), rng.choice([32, 48, 64]), rng.choice([32, 64, 96])
        if rng.random() < 0.3:                      # full-width features (256 / 512 channels on maps of 4 x 32 and 8 x 32): the patch kernels and conv_w1
            kw.update(ngf=rng.choice([32, 64]), n_source=rng.randint(1, 2), enc_blocks=rng.randint(0, 1), n_blocks=rng.randint(0, 1))
            kw["fuse_ngf"] = 16 * kw["ngf"]
            B, H, W = rng.randint(1, 2), rng.choice([32, 64]), 256
        mask = rng.choice(["box", "bernoulli", "soft"])
        desc = dict(kw, B=B, H=H, W=W, mask=mask)
        cfg = O.TSNetConfig(**kw)
        sd = O.synth_state_dict(cfg, seed=40 + i, bias_std=0.02)
        sd = {k: (v * (4.0 if mode == "fp32" else 2.0) if k.endswith("weight") else v) for k, v in sd.items()}
        inp = O.synth_inputs(cfg, B, H, W, seed=140 + i, mask_mode=mask)
        ref = O.tsnet_forward(sd, cfg, *inp, want_stages=False) if mode == "fp32" else None
        try:
            eng = Hh.make_engine(cfg, sd, H, W, B, "cpu", lib=lib, operands=mode)
        except Exception as ex:                     # noqa: BLE001 -- the engine's refusal, with its message
            refused.append((desc, str(ex)[:200]))
            continue
        rec, flows = Hh.run_engine(eng, inp, "cpu")
        rec2, _ = Hh.run_engine(eng, inp, "cpu")
        eng.set_sources(inp[0], inp[1], inp[2])
        rec3, _ = eng.forward_target(inp[3], inp[4])
        same = bool((rec == rec2).all()) and bool((rec == rec3).all())
        if mode == "fp32":
            d_rec = (rec - ref["rec_tar_img"]).abs().max().item()
            d_flow = max((a - b).abs().max().item() for a, b in zip(flows, ref["flows"]))
            ok = d_rec < 1e-3 and d_flow < 1e-3 and same
            line = f"d_rec {d_rec:.2e} d_flow {d_flow:.2e}"
        else:
            r = Hh.bf16_mode_report(eng, cfg, sd, inp, rec, B, "cpu", mode=mode, flows=flows)
            ok = (same and r["src_fea"] < 5e-2 and r["tar_fea"] < 2e-2 and r["sg"] < 8e-2 and r["flow_on_engine_features"] < 1e-4
                  and r["pg_on_engine_features"] < 4e-3 and r["decoder_on_engine_features"] < 3e-2 and bool(rec.isfinite().all()))
            line = " ".join(f"{k}={r[k]:.2e}" for k in ("src_fea", "tar_fea", "sg", "flow_on_engine_features", "pg_on_engine_features", "decoder_on_engine_features",
                                                            "decoder_on_engine_features_mean", "decoder_on_engine_features_p99999"))
        eng.close()
        if not ok:
            bad.append((desc, line, same))
        print(f"[{i:3d}] {desc}  {line}  twice + clip mode the same bits {same}  {'ok' if ok else 'WRONG'}", flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s: wrong {len(bad)}, refused {len(refused)}")
    for r in refused:
        print("REFUSED", *r)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
