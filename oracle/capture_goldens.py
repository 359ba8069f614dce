"""Capture golden vectors from the REAL reference -- authoring container only.

Runs only where /root/reference exists (never on the GPU box).  It imports the
reference model unmodified with the two shims of SURVEY.md Appendix B (stub
`torchvision`, `.cuda()` -> no-op), loads PRNG-generated weights into it through
`load_state_dict`, runs `set_test_input()` + `forward()` and writes *data only*
(inputs are regenerated from the PRNG; expected outputs are stored) under
tests/golden/.  It also asserts that oracle/tsnet_oracle.py reproduces the reference
outputs, which is what pins the oracle.

    python oracle/capture_goldens.py            # all cases (cfg0 full size takes ~1 min)
    python oracle/capture_goldens.py --skip-full
"""
from __future__ import annotations

import argparse
import io
import contextlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
THREADS = 8  # recorded in every golden; 1-vs-8-thread forwards differ by ~4e-6


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; goldens can only be captured in the authoring container")
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models          # TSNet.py:5 (VGG is train-only)
    torch.Tensor.cuda = lambda self, *a, **k: self            # TSNet.py:286-290,362
    torch.nn.Module.cuda = lambda self, *a, **k: self         # networks.py:116
    sys.path.insert(0, REF)
    import model.TSNet as ref_face
    import model.TSNet_pose as ref_pose
    return ref_face, ref_pose


def build_reference_model(ref_face, ref_pose, cfg, sd):
    from oracle.tsnet_oracle import TSNetConfig  # noqa: F401
    with contextlib.redirect_stdout(io.StringIO()):
        if cfg.pose:
            m = ref_pose.TSNet(is_train=False, label_nc=cfg.label_nc, n_blocks=cfg.n_blocks,
                               n_downsampling=cfg.n_downsampling, n_source=cfg.n_source, use_mask=cfg.use_mask)
        else:
            m = ref_face.TSNet(is_train=False, label_nc=cfg.label_nc, n_blocks=cfg.n_blocks,
                               n_downsampling=cfg.n_downsampling, n_source=cfg.n_source, return_flow=True)
    m.eval()
    for net in ("img_enc", "lbl_enc", "fuse_net", "dec"):
        sub = {k[len(net) + 1:]: v for k, v in sd.items() if k.startswith(net + ".")}
        getattr(m, net).load_state_dict(sub, strict=True)   # same call as demo_face.py:126-129
    return m


def run_case(ref_face, ref_pose, name, cfg, B, H, W, wseed, iseed, mask_mode="bernoulli", bias_std=0.0, full=False):
    from oracle import tsnet_oracle as O
    sd = O.synth_state_dict(cfg, seed=wseed, bias_std=bias_std)
    src_img, src_lbl, src_bbox, tar_lbl, tar_bbox = O.synth_inputs(cfg, B, H, W, seed=iseed, mask_mode=mask_mode)
    m = build_reference_model(ref_face, ref_pose, cfg, sd)
    m.set_test_input([x.clone() for x in src_img], src_lbl, src_bbox, tar_lbl, tar_bbox)
    with torch.no_grad():
        m.forward()
    rec = m.rec_tar_img.detach().clone()
    flows = [f.detach().clone() for f in getattr(m, "warp_grid2d_list", [])]

    # pin the oracle against the reference on this very case
    got = O.tsnet_forward(sd, cfg, src_img, src_lbl, src_bbox, tar_lbl, tar_bbox, want_stages=True)
    d_rec = (got["rec_tar_img"] - rec).abs().max().item()
    d_flow = max([(a - b).abs().max().item() for a, b in zip(got["flows"], flows)] or [0.0])
    print(f"[{name}] oracle vs reference: max|d rec|={d_rec:.3e} max|d flow|={d_flow:.3e}")
    assert d_rec <= 1e-6 and d_flow <= 1e-6, "oracle restatement diverges from the reference"

    meta = dict(name=name, B=B, H=H, W=W, wseed=wseed, iseed=iseed, mask_mode=mask_mode, bias_std=bias_std,
                threads=THREADS, torch=torch.__version__,
                cfg=dict(label_nc=cfg.label_nc, n_blocks=cfg.n_blocks, n_downsampling=cfg.n_downsampling,
                         n_source=cfg.n_source, pose=cfg.pose, use_mask=cfg.use_mask),
                oracle_vs_ref=dict(rec=d_rec, flow=d_flow))
    st = got["stages"]
    arrays = {}
    if full:
        # full-size case: summaries + crops only
        arrays["rec_crop"] = rec[:, :, 96:128, 96:128].numpy()
        arrays["rec_rowsum64"] = rec.double().sum(dim=3).numpy()           # (B,3,H) fp64 row checksums
        for i, f in enumerate(flows):
            arrays[f"flow{i}"] = f.numpy()
        summ = {}
        for k in ("tar_fea", "pg", "sg", "dec_map", "dec_fea"):
            t = st[k]
            summ[k] = dict(mean=t.double().mean().item(), absmax=t.abs().max().item(), sum64=t.double().sum().item())
        for i, t in enumerate(st["src_fea"]):
            summ[f"src_fea{i}"] = dict(mean=t.double().mean().item(), absmax=t.abs().max().item(), sum64=t.double().sum().item())
        summ["rec"] = dict(mean=rec.double().mean().item(), absmax=rec.abs().max().item(), sum64=rec.double().sum().item())
        meta["summary"] = summ
        arrays["src_fea0_crop"] = st["src_fea"][0][:, :16, :8, :8].numpy()
        arrays["tar_fea_crop"] = st["tar_fea"][:, :16, :8, :8].numpy()
        arrays["pg_crop"] = st["pg"][:, :16, :8, :8].numpy()
        arrays["sg_crop"] = st["sg"][:, :16, :8, :8].numpy()
    else:
        arrays["rec"] = rec.numpy()
        for i, f in enumerate(flows):
            arrays[f"flow{i}"] = f.numpy()
        # stage tensors come from the oracle, which the assert above ties to the reference
        arrays["tar_fea"] = st["tar_fea"].numpy()
        arrays["src_fea0"] = st["src_fea"][0].numpy()
        arrays["pg"] = st["pg"].numpy()
        arrays["sg"] = st["sg"].numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), meta=json.dumps(meta), **arrays)
    return meta


def run_train_case(ref_face, ref_pose, name, cfg, B, wseed, iseed):
    """Training-mode forward extras: the reference model is built in test mode (no
    discriminators / VGG), then driven through `set_train_input` + `forward` with `is_train` switched on --
    exactly the extra branches TSNet.py:327-331, 372-390, 402-405.  256x256 only (F.fold(.., 256, ..), :379)."""
    from oracle import tsnet_oracle as O
    H = W = 256
    sd = O.synth_state_dict(cfg, seed=wseed)
    src_img, src_lbl, src_bbox, tar_lbl, tar_bbox = O.synth_inputs(cfg, B, H, W, seed=iseed, mask_mode="box")
    tar_img = O.synth_inputs(cfg, B, H, W, seed=iseed + 1000, mask_mode="box")[0][0]      # another image tensor as the target frame
    m = build_reference_model(ref_face, ref_pose, cfg, sd)
    m.is_train = True
    m.set_train_input([x.clone() for x in src_img], src_lbl, src_bbox, tar_img.clone(), tar_lbl, tar_bbox)
    with torch.no_grad():
        m.forward()
    warp = [t.detach().clone() for t in m.warp_src_img_list]
    lw = float(m.loss_warp)
    la = None if cfg.pose else float(m.loss_align)                 # TSNet_pose.py has no alignment loss
    got = O.tsnet_forward(sd, cfg, src_img, src_lbl, src_bbox, tar_lbl, tar_bbox, tar_img=tar_img)
    tr = got["train"]
    d_warp = max((a - b).abs().max().item() for a, b in zip(tr["warp_src_img_list"], warp))
    d_lw = abs(float(tr["loss_warp"]) - lw)
    d_la = 0.0 if cfg.pose else abs(float(tr["loss_align"]) - la)
    assert not cfg.pose or tr["loss_align"] is None
    d_rec = (got["rec_tar_img"] - m.rec_tar_img).abs().max().item()
    print(f"[{name}] oracle vs reference: max|d warp|={d_warp:.3e} d loss_warp={d_lw:.3e} d loss_align={d_la:.3e} max|d rec|={d_rec:.3e}")
    if cfg.pose:      # the composite really happened in the reference: background columns carry -mean/255
        assert all(float((t[:, :, :, :64] - t[:, :, :1, :1]).abs().max()) == 0.0 for t in warp)
    assert d_warp <= 1e-6 and d_lw <= 1e-6 and d_la <= 1e-6 and d_rec <= 1e-6, "oracle restatement diverges from the reference"
    meta = dict(name=name, B=B, H=H, W=W, wseed=wseed, iseed=iseed, mask_mode="box", bias_std=0.0, threads=THREADS,
                torch=torch.__version__, loss_warp=lw, loss_align=la,
                cfg=dict(label_nc=cfg.label_nc, n_blocks=cfg.n_blocks, n_downsampling=cfg.n_downsampling,
                         n_source=cfg.n_source, pose=cfg.pose, use_mask=cfg.use_mask),
                oracle_vs_ref=dict(warp=d_warp, loss_warp=d_lw, loss_align=d_la, rec=d_rec))
    arrays = {}
    for i, t in enumerate(warp):
        arrays[f"warp{i}_crop"] = t[:, :, 96:160, 96:160].numpy()
        arrays[f"warp{i}_rowsum64"] = t.double().sum(dim=3).numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), meta=json.dumps(meta), **arrays)
    return meta


def capture_modules(ref_face):
    """G1: standalone small Encoder / Decoder / FuseNet / ResnetBlock instances
    (they construct without init_net)."""
    from wacv23_tsnet_amd import prng
    import torch.nn as nn
    out = {}
    meta = {"threads": THREADS, "torch": torch.__version__}

    def load_prng(mod, tag, seed=5, bias_std=0.05):
        sd = {}
        for k, v in mod.state_dict().items():
            sd[k] = prng.normal(seed, f"{tag}.{k}", tuple(v.shape), 0.05 if k.endswith("weight") else bias_std)
        mod.load_state_dict(sd)
        return sd

    # Encoder with coords, ngf=8, 2 downs, 2 blocks, 32x32 input of 5 channels
    enc = ref_face.Encoder(input_nc=5, ngf=8, n_downsampling=2, n_blocks=2, addcoords=True).eval()
    load_prng(enc, "g1.enc")
    x = prng.uniform01(7, "g1.enc.x", (2, 5, 32, 32)) * 2 - 1
    with torch.no_grad():
        out["enc_y"] = enc(x).numpy()
    # Decoder ngf=8, 2 ups, 1 block
    dec = ref_face.Decoder(output_nc=3, ngf=8, n_downsampling=2, return_fea=True, n_blocks=1).eval()
    load_prng(dec, "g1.dec")
    p = prng.uniform01(7, "g1.dec.p", (2, 32, 8, 8)) * 2 - 1
    s = prng.uniform01(7, "g1.dec.s", (2, 32, 8, 8)) * 2 - 1
    with torch.no_grad():
        y, fea = dec(p, s)
    out["dec_y"], out["dec_fea"] = y.numpy(), fea.numpy()
    # FuseNet ngf=64 (two 32-channel halves)
    fn = ref_face.FuseNet(ngf=64, n_blocks=1).eval()
    load_prng(fn, "g1.fuse")
    a = prng.uniform01(7, "g1.fuse.a", (2, 32, 8, 8)) * 2 - 1
    b = prng.uniform01(7, "g1.fuse.b", (2, 32, 8, 8))
    with torch.no_grad():
        out["fuse_y"] = fn(a, b).numpy()
    # ResnetBlock dim=16
    rb = ref_face.ResnetBlock(16, padding_type="reflect", norm_layer=nn.InstanceNorm2d).eval()
    load_prng(rb, "g1.rb")
    xr = prng.uniform01(7, "g1.rb.x", (2, 16, 12, 10)) * 2 - 1
    with torch.no_grad():
        out["rb_y"] = rb(xr).numpy()
    np.savez_compressed(os.path.join(GOLD, "g1_modules.npz"), meta=json.dumps(meta), **out)
    print("[g1_modules] captured", {k: v.shape for k, v in out.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only-train", action="store_true", help="(re)capture only the training-mode extras golden and merge it into the manifest")
    args = ap.parse_args()
    torch.set_num_threads(THREADS)
    os.makedirs(GOLD, exist_ok=True)
    ref_face, ref_pose = import_reference()
    from oracle.tsnet_oracle import TSNetConfig

    train_case = lambda: run_train_case(ref_face, ref_pose, "g5_train_extras_256_k2", TSNetConfig(label_nc=2, n_blocks=0, n_source=2), 1, 11, 12)
    train_case_pose = lambda: run_train_case(ref_face, ref_pose, "g5_train_extras_pose_256_k2", TSNetConfig(label_nc=25, n_blocks=0, n_source=2, pose=True), 1, 13, 14)
    if args.only_train:
        mpath = os.path.join(GOLD, "MANIFEST.json")
        metas = [m for m in json.load(open(mpath)) if m["name"] not in ("g5_train_extras_256_k2", "g5_train_extras_pose_256_k2")]
        metas.append(train_case())
        metas.append(train_case_pose())
        with open(mpath, "w") as f:
            json.dump(metas, f, indent=1)
        return

    capture_modules(ref_face)
    metas = []
    # G2/G3: whole-forward cases at small spatial size, full channel widths
    metas.append(run_case(ref_face, ref_pose, "g3_face_64_k2_nb0", TSNetConfig(label_nc=2, n_blocks=0, n_source=2), 2, 64, 64, 0, 1))
    metas.append(run_case(ref_face, ref_pose, "g3_face_64_k2_nb1_bias", TSNetConfig(label_nc=2, n_blocks=1, n_source=2), 2, 64, 64, 2, 3, bias_std=0.02))
    metas.append(run_case(ref_face, ref_pose, "g2_face_64_softmask", TSNetConfig(label_nc=2, n_blocks=0, n_source=2), 1, 64, 64, 0, 4, mask_mode="soft"))
    metas.append(run_case(ref_face, ref_pose, "g2_face_64_ones", TSNetConfig(label_nc=2, n_blocks=0, n_source=1), 1, 64, 64, 0, 5, mask_mode="ones"))
    metas.append(run_case(ref_face, ref_pose, "g2_face_64_zeros", TSNetConfig(label_nc=2, n_blocks=0, n_source=1), 1, 64, 64, 0, 6, mask_mode="zeros"))
    metas.append(run_case(ref_face, ref_pose, "g2_face_32_k3", TSNetConfig(label_nc=2, n_blocks=0, n_source=3), 1, 32, 32, 0, 7, mask_mode="box"))
    metas.append(run_case(ref_face, ref_pose, "g3_face_128x64_k2", TSNetConfig(label_nc=2, n_blocks=0, n_source=2), 1, 128, 64, 0, 8, mask_mode="box"))
    # pose variant (L=25, composite needs 256x256 output)
    metas.append(run_case(ref_face, ref_pose, "g3_pose_256_k1_nb1", TSNetConfig(label_nc=25, n_blocks=1, n_source=1, pose=True), 1, 256, 256, 9, 10, mask_mode="box", bias_std=0.02, full=True))
    if not args.skip_full:
        # G4: cfg0 -- TSNet(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3), B=4, 256x256
        metas.append(run_case(ref_face, ref_pose, "g4_cfg0_full", TSNetConfig(label_nc=2, n_blocks=0, n_source=3), 4, 256, 256, 0, 1, full=True))
    metas.append(train_case())
    metas.append(train_case_pose())
    with open(os.path.join(GOLD, "MANIFEST.json"), "w") as f:
        json.dump(metas, f, indent=1)


if __name__ == "__main__":
    main()
