"""CPU tier: the WHOLE forward schedule of the engine (every kernel, every launch geometry, the arena,
clip mode) executed under the fiber emulator on a narrow net (ngf=8) and compared with the oracle.
Numerics on real hardware at full width are the GPU tier's job (tests/test_gpu_forward.py)."""
import ctypes as C

import pytest
import torch

import helpers as Hh
from oracle import tsnet_oracle as O


def _case(K=2, nb=1, L=2, B=2, H=32, W=32, enc_blocks=2, pose=False, wscale=4.0, bias_std=0.02, mask="box", seed=3):
    cfg = O.TSNetConfig(label_nc=L, n_blocks=nb, n_source=K, ngf=8, enc_blocks=enc_blocks, fuse_ngf=128, pose=pose)
    sd = O.synth_state_dict(cfg, seed=seed, bias_std=bias_std)
    sd = {k: (v * wscale if k.endswith("weight") else v) for k, v in sd.items()}   # non-trivial activations at width 8
    inp = O.synth_inputs(cfg, B, H, W, seed=seed + 1, mask_mode=mask)
    return cfg, sd, inp


@pytest.mark.parametrize("kw", [dict(K=2, nb=1), dict(K=3, nb=0, mask="bernoulli"), dict(K=1, nb=2, L=5, H=48, W=32, mask="soft")])
def test_forward_matches_oracle(emu_lib, kw):
    cfg, sd, inp = _case(**kw)
    B, H, W = inp[3].shape[0], inp[3].shape[2], inp[3].shape[3]
    ref = O.tsnet_forward(sd, cfg, *inp, want_stages=True)
    eng = Hh.make_engine(cfg, sd, H, W, B, "cpu", lib=emu_lib)
    emu_lib.tsnet_debug_counters(None, 1)      # reset the process-wide launch counters
    rec, flows = Hh.run_engine(eng, inp, "cpu")
    assert (rec - ref["rec_tar_img"]).abs().max().item() < 5e-4
    for a, b in zip(flows, ref["flows"]):
        assert (a - b).abs().max().item() < 1e-4
    rep = Hh.stage_report(eng, ref["stages"], cfg.n_source, B, "cpu")
    assert max(rep[k] for k in rep if k.startswith("src_fea")) < 1e-4 and rep["tar_fea"] < 1e-4 and rep["sg"] < 1e-4 and rep["pg"] < 2e-3
    assert max(rep[f"dec_up{i}"] for i in range(cfg.n_downsampling)) < 2e-3         # downstream of pg (flow error x feature gradient)
    cnt = (C.c_int64 * 4)()
    emu_lib.tsnet_debug_counters(cnt, 1)
    # 32-wide frames: the 8-channel stems (label_nc = 2: both; 5: the label encoder's) and the last decoder up-convolution split into
    # 4 x 32 rectangles (patch kernels); everything at the lower resolutions runs on the general kernel
    assert cnt[1] > 0 and cnt[0] == (3 if cfg.label_nc == 2 else 2) and cnt[2] == 1
    eng.close()


def test_clip_mode_and_determinism(emu_lib):
    cfg, sd, inp = _case(K=2, nb=0)
    eng = Hh.make_engine(cfg, sd, 32, 32, 2, "cpu", lib=emu_lib)
    r1, f1 = Hh.run_engine(eng, inp, "cpu")
    r2, _ = Hh.run_engine(eng, inp, "cpu")
    assert torch.equal(r1, r2)
    eng.set_sources(inp[0], inp[1], inp[2])
    r3, f3 = eng.forward_target(inp[3], inp[4], return_flow=True)
    assert torch.equal(r1, r3) and all(torch.equal(a, b) for a, b in zip(f1, f3))
    # a second driving frame against the cached sources == the one-shot forward on that frame
    inp2 = O.synth_inputs(cfg, 2, 32, 32, seed=99, mask_mode="box")
    r4, _ = eng.forward_target(inp2[3], inp2[4])
    r5, _ = Hh.run_engine(eng, (inp[0], inp[1], inp[2], inp2[3], inp2[4]), "cpu")
    assert torch.equal(r4, r5)
    eng.close()


def test_batch_items_independent(emu_lib):
    cfg, sd, inp = _case(K=2, nb=0, B=3)
    eng = Hh.make_engine(cfg, sd, 32, 32, 3, "cpu", lib=emu_lib)
    full, _ = Hh.run_engine(eng, inp, "cpu")
    sub = ([x[1:2] for x in inp[0]], [x[1:2] for x in inp[1]], [x[1:2] for x in inp[2]], inp[3][1:2], inp[4][1:2])
    one, _ = Hh.run_engine(eng, sub, "cpu")
    assert torch.equal(one, full[1:2])          # per-sample statistics, scales and tiles: the same bits in any batch
    eng.close()


def test_pose_composite(emu_lib):
    cfg, sd, inp = _case(K=1, nb=0, L=5, B=1, H=256, W=256, enc_blocks=0, pose=True)
    ref = O.tsnet_forward(sd, cfg, *inp)
    eng = Hh.make_engine(cfg, sd, 256, 256, 1, "cpu", lib=emu_lib)
    rec, _ = Hh.run_engine(eng, inp, "cpu", return_flow=False)
    # 4x weights on a 256 x 256 frame: an ill-conditioned map (the fp32 oracle itself is 3.4e-4 from its fp64 evaluation); the engine must
    # be as close to the fp64 result as the fp32 oracle is, and within the north-star 1e-3 of the fp32 oracle
    r64 = O.tsnet_forward({k: v.double() for k, v in sd.items()}, cfg, *[[t.double() for t in x] if isinstance(x, list) else x.double() for x in inp])["rec_tar_img"]
    assert (rec.double() - r64).abs().max().item() <= (ref["rec_tar_img"].double() - r64).abs().max().item() + 1e-4
    assert (rec - ref["rec_tar_img"]).abs().max().item() < 1e-3
    assert torch.equal(rec[..., :64], ref["rec_tar_img"][..., :64]) and torch.equal(rec[..., 192:], ref["rec_tar_img"][..., 192:])
    eng.close()


@pytest.mark.parametrize("H,W,B", [(64, 64, 1), (96, 64, 2)])
def test_conv_epilogue_statistics_path(emu_lib, H, W, B):
    """64x64 input -> 8x8 features: every image is one ragged 128-position tile at the feature resolution, two / eight / 32 whole tiles above
    it: the in-kernel statistics finalize (at most 32 tiles per image; the 64x64 stem is exactly 32, four batches of the fold's eight-entry
    loads).  96x64, two images: the stem and the last up-convolution have 48 tiles per image -- the in_finalize2 launch with a ragged last
    batch of its 16 groups (test_pose_composite's 256 x 256 frame runs it on 512 / 128 / 64 partials)."""
    cfg, sd, inp = _case(K=1, nb=1 if B == 1 else 0, B=B, H=H, W=W, enc_blocks=1 if B == 1 else 0)     # (the two-image case: the stem / up-convolution layers are its point)
    ref = O.tsnet_forward(sd, cfg, *inp, want_stages=True)
    eng = Hh.make_engine(cfg, sd, H, W, B, "cpu", lib=emu_lib)
    rec, _ = Hh.run_engine(eng, inp, "cpu", return_flow=False)
    assert (rec - ref["rec_tar_img"]).abs().max().item() < 5e-4
    rep = Hh.stage_report(eng, ref["stages"], cfg.n_source, B, "cpu")
    assert max(rep[k] for k in rep if k.startswith("src_fea")) < 1e-4 and rep["tar_fea"] < 1e-4 and rep["sg"] < 1e-4
    eng.close()


@pytest.mark.parametrize("pose", [False])
def test_vector_rgb_head(emu_lib, pose):
    """ngf=16: head_conv3_kernel (the form every reference-width model runs); ragged 48x40 frames exercise its partial tiles and
    reflection.  (Its composite epilogue needs a 256 x 256 pose forward -- 47 s of emulation: held by the GPU tier's pose goldens at
    the reference width, tests/test_gpu_forward.py::test_pose_golden_composite.)"""
    H, W = (256, 256) if pose else (48, 40)
    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=1, ngf=16, enc_blocks=0, fuse_ngf=256, pose=pose)
    sd = O.synth_state_dict(cfg, seed=8, bias_std=0.02)
    sd = {k: (v * 3 if k.endswith("weight") else v) for k, v in sd.items()}
    inp = O.synth_inputs(cfg, 1, H, W, seed=9, mask_mode="box")
    ref = O.tsnet_forward(sd, cfg, *inp)["rec_tar_img"]
    eng = Hh.make_engine(cfg, sd, H, W, 1, "cpu", lib=emu_lib)
    rec, _ = Hh.run_engine(eng, inp, "cpu", return_flow=False)
    assert (rec - ref).abs().max().item() < 5e-4
    if pose:
        assert torch.equal(rec[..., :64], ref[..., :64])
    eng.close()


def test_patch_kernel_schedule(emu_lib):
    """32 x 256 frames give 4 x 32 feature maps -- one patch tile of conv_h2.hpp -- so every 3x3 layer of the forward runs on the patch
    kernels (h2s stems, h2d downsampling, h2 ResnetBlock / FuseNet / decoder), each applying its producer's InstanceNorm + ReLU while staging;
    only the two 1x1 convolutions run on the general kernel.  Must match the oracle and stay bit-identical between the one-shot forward and
    clip mode."""
    cfg = O.TSNetConfig(label_nc=2, n_blocks=1, n_source=1, ngf=32, enc_blocks=1, fuse_ngf=512)
    sd = O.synth_state_dict(cfg, seed=14, bias_std=0.02)
    sd = {k: (v * 2 if k.endswith("weight") else v) for k, v in sd.items()}
    inp = O.synth_inputs(cfg, 1, 32, 256, seed=15, mask_mode="box")
    ref = O.tsnet_forward(sd, cfg, *inp, want_stages=True)
    eng = Hh.make_engine(cfg, sd, 32, 256, 1, "cpu", lib=emu_lib)
    emu_lib.tsnet_debug_counters(None, 1)
    rec, flows = Hh.run_engine(eng, inp, "cpu")
    cnt = (C.c_int64 * 4)()
    emu_lib.tsnet_debug_counters(cnt, 1)
    # patch kernels: 2 stems + 2 (the 128-channel downsampling layer of each encoder) + 2 (encoder block) + 2 (halves of fuse conv1)
    # + 1 (fuse conv2) + 2 (decoder block) + 3 (decoder up-convolutions) = 14; general kernel: the 32- and 64-channel downsampling layers
    # (2 x 2), fuse_net.conv and dec.map_conv
    assert cnt[0] == 14 and cnt[1] == 6 and cnt[2] == 1
    assert (rec - ref["rec_tar_img"]).abs().max().item() < 5e-4
    rep = Hh.stage_report(eng, ref["stages"], cfg.n_source, 1, "cpu")
    assert max(rep[k] for k in rep if k.startswith("src_fea")) < 1e-4 and rep["tar_fea"] < 1e-4 and rep["sg"] < 1e-4
    assert max(rep[k] for k in rep if k.startswith("dec_up")) < 2e-3
    eng.set_sources(inp[0], inp[1], inp[2])
    r2, _ = eng.forward_target(inp[3], inp[4])
    assert torch.equal(rec, r2)
    eng.close()


def test_h2_samples_do_not_see_their_batch(emu_lib):
    """The fp16 operand scales that come from the data (packed inputs, decoder stream) are taken per image: a sample's result must be the
    same bits whether its neighbour in the batch is ordinary or has a 40x larger input range (the basis of batch sharding across GPUs)."""
    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=1, ngf=32, enc_blocks=0, fuse_ngf=512)
    sd = O.synth_state_dict(cfg, seed=14, bias_std=0.02)
    sd = {k: (v * 2 if k.endswith("weight") else v) for k, v in sd.items()}
    inp = O.synth_inputs(cfg, 2, 32, 256, seed=16, mask_mode="box")
    inp[0][0][1] *= 40.0                                     # second sample: source image far outside [0, 255]
    eng = Hh.make_engine(cfg, sd, 32, 256, 2, "cpu", lib=emu_lib)
    emu_lib.tsnet_debug_counters(None, 1)
    rec2, _ = Hh.run_engine(eng, inp, "cpu", return_flow=False)
    cnt = (C.c_int64 * 4)()
    emu_lib.tsnet_debug_counters(cnt, 1)
    assert cnt[0] > 0                                        # patch kernels ran
    calm = [[t.clone() for t in x] if isinstance(x, list) else x.clone() for x in inp]
    calm[0][0][1] /= 40.0                                    # the same first sample next to an ordinary neighbour
    rec2b, _ = Hh.run_engine(eng, calm, "cpu", return_flow=False)
    assert torch.equal(rec2[:1], rec2b[:1]) and not torch.equal(rec2[1:], rec2b[1:])
    # alone (B = 1) too: the forward takes the same kernel, tile and accumulation order in every batch (round 5)
    one, _ = Hh.run_engine(eng, [[t[:1] for t in x] if isinstance(x, list) else x[:1] for x in calm], "cpu", return_flow=False)
    assert torch.equal(one, rec2b[:1])
    eng.close()


def test_bf16_operand_mode(emu_lib):
    """tsnet_cfg.operand_mode = 1 (BASELINE.json configs[2] / [4]): every convolution reads ONE bf16 plane of its input and of its weights
    (one product, the transform fused), fp32 accumulate."""
    cfg = O.TSNetConfig(label_nc=2, n_blocks=1, n_source=1, ngf=32, enc_blocks=1, fuse_ngf=512)
    sd = O.synth_state_dict(cfg, seed=14, bias_std=0.02)
    sd = {k: (v * 2 if k.endswith("weight") else v) for k, v in sd.items()}
    inp = O.synth_inputs(cfg, 1, 32, 256, seed=15, mask_mode="box")
    eng = Hh.make_engine(cfg, sd, 32, 256, 1, "cpu", lib=emu_lib, operands="bf16")
    emu_lib.tsnet_debug_counters(None, 1)
    rec, flows = Hh.run_engine(eng, inp, "cpu")
    cnt = (C.c_int64 * 4)()
    emu_lib.tsnet_debug_counters(cnt, 1)
    assert cnt[0] == 12 and cnt[1] == 8          # bf16 operands: the stride-2 layers take the general kernel (engine.cpp conv_class)
    r = Hh.bf16_mode_report(eng, cfg, sd, inp, rec, 1, "cpu", flows=flows)
    print("[bf16 mode] " + " ".join(f"{k}={v:.3e}" for k, v in r.items()))
    assert r["src_fea"] < 5e-2 and r["tar_fea"] < 2e-2 and r["sg"] < 8e-2          # bf16 flip noise on a +-18 range
    assert r["flow_on_engine_features"] < 1e-4 and r["pg_on_engine_features"] < 4e-3   # the transformation branch on the engine's own features: fp32-class
    assert r["decoder_on_engine_features"] < 3e-2
    assert r["end_to_end_vs_fp32_oracle"] > 10 * r["decoder_on_engine_features"]    # it IS a bf16 computation
    eng.set_sources(inp[0], inp[1], inp[2])
    r2, _ = eng.forward_target(inp[3], inp[4])
    assert torch.equal(rec, r2)
    eng.close()


def test_bf16_storage_mode(emu_lib):
    """tsnet_cfg.operand_mode = 2: bf16 operands AND bf16 storage of the large convolution-to-convolution activations (the encoders' stem /
    first down-convolution outputs, the decoder's upsampled inputs and up-convolution outputs): the producers round once where they store
    (statistics from the fp32 accumulators), the consumers -- general, stride-2 patch and 3x3 patch kernels, upsample, RGB head -- widen
    exactly.  Against the oracle that rounds at the same points; and different from mode 1 (it IS another computation)."""
    cfg = O.TSNetConfig(label_nc=2, n_blocks=1, n_source=1, ngf=32, enc_blocks=1, fuse_ngf=512)
    sd = O.synth_state_dict(cfg, seed=14, bias_std=0.02)
    sd = {k: (v * 2 if k.endswith("weight") else v) for k, v in sd.items()}
    inp = O.synth_inputs(cfg, 1, 32, 256, seed=15, mask_mode="box")
    eng = Hh.make_engine(cfg, sd, 32, 256, 1, "cpu", lib=emu_lib, operands="bf16s")
    rec, flows = Hh.run_engine(eng, inp, "cpu")
    r = Hh.bf16_mode_report(eng, cfg, sd, inp, rec, 1, "cpu", mode="bf16s", flows=flows)
    print("[bf16 storage mode] " + " ".join(f"{k}={v:.3e}" for k, v in r.items()))
    assert r["src_fea"] < 5e-2 and r["tar_fea"] < 2e-2 and r["sg"] < 8e-2
    assert r["flow_on_engine_features"] < 1e-4 and r["pg_on_engine_features"] < 4e-3
    assert r["decoder_on_engine_features"] < 3e-2
    e1 = Hh.make_engine(cfg, sd, 32, 256, 1, "cpu", lib=emu_lib, operands="bf16")
    rec1, _ = Hh.run_engine(e1, inp, "cpu")
    assert not torch.equal(rec, rec1)
    e1.close()
    eng.set_sources(inp[0], inp[1], inp[2])
    r2, _ = eng.forward_target(inp[3], inp[4])
    assert torch.equal(rec, r2)
    eng.close()


def test_wide_frame_runs_the_feature_resolution_kernels(emu_lib):
    """A 32 x 256 frame (features 4 x 32): here the ResnetBlocks, FuseNet and the first decoder up-convolution run the Winograd-along-x kernel
    (conv_w1.hpp), the stride-2 layers their patch kernel -- the kernels of the full-size forward, which 32 x 32 frames never reach.  Against
    the oracle; and, since a layer has one packed form and one kernel in every batch, a frame run alone equals its copy inside the batch on
    everything those kernels produce (the layers left on the direct kernel take other tiles for a single frame: agreement to rounding)."""
    cfg, sd, inp = _case(K=2, nb=1, B=2, H=32, W=256, enc_blocks=1, seed=5)
    ref = O.tsnet_forward(sd, cfg, *inp, want_stages=True)
    eng = Hh.make_engine(cfg, sd, 32, 256, 2, "cpu", lib=emu_lib)
    emu_lib.tsnet_debug_counters(None, 1)
    rec, flows = Hh.run_engine(eng, inp, "cpu")
    cnt = (C.c_int64 * 4)()
    emu_lib.tsnet_debug_counters(cnt, 1)
    assert cnt[3] == 34064, cnt[3]                       # the ResnetBlock layers ran conv_w1 (4 x 32 pixels x 64 channels, Winograd form)
    assert (rec - ref["rec_tar_img"]).abs().max().item() < 5e-4
    for a, b in zip(flows, ref["flows"]):
        assert (a - b).abs().max().item() < 1e-4
    rep = Hh.stage_report(eng, ref["stages"], cfg.n_source, 2, "cpu")
    assert max(rep[k] for k in rep if k.startswith("src_fea")) < 1e-4 and rep["tar_fea"] < 1e-4 and rep["sg"] < 1e-4
    src_batch = Hh.nhwc_to_nchw(eng.stage("src_fea", "cpu").cpu())
    sub = ([x[1:2] for x in inp[0]], [x[1:2] for x in inp[1]], [x[1:2] for x in inp[2]], inp[3][1:2], inp[4][1:2])
    one, _ = Hh.run_engine(eng, sub, "cpu")
    src_one = Hh.nhwc_to_nchw(eng.stage("src_fea", "cpu").cpu())
    assert torch.equal(one, rec[1:2])                     # a frame alone = the same frame in a batch, bit for bit
    assert torch.equal(src_one[0], src_batch[1]) and torch.equal(src_one[1], src_batch[3])       # image index s * B + b
    eng.close()
