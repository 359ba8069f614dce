"""GPU tier: whole-forward parity of the HIP path (through the C ABI) against the committed golden
vectors captured from the reference (tests/golden, oracle/capture_goldens.py), plus size-independent
properties at the full cfg0 size.  Tolerances: final image 1e-3 max-abs (BASELINE.json north_star),
flows 1e-4, feature stages 2e-4 on a +-18 range."""
import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import tsnet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL_REC, TOL_FLOW, TOL_FEA = 1e-3, 1e-4, 2e-4
# pg = grid_sample(src_fea, flow): its error is (flow error in pixels) x (feature gradient, up to ~20 per pixel on
# the +-18 range), so a 1e-5 flow error is ~3e-3 here; the reference's own fp32 noise has the same amplification.
TOL_PG = 4e-3

SMALL = ["g3_face_64_k2_nb0", "g2_face_64_softmask", "g2_face_64_ones", "g2_face_64_zeros", "g2_face_32_k3",
         "g3_face_128x64_k2", "g3_face_64_k2_nb1_bias"]


@pytest.mark.parametrize("name", SMALL)
def test_small_goldens(name):
    meta, z, cfg, sd, inputs = Hh.golden_case(name)
    eng = Hh.make_engine(cfg, sd, meta["H"], meta["W"], meta["B"], DEV)
    rec, flows = Hh.run_engine(eng, inputs, DEV)
    d_rec = np.abs(rec.numpy() - z["rec"]).max()
    d_flow = max(np.abs(flows[i].numpy() - z[f"flow{i}"]).max() for i in range(cfg.n_source))
    B = meta["B"]
    src = Hh.nhwc_to_nchw(eng.stage("src_fea", DEV).cpu())[:B].numpy()
    d_src = np.abs(src - z["src_fea0"]).max()
    d_tar = np.abs(Hh.nhwc_to_nchw(eng.stage("tar_fea", DEV).cpu()).numpy() - z["tar_fea"]).max()
    d_pg = np.abs(Hh.nhwc_to_nchw(eng.stage("pg", DEV).cpu()).numpy() - z["pg"]).max()
    d_sg = np.abs(Hh.nhwc_to_nchw(eng.stage("sg", DEV).cpu()).numpy() - z["sg"]).max()
    print(f"[{name}] d_rec={d_rec:.2e} d_flow={d_flow:.2e} d_src={d_src:.2e} d_tar={d_tar:.2e} d_pg={d_pg:.2e} d_sg={d_sg:.2e}")
    assert d_flow <= TOL_FLOW and d_src <= TOL_FEA and d_tar <= TOL_FEA and d_sg <= TOL_FEA
    assert d_pg <= TOL_PG
    assert d_rec <= TOL_REC
    eng.close()


def test_pose_golden_composite():
    meta, z, cfg, sd, inputs = Hh.golden_case("g3_pose_256_k1_nb1")
    eng = Hh.make_engine(cfg, sd, 256, 256, 1, DEV)
    rec, _ = Hh.run_engine(eng, inputs, DEV, return_flow=False)
    assert np.abs(rec[:, :, 96:128, 96:128].numpy() - z["rec_crop"]).max() <= TOL_REC
    rows = rec.double().sum(dim=3).numpy()
    assert np.abs(rows - z["rec_rowsum64"]).max() <= 256 * TOL_REC
    # background columns are exactly -mean/255 (TSNet_pose.py:276-280,416-417)
    bg = (-torch.tensor(cfg.mean, dtype=torch.float32) / 255.0).view(1, 3, 1, 1)
    assert torch.equal(rec[:, :, :, :64], bg.expand(1, 3, 256, 64))
    assert torch.equal(rec[:, :, :, 192:], bg.expand(1, 3, 256, 64))
    eng.close()


@pytest.fixture(scope="module")
def cfg0():
    meta, z, cfg, sd, inputs = Hh.golden_case("g4_cfg0_full")
    eng = Hh.make_engine(cfg, sd, 256, 256, 4, DEV)
    rec, flows = Hh.run_engine(eng, inputs, DEV)
    yield dict(meta=meta, z=z, cfg=cfg, sd=sd, inputs=inputs, eng=eng, rec=rec, flows=flows)
    eng.close()


def test_cfg0_full_size_golden(cfg0):
    """BASELINE.json configs[1]: same tensors as cfg0, fp32 on one MI355X, parity vs the reference's CPU forward."""
    z, rec, flows = cfg0["z"], cfg0["rec"], cfg0["flows"]
    d_crop = np.abs(rec[:, :, 96:128, 96:128].numpy() - z["rec_crop"]).max()
    d_rows = np.abs(rec.double().sum(dim=3).numpy() - z["rec_rowsum64"]).max()
    d_flow = max(np.abs(flows[i].numpy() - z[f"flow{i}"]).max() for i in range(3))
    print(f"[cfg0] d_crop={d_crop:.2e} d_rowsum={d_rows:.2e} d_flow={d_flow:.2e}")
    assert d_flow <= TOL_FLOW
    assert d_crop <= TOL_REC
    assert d_rows <= 256 * TOL_REC       # checksum of every output row (covers all pixels)
    s = cfg0["meta"]["summary"]["rec"]
    assert abs(rec.double().mean().item() - s["mean"]) <= 1e-4


def test_cfg0_stage_crops(cfg0):
    z, eng = cfg0["z"], cfg0["eng"]
    src = Hh.nhwc_to_nchw(eng.stage("src_fea", DEV).cpu())[:4, :16, :8, :8].numpy()
    assert np.abs(src - z["src_fea0_crop"]).max() <= TOL_FEA
    for k in ("tar_fea", "pg", "sg"):
        t = Hh.nhwc_to_nchw(eng.stage(k, DEV).cpu())[:, :16, :8, :8].numpy()
        assert np.abs(t - z[k + "_crop"]).max() <= (TOL_PG if k == "pg" else TOL_FEA), k


def test_cfg0_deterministic_and_clip_mode(cfg0):
    """Re-running is bit-identical (fixed reduction orders, no float atomics), and the clip-mode path
    (sources cached once) equals the one-shot forward bit for bit."""
    eng, inputs, rec = cfg0["eng"], cfg0["inputs"], cfg0["rec"]
    rec2, _ = Hh.run_engine(eng, inputs, DEV)
    assert torch.equal(rec, rec2)
    to = lambda t: t.to(DEV)
    eng.set_sources([to(x) for x in inputs[0]], [to(x) for x in inputs[1]], [to(x) for x in inputs[2]])
    r3, _ = eng.forward_target(to(inputs[3]), to(inputs[4]))
    r4, _ = eng.forward_target(to(inputs[3]), to(inputs[4]))
    torch.cuda.synchronize()
    assert torch.equal(rec, r3.cpu()) and torch.equal(rec, r4.cpu())


def test_cfg0_batch_items_independent(cfg0):
    """Every (source-set, driving-frame) pair is independent (per-sample InstanceNorm/softmax): running
    items 1..2 alone reproduces their rows of the B=4 result -- the property multi-GPU sharding rests on."""
    eng, inputs, rec = cfg0["eng"], cfg0["inputs"], cfg0["rec"]
    sl = slice(1, 3)
    sub = ([x[sl] for x in inputs[0]], [x[sl] for x in inputs[1]], [x[sl] for x in inputs[2]], inputs[3][sl], inputs[4][sl])
    r, _ = Hh.run_engine(eng, sub, DEV)
    # the same bits: every tile shape of a kernel runs the same chains per output, statistics partials are per (image, tile) and added
    # in a fixed order, operand scales are per image
    assert torch.equal(r, rec[sl])


def test_single_frame_forward(cfg0):
    """B = 1, the reference demo's own call (demo_face.py:185-192).  Every layer runs the same kernel, tile and accumulation order in every
    batch (round 5: the decoder's second up-convolution moved to the Winograd-along-x kernel and the forward dropped the two-K-group tiles
    of single-frame launches): a frame run alone is the same BITS as its copy inside a batch, and run-to-run identical."""
    import ctypes
    from wacv23_tsnet_amd import _lib
    eng, inputs, rec, z = cfg0["eng"], cfg0["inputs"], cfg0["rec"], cfg0["z"]
    sl = slice(0, 1)
    sub = ([x[sl] for x in inputs[0]], [x[sl] for x in inputs[1]], [x[sl] for x in inputs[2]], inputs[3][sl], inputs[4][sl])
    r1, f1 = Hh.run_engine(eng, sub, DEV)
    cnt = (ctypes.c_int64 * 4)()
    _lib.load().tsnet_debug_counters(cnt, 0)
    assert cnt[3] == 34064, cnt[3]                  # the ResnetBlock layers: 4 x 32 pixels x 64 channels, Winograd form (conv_w1)
    r2, _ = Hh.run_engine(eng, sub, DEV)
    assert torch.equal(r1, r2)
    d_crop = np.abs(r1[:, :, 96:128, 96:128].numpy() - z["rec_crop"][sl]).max()
    d_flow = max(np.abs(f1[i].numpy() - z[f"flow{i}"][sl]).max() for i in range(3))
    d_batch = (r1 - rec[sl]).abs().max().item()
    print(f"[cfg0, B=1] d_crop={d_crop:.2e} d_flow={d_flow:.2e} vs the same frame in the batch of 4: {d_batch:.2e}")
    assert d_crop <= TOL_REC and d_flow <= TOL_FLOW
    assert torch.equal(r1, rec[sl]) and d_batch == 0.0
    Hh.run_engine(eng, inputs, DEV)                 # leave the engine as the other tests expect it (last_B = 4)


def test_model_shell_matches_engine(cfg0):
    """The reference-surface Python object (TSNet / set_test_input / forward) drives the same path."""
    from wacv23_tsnet_amd.model import TSNet
    sd, inputs, rec = cfg0["sd"], cfg0["inputs"], cfg0["rec"]
    m = TSNet(is_train=False, label_nc=2, n_blocks=0, n_downsampling=3, n_source=3, return_flow=True)
    m.load_checkpoint({net: {k[len(net) + 1:]: v for k, v in sd.items() if k.startswith(net + ".")} for net in ("img_enc", "lbl_enc", "fuse_net", "dec")})
    m = m.cuda()
    m.set_test_input(*inputs)
    m.forward()
    torch.cuda.synchronize()
    assert torch.equal(m.rec_tar_img.cpu(), rec)
    assert len(m.warp_grid2d_list) == 3 and tuple(m.warp_grid2d_list[0].shape) == (4, 32, 32, 2)


def test_error_behaviour(cfg0):
    eng, inputs = cfg0["eng"], cfg0["inputs"]
    to = lambda t: t.to(DEV)
    with pytest.raises(ValueError):
        eng.forward([to(x)[:, :2] for x in inputs[0]], [to(x) for x in inputs[1]], [to(x) for x in inputs[2]], to(inputs[3]), to(inputs[4]))
    big = [torch.zeros(5, 3, 256, 256, device=DEV)] * 3
    with pytest.raises(RuntimeError, match="max_batch"):
        eng.forward(big, [torch.zeros(5, 2, 256, 256, device=DEV)] * 3, [torch.zeros(5, 256, 256, device=DEV)] * 3,
                    torch.zeros(5, 2, 256, 256, device=DEV), torch.zeros(5, 256, 256, device=DEV))


# ---- the other BASELINE.json configs as parity cases (fp32 path; compared with the oracle run on this host)
def _oracle_case(cfg, B, H, W, wseed, iseed, bias_std=0.02, mask="box"):
    sd = O.synth_state_dict(cfg, seed=wseed, bias_std=bias_std)
    inp = O.synth_inputs(cfg, B, H, W, seed=iseed, mask_mode=mask)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = O.tsnet_forward(sd, cfg, *inp)
    return sd, inp, ref


def test_cfg2_shape_face_checkpoint_schema_b8():
    """configs[2] shape: face model with n_blocks=4 (demo_face.py:30-33), B=8, synthetic checkpoint incl. biases."""
    cfg = O.TSNetConfig(label_nc=2, n_blocks=4, n_source=3)
    sd, inp, ref = _oracle_case(cfg, 8, 256, 256, 21, 22)
    eng = Hh.make_engine(cfg, sd, 256, 256, 8, DEV)
    rec, flows = Hh.run_engine(eng, inp, DEV)
    d = (rec - ref["rec_tar_img"]).abs().max().item()
    df = max((a - b).abs().max().item() for a, b in zip(flows, ref["flows"]))
    print(f"[cfg2-shape] d_rec={d:.2e} d_flow={df:.2e}")
    assert d <= TOL_REC and df <= TOL_FLOW
    eng.close()


def test_cfg3_shape_pose_l25():
    """configs[3] per-GPU shard: TSNet_pose shape (L=25, n_blocks=4, use_mask), B=4 per GPU."""
    cfg = O.TSNetConfig(label_nc=25, n_blocks=4, n_source=3, pose=True)
    sd, inp, ref = _oracle_case(cfg, 4, 256, 256, 23, 24)
    eng = Hh.make_engine(cfg, sd, 256, 256, 4, DEV)
    rec, _ = Hh.run_engine(eng, inp, DEV, return_flow=False)
    d = (rec - ref["rec_tar_img"]).abs().max().item()
    print(f"[cfg3-shape] d_rec={d:.2e}")
    assert d <= TOL_REC
    eng.close()


def test_cfg3_full_batch_32_on_one_gpu():
    """configs[3] whole: TSNet_pose, bs = 32 (the reference spreads it over 8 GPUs with DataParallel; 288 GB hold it on one).  Every sample is
    independent, so besides the oracle gate the first four samples must equal, bit for bit, a B = 4 run of the same engine."""
    cfg = O.TSNetConfig(label_nc=25, n_blocks=4, n_source=3, pose=True)
    sd, inp, ref = _oracle_case(cfg, 32, 256, 256, 23, 27)
    eng = Hh.make_engine(cfg, sd, 256, 256, 32, DEV)
    rec, _ = Hh.run_engine(eng, inp, DEV, return_flow=False)
    d = (rec - ref["rec_tar_img"]).abs().max().item()
    sub = [[t[:4] for t in x] if isinstance(x, list) else x[:4] for x in inp]
    rec4, _ = Hh.run_engine(eng, sub, DEV, return_flow=False)
    print(f"[cfg3-bs32] d_rec={d:.2e}")
    assert d <= TOL_REC
    assert torch.equal(rec[:4], rec4)
    eng.close()


def test_cfg4_shape_512_k5():
    """configs[4] shape: 512x512, n_source=5 (P=4096 positions: the fused flow kernel never builds P x P)."""
    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=5)
    sd, inp, ref = _oracle_case(cfg, 1, 512, 512, 25, 26, mask="bernoulli")
    eng = Hh.make_engine(cfg, sd, 512, 512, 1, DEV)
    rec, flows = Hh.run_engine(eng, inp, DEV)
    d = (rec - ref["rec_tar_img"]).abs().max().item()
    df = max((a - b).abs().max().item() for a, b in zip(flows, ref["flows"]))
    print(f"[cfg4-shape] d_rec={d:.2e} d_flow={df:.2e}")
    assert d <= TOL_REC and df <= TOL_FLOW
    eng.close()


def test_packed_weight_buffer_goes_through_rccl(cfg0):
    """The multi-GPU path broadcasts the engine's packed weight buffer in place (dist.build_replica).  With one GPU
    the collective is a self-broadcast, but it takes the same route: RCCL must accept the zero-copy alias of the
    engine's allocation, and the engine must compute the same image afterwards."""
    import socket
    import torch.distributed as dist
    eng, inputs = cfg0["eng"], cfg0["inputs"]
    rec0, _ = Hh.run_engine(eng, inputs, DEV, return_flow=False)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        buf = eng.packed_weights(DEV)
        # what a replica receives is what its kernels read: two fp16 planes of the 67 M weights (+ K / N padding), the RGB-head table,
        # biases and the per-layer un-scale factors -- nothing else (<= 300 MB; + 84 MB since the ResnetBlock / FuseNet layers carry
        # their Winograd-transformed filters: 12 instead of 9 taps)
        assert buf.dtype == torch.uint8 and 4 * 67_000_000 < buf.numel() < 360_000_000
        dist.broadcast(buf, src=0)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    rec1, _ = Hh.run_engine(eng, inputs, DEV, return_flow=False)
    assert torch.equal(rec0, rec1)


# ---- bf16-operand mode (tsnet_cfg.operand_mode = 1): BASELINE.json configs[2] and configs[4]
def _bf16_case(cfg, B, H, W, wseed, iseed, mask="box", mode="bf16"):
    sd = O.synth_state_dict(cfg, seed=wseed, bias_std=0.02)
    inp = O.synth_inputs(cfg, B, H, W, seed=iseed, mask_mode=mask)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    eng = Hh.make_engine(cfg, sd, H, W, B, DEV, operands=mode)
    rec, flows = Hh.run_engine(eng, inp, DEV)
    r = Hh.bf16_mode_report(eng, cfg, sd, inp, rec, B, DEV, mode=mode, flows=flows)
    eng.close()
    return r


# Gates of the bf16-operand mode: 1.25 x the worst value measured over the three (weight seed, input seed) pairs of each configuration
# (profiles/round3_bf16_seed_sweep.txt).  max |d| of the smooth stages against the oracle that rounds the same operands, and the
# end-to-end MEAN |d| against that oracle: the maximum is chaotic (softmax(100 corr) turns a 1e-2 feature difference into another flow at
# a few positions), the mean is not.
BF16_GATES = {
    "cfg2": dict(src_fea=0.175, tar_fea=0.0095, sg=0.0553, decoder_on_engine_features=0.0315, end_to_end_vs_bf16_oracle_mean=0.128, decoder_on_engine_features_mean=0.0032),
    # decoder_on_engine_features is a MAXIMUM over 786 k pixels of bf16 rounding flips: one pixel off a smooth tail (8.2e-3, then 6.5, 5.5,
    # 5.4e-3 ...; 99.999 % of the pixels below 4.7e-3) that moves with any change of the decoder's input -- 5.1 / 5.3 / 6.3e-3 on the three
    # draws before flow_kernel_p changed the flows by <= 1e-5, 6.3 / 7.1 / 8.2e-3 after (the forward is bit-deterministic; the MEAN,
    # 1.8-1.9e-4, did not move).  1.33 x the worst.
    "cfg4": dict(src_fea=0.173, tar_fea=0.0088, sg=0.0426, decoder_on_engine_features=0.011, end_to_end_vs_bf16_oracle_mean=0.1415, decoder_on_engine_features_mean=0.00023,
                 # the robust form of the decoder gate: 1.25 x the documented 4.7e-3 tail; the maximum above stays as it was frozen in round 4
                 decoder_on_engine_features_p99999=0.0059),
}
# FROZEN: these limits are not re-fitted when a kernel changes.  Widening one needs >= 5 draws and an entry in DESIGN.md section 3.2.


# bf16 STORAGE mode (operand_mode 2): against the oracle that rounds operands AND the stored activations at the same points; 1.3 x the worst
# of two draws per configuration (gpurun_out/c7/bf16_modes.txt of round 4)
BF16S_GATES = {
    "cfg2": dict(src_fea=0.18, tar_fea=0.013, sg=0.056, decoder_on_engine_features=0.038, end_to_end_vs_bf16_oracle_mean=0.137, decoder_on_engine_features_mean=0.0042),
    "cfg4": dict(src_fea=0.19, tar_fea=0.0145, sg=0.044, decoder_on_engine_features=0.0165, end_to_end_vs_bf16_oracle_mean=0.149, decoder_on_engine_features_mean=0.00072),
}


def _gate(tag, r, gates=None):
    print(f"[{tag} bf16] " + " ".join(f"{k}={v:.3e}" for k, v in r.items()))
    for k, lim in (gates or BF16_GATES)[tag].items():
        assert r[k] <= lim, (tag, k, r[k], lim)
    # the transformation branch fed the engine's OWN features is fp32-class arithmetic in every operand mode: the fp32 mode's tolerances
    # (flows 1e-4, warped feature 4e-3) hold here too -- the chaos of the end-to-end figures enters through the features, not through it
    assert r["flow_on_engine_features"] <= 1e-4 and r["pg_on_engine_features"] <= 4e-3, (tag, r["flow_on_engine_features"], r["pg_on_engine_features"])


@pytest.mark.parametrize("wseed,iseed", [(21, 22), (31, 32), (41, 42)])
def test_cfg2_bf16_face_checkpoint_shape_b8(wseed, iseed):
    """configs[2]: face-checkpoint shape (n_blocks=4), bs=8, bf16 operands on one MI355X, three (weights, inputs) draws.  Gates: BF16_GATES;
    the end-to-end maxima are reported, not gated (helpers.bf16_mode_report)."""
    _gate("cfg2", _bf16_case(O.TSNetConfig(label_nc=2, n_blocks=4, n_source=3), 8, 256, 256, wseed, iseed))


@pytest.mark.parametrize("wseed,iseed", [(25, 26), (35, 36), (45, 46)])
def test_cfg4_bf16_512_k5(wseed, iseed):
    """configs[4] per-GPU shard: 512 x 512, n_source=5, bf16 operands (P = 4096 positions), three (weights, inputs) draws."""
    _gate("cfg4", _bf16_case(O.TSNetConfig(label_nc=2, n_blocks=0, n_source=5), 1, 512, 512, wseed, iseed, mask="bernoulli"))


def test_bf16_storage_mode_cfg2_and_cfg4():
    """tsnet_cfg.operand_mode = 2 (bf16 operands + bf16 storage of the large convolution-to-convolution activations: VERDICT r3 #5) at the
    shapes of configs[2] and configs[4]: the same stage-wise gates as mode 1, against the oracle that rounds at the same points."""
    _gate("cfg2", _bf16_case(O.TSNetConfig(label_nc=2, n_blocks=4, n_source=3), 8, 256, 256, 21, 22, mode="bf16s"), BF16S_GATES)
    _gate("cfg4", _bf16_case(O.TSNetConfig(label_nc=2, n_blocks=0, n_source=5), 1, 512, 512, 25, 26, mask="bernoulli", mode="bf16s"), BF16S_GATES)
