"""Pins the oracle: oracle/tsnet_oracle.py must reproduce the vectors captured from the REAL reference
(oracle/capture_goldens.py, run in the authoring container; tests/golden/MANIFEST.json records that
oracle-vs-reference was 0.0 at capture).  Bit-exact at the capture thread count, a few 1e-6 otherwise."""
import json

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import tsnet_oracle as O

TOL = 2e-5   # 1-thread vs 8-thread oneDNN summation order (3.9e-6 on the image)

SMALL = ["g3_face_64_k2_nb0", "g3_face_64_k2_nb1_bias", "g2_face_64_softmask", "g2_face_64_ones", "g2_face_64_zeros",
         "g2_face_32_k3", "g3_face_128x64_k2"]


@pytest.fixture(autouse=True)
def _threads():
    old = torch.get_num_threads()
    torch.set_num_threads(8)
    yield
    torch.set_num_threads(old)


@pytest.mark.parametrize("name", SMALL)
def test_small_cases(name):
    meta, z, cfg, sd, inputs = Hh.golden_case(name)
    assert meta["oracle_vs_ref"] == {"rec": 0.0, "flow": 0.0}
    out = O.tsnet_forward(sd, cfg, *inputs, want_stages=True)
    assert np.abs(out["rec_tar_img"].numpy() - z["rec"]).max() <= TOL
    for i in range(cfg.n_source):
        assert np.abs(out["flows"][i].numpy() - z[f"flow{i}"]).max() <= TOL
    for k in ("tar_fea", "pg", "sg"):
        assert np.abs(out["stages"][k].numpy() - z[k]).max() <= 2e-4, k
    assert np.abs(out["stages"]["src_fea"][0].numpy() - z["src_fea0"]).max() <= 2e-4


def test_pose_composite_case():
    meta, z, cfg, sd, inputs = Hh.golden_case("g3_pose_256_k1_nb1")
    out = O.tsnet_forward(sd, cfg, *inputs)
    rec = out["rec_tar_img"]
    assert np.abs(rec[:, :, 96:128, 96:128].numpy() - z["rec_crop"]).max() <= TOL
    assert np.abs(rec.double().sum(dim=3).numpy() - z["rec_rowsum64"]).max() <= 256 * TOL


def test_cfg0_full_size():
    """BASELINE.json configs[0]: TSNet(label_nc=2, n_downsampling=3, n_source=3) on 4x3x256x256, CPU."""
    meta, z, cfg, sd, inputs = Hh.golden_case("g4_cfg0_full")
    out = O.tsnet_forward(sd, cfg, *inputs)
    rec = out["rec_tar_img"]
    assert np.abs(rec[:, :, 96:128, 96:128].numpy() - z["rec_crop"]).max() <= TOL
    assert np.abs(rec.double().sum(dim=3).numpy() - z["rec_rowsum64"]).max() <= 256 * TOL
    for i in range(3):
        assert np.abs(out["flows"][i].numpy() - z[f"flow{i}"]).max() <= TOL
    assert abs(rec.double().mean().item() - meta["summary"]["rec"]["mean"]) < 1e-6


def test_manifest_lists_every_golden():
    import os
    names = {m["name"] for m in json.load(open(os.path.join(Hh.GOLD, "MANIFEST.json")))}
    assert set(SMALL) | {"g3_pose_256_k1_nb1", "g4_cfg0_full"} <= names


@pytest.mark.parametrize("name", ["g10_face_test114_to_val024_b1", "g10_pose_00110_to_00164_b1"])
def test_demo_range_inputs(name):
    """Realistic inputs (oracle/capture_demo_input_goldens.py): frames of the reference's demo clips as demo_face.py:150-192 / demo_pose.py
    feed them -- BGR - IMG_MEAN in [-112, 154], edge-map / skeleton labels, box masks, n_blocks = 4, K = 3, B = 1 -- rebuilt from the bytes
    stored with the golden; the oracle reproduces the reference's fp32 crops / lattice / row checksums and flows (0.0 at capture)."""
    meta, z, cfg, sd, inputs = Hh.golden_case(name)
    assert meta["inputs"] == "stored" and meta["oracle_vs_ref"]["rec"] == 0.0 and meta["oracle_vs_ref"]["rec64"] == 0.0
    lo, hi = min(float(x.min()) for x in inputs[0]), max(float(x.max()) for x in inputs[0])
    assert lo < -100 and hi > 100 and [lo, hi] == meta["image_range"]
    assert all(torch.equal(l.sum(dim=1), torch.ones_like(l[:, 0])) for l in inputs[1] + [inputs[3]])       # one-hot labels
    out = O.tsnet_forward(sd, cfg, *inputs)
    rec = out["rec_tar_img"]
    H, W = meta["H"], meta["W"]
    for tag, (ys, xs) in {"c": (slice(96, 128), slice(96, 128)), "tl": (slice(0, 16), slice(0, 16)), "br": (slice(H - 16, H), slice(W - 16, W)),
                          "sub4": (slice(None, None, 4), slice(None, None, 4))}.items():
        assert np.abs(rec[:, :, ys, xs].numpy() - z[f"rec32_{tag}"]).max() <= TOL, tag
    assert np.abs(rec.double().sum(dim=3).numpy() - z["rec32_rowsum"]).max() <= W * TOL
    if meta["has_flow"]:
        for i in range(cfg.n_source):
            assert np.abs(out["flows"][i].numpy() - z[f"flow32_{i}"]).max() <= TOL
