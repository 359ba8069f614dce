"""demo post-processing (frame re-normalisation + sample_img) on the device.
  * tests/golden/g8_demo_post.npz holds the bytes the reference's OWN statements (demo/demo_face.py:27,94-103,180-182,195-199, lifted
    with ast and executed by oracle/capture_demo_goldens.py) produce on PRNG frames; oracle/demo_oracle.py must reproduce them exactly;
  * the device bytes must EQUAL the fp64-statistics evaluation of those lines (demo_oracle stats64=True): every per-pixel operation is
    the reference's fp32 operation in the reference's order, and the four statistics are well-defined numbers there;
  * against the reference's fp32-statistics bytes they may differ by 1 LSB where torch's fp32 mean / std (summation order of the
    host's vectorised reduction) differ from the exact value in the last bit and a pixel sits on an integer boundary: counted, bounded."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import demo_oracle as DO
from wacv23_tsnet_amd import demo, prng


def _frames(B, H, W, seed):
    """generator-like frames: tanh-range values with channel-dependent offset / contrast"""
    g = prng.normal(seed, "rec", (B, 3, H, W))
    x = torch.tanh(0.6 * g + torch.tensor([0.1, -0.2, 0.3]).view(1, 3, 1, 1)) * torch.tensor([0.5, 0.35, 0.6]).view(1, 3, 1, 1)
    ref = prng.uniform01(seed, "ref", (1, 3, H, W)) * 255.0 - torch.from_numpy(DO.IMG_MEAN).view(1, 3, 1, 1)
    return x.float().contiguous(), ref.float().contiguous()


def _golden():
    import json
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_demo_post.npz"))
    return json.loads(str(z["meta"]))["cases"], z


def test_oracle_reproduces_reference_bytes():
    cases, z = _golden()
    for tag, c in cases.items():
        rec, ref = _frames(c["B"], c["H"], c["W"], c["seed"])
        ref_mean, ref_std = DO.ref_statistics(ref)
        got = np.stack([DO.postprocess_frame(rec[b:b + 1].clone(), ref_mean, ref_std) for b in range(c["B"])])
        assert np.array_equal(got, z[f"{tag}_rgb"]), tag
        assert np.array_equal(ref_mean.view(3).numpy(), z[f"{tag}_ref_mean"]) and np.array_equal(ref_std.view(3).numpy(), z[f"{tag}_ref_std"])


def _check(lib, dev, B, H, W, seed, tag=None):
    rec, ref = _frames(B, H, W, seed)
    rm64, rs64 = DO.ref_statistics(ref, stats64=True)
    want64 = np.stack([DO.postprocess_frame(rec[b:b + 1].clone(), rm64, rs64, stats64=True) for b in range(B)])
    post = demo.DemoPostprocessor(ref.to(dev), lib=lib)
    assert torch.equal(post.ref_mean.cpu(), rm64.view(3)) and torch.equal(post.ref_std.cpu(), rs64.view(3))
    got = post(rec.to(dev))
    if dev != "cpu":
        torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == want64.shape == (B, H, W, 3) and got.dtype == np.uint8
    assert np.array_equal(got, want64)                    # EQUAL to the fp64-statistics evaluation of the reference's lines
    assert got.min() == 0 and got.max() == 255            # the clip is exercised on both sides
    if tag is not None:                                   # the reference's own bytes (fp32 statistics): at most 1 LSB, counted
        cases, z = _golden()
        assert cases[tag] == dict(B=B, H=H, W=W, seed=seed)
        d = np.abs(got.astype(np.int16) - z[f"{tag}_rgb"].astype(np.int16))
        print(f"[demo_post {tag}] bytes differing from the reference's: {int((d != 0).sum())} of {d.size} (max {int(d.max())} LSB)")
        assert d.max() <= 1 and (d != 0).mean() <= 5e-4
    return got, want64


def test_demo_postprocess_emulated(emu_lib):
    _check(emu_lib, "cpu", 2, 64, 48, seed=31, tag="a")
    _check(emu_lib, "cpu", 1, 33, 17, seed=32, tag="b")    # ragged size, single frame (the demo's case)


def test_strip_and_gif_writers(tmp_path, emu_lib):
    got, _ = _check(emu_lib, "cpu", 2, 32, 32, seed=33)
    src = demo.input_to_rgb(torch.zeros(3, 32, 32))
    strip = demo.save_strip(src, src, got[0], str(tmp_path / "strip.png"))
    assert strip.shape == (32, 96, 3) and np.array_equal(strip[:, 64:], got[0])
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "strip.png").convert("RGB")), strip)      # PNG is lossless
    demo.save_gif([strip, np.ascontiguousarray(strip[::-1])], str(tmp_path / "clip.gif"))    # PIL merges identical frames
    assert Image.open(tmp_path / "clip.gif").n_frames == 2


@pytest.mark.gpu
def test_demo_postprocess_gpu():
    from wacv23_tsnet_amd import _lib
    _check(_lib.load(), "cuda", 2, 256, 256, seed=41, tag="c")
    _check(_lib.load(), "cuda", 1, 256, 256, seed=42, tag="d")
