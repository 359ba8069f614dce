"""key points -> edge map, bounding-box mask, one-hot label.
  * the oracle restatement (oracle/raster_oracle.py) reproduces the REAL reference's maps on every frame of the demo clips bit for bit
    (tests/golden/g7_raster_face.npz, captured by oracle/capture_raster_goldens.py from the imported reference);
  * the device kernels (csrc/raster.hpp; CPU emulation build here, the HIP library in the gpu-marked test): bounding box and one-hot
    are integer work and must be EQUAL; the edge map uses the closed-form parabola where the reference runs scipy's Levenberg-Marquardt
    fit, so a sample within the optimiser's error of an integer may truncate the other way: the Hamming distance to the reference's
    map is measured on all 78 frames and bounded by what was measured (printed)."""
import json
import os

import numpy as np
import pytest
import torch

import helpers as Hh
from oracle import raster_oracle as RO
from wacv23_tsnet_amd import raster


def _golden():
    z = np.load(os.path.join(Hh.GOLD, "g7_raster_face.npz"))
    return json.loads(str(z["meta"])), z


def _maps(z, clip, w):
    return (np.unpackbits(z[f"{clip}_edges"], axis=-1)[:, :, :w] * 255).astype(np.uint8), (np.unpackbits(z[f"{clip}_bbox"], axis=-1)[:, :, :w] * 255).astype(np.uint8)


def test_oracle_reproduces_reference_maps():
    meta, z = _golden()
    assert len(RO.sub_edges()) == 34
    for clip, m in meta["clips"].items():
        w, h = m["size"]
        edges, boxes = _maps(z, clip, w)
        kps = z[f"{clip}_keypoints"]
        for f in range(0, kps.shape[0], 3):          # every third frame: scipy's fit is the slow part
            assert np.array_equal(RO.face_edge_map(kps[f], (w, h), m["bw"]), edges[f]), (clip, f)
            assert np.array_equal(RO.bbox_mask(kps[f], (w, h)), boxes[f]), (clip, f)
    assert np.array_equal(RO.vl2ch(z["vl2ch_face_in"], 2), z["vl2ch_face_out"])
    assert np.array_equal(RO.vl2ch(z["vl2ch_pose_in"], 25), z["vl2ch_pose_out"])


def test_crop_arithmetic_matches_golden():
    meta, z = _golden()
    for clip, m in meta["clips"].items():
        kp0 = z[f"{clip}_keypoints"][0].copy()
        kp0[:, 0] += m["crop"][2]
        kp0[:, 1] += m["crop"][0]                    # back to frame coordinates
        assert list(raster.crop_coords(kp0)) == m["crop"] == list(RO.crop_coords(kp0))


def _device_check(lib, dev):
    meta, z = _golden()
    r = raster.FaceRasteriser(dev, lib=lib)
    report = {}
    for clip, m in meta["clips"].items():
        w, h = m["size"]
        want_e, want_b = _maps(z, clip, w)
        kp = z[f"{clip}_keypoints"].copy()
        kp[:, :, 0] += m["crop"][2]
        kp[:, :, 1] += m["crop"][0]
        edges, bbox, crop, bw = r.rasterise(list(kp))
        if torch.device(dev).type == "cuda":
            torch.cuda.synchronize()
        assert list(crop) == m["crop"] and bw == m["bw"]
        got_e, got_b = edges.cpu().numpy(), bbox.cpu().numpy()
        assert np.array_equal(got_b, want_b), clip                                   # integer work: bit-exact
        ham = (got_e != want_e).reshape(got_e.shape[0], -1).sum(axis=1)
        # one-pixel tolerance: every pixel one map sets lies within one pixel (8-neighbourhood) of a pixel the other sets
        from scipy.ndimage import binary_dilation
        st = np.ones((1, 3, 3), bool)
        stray = int(((got_e > 0) & ~binary_dilation(want_e > 0, structure=st)).sum() + ((want_e > 0) & ~binary_dilation(got_e > 0, structure=st)).sum())
        report[clip] = dict(frames=int(got_e.shape[0]), edge_pixels=int((want_e > 0).sum()), hamming_total=int(ham.sum()),
                            hamming_max_per_frame=int(ham.max()), frames_exact=int((ham == 0).sum()), pixels_beyond_one_pixel=stray)
        # Measured: 2-3 % of the edge pixels differ, all at the END POINTS of the 34 curve pieces, where the
        # true ordinate is an integer key point and the reference's own fitted value (217.99999.. or 218.0000..) truncates either way.
        assert stray == 0, report[clip]
        assert ham.max() <= 160 and ham.sum() <= 0.05 * (want_e > 0).sum(), report[clip]
    print("[raster] " + json.dumps(report))
    for name, nc in (("face", 2), ("pose", 25)):
        out = r.vl2ch(torch.from_numpy(z[f"vl2ch_{name}_in"].astype(np.float32)), nc)
        assert np.array_equal(out.cpu().numpy().astype(np.uint8), z[f"vl2ch_{name}_out"])
    return report


def test_device_kernels_emulated(emu_lib):
    _device_check(emu_lib, "cpu")


@pytest.mark.gpu
def test_device_kernels_gpu():
    from wacv23_tsnet_amd import _lib
    _device_check(_lib.load(), "cuda")
