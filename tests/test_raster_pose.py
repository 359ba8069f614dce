"""SURVEY.md section 8-f rank 3, pose clips: OpenPose key points -> class-index skeleton label, crop, bounding-box mask, 128 x 256 nearest
resize + padding to 256 x 256.
  * the oracle restatement (oracle/raster_pose_oracle.py) reproduces the REAL reference's outputs on the 60 frames of the demo clips
    (tests/golden/g9_raster_pose.npz, captured by oracle/capture_raster_pose_goldens.py from the imported reference);
  * the host logic of the product (file reader, crop rectangle, PIL-order index tables) against the same golden and against PIL itself;
  * the device kernels (csrc/raster.hpp; CPU emulation build here, the HIP library in the gpu-marked test) against the golden.  Every piece
    of the skeleton is a straight line through two key points, which the reference fits with scipy's Levenberg-Marquardt; where a
    key-point coordinate is integer-valued the stroke's end sample has an exactly integer ordinate and truncates to n or n - 1 by the last
    bits of the optimiser's result.  The host fit (csrc/lmfit.hpp, tsnet_fit_pose_curves) reproduces those bits: the labels must EQUAL the
    reference's on all 60 frames, on the re-drawn driving frames and on the limb-rescaled ones."""
import json
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

import helpers as Hh
from oracle import raster_pose_oracle as PO
from wacv23_tsnet_amd import raster

sys.path.insert(0, os.path.join(Hh.ROOT, "tools")) if hasattr(Hh, "ROOT") else sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import pose_preprocess as pre        # loader-side preprocessing of a driving clip: a tool outside the product package


def _golden():
    z = np.load(os.path.join(Hh.GOLD, "g9_raster_pose.npz"))
    return json.loads(str(z["meta"])), z


def _bits(z, key, w):
    return np.unpackbits(z[key], axis=-1)[:, :, :w]


def test_oracle_reproduces_reference_labels():
    meta, z = _golden()
    for clip, m in meta["clips"].items():
        size, crop = tuple(m["size"]), m["crop"]
        cw = crop[2] - crop[0]
        pts, cls_crop, cls_256 = z[f"{clip}_pts"], z[f"{clip}_cls_crop"], z[f"{clip}_cls_256"]
        bbox_crop, bbox_256 = _bits(z, f"{clip}_bbox_crop", cw), _bits(z, f"{clip}_bbox_256", 256)
        assert list(PO.crop_coords(pts[0][:25], size)[0]) == crop
        assert np.array_equal(PO.select_person(str(z[f"{clip}_json0"])), pts[0])
        for f in range(0, pts.shape[0], 4):              # every fourth frame: scipy's fits are the slow part
            full = PO.skeleton_classes(pts[f], size)
            got = full[crop[1]:crop[3], crop[0]:crop[2]]
            assert np.array_equal(got, cls_crop[f]), (clip, f)
            box = PO.label_bbox(got)
            assert np.array_equal(box != 0, bbox_crop[f] != 0), (clip, f)
            assert np.array_equal(PO.resize_square(got), cls_256[f]), (clip, f)
            assert np.array_equal(PO.resize_square(box) != 0, bbox_256[f] != 0), (clip, f)
            re = PO.skeleton_classes(z[f"{clip}_pts_redraw"][f], (cw, crop[3] - crop[1]))
            assert np.array_equal(re, z[f"{clip}_cls_redraw"][f]), (clip, f)


def test_host_logic_matches_golden_and_pil():
    meta, z = _golden()
    for clip, m in meta["clips"].items():
        pts = z[f"{clip}_pts"]
        assert np.array_equal(raster.read_openpose(str(z[f"{clip}_json0"])), pts[0])
        assert list(raster.pose_crop_coords(pts[0], tuple(m["size"]))) == m["crop"]
    for n_in, n_out in [(780, 256), (390, 128), (742, 256), (370, 128), (1080, 256), (100, 256), (257, 256), (1000, 333), (333, 1000), (1, 7)]:
        ramp = np.tile(np.arange(n_in, dtype=np.int32)[:, None], (1, 2))
        want = np.asarray(Image.fromarray(ramp, mode="I").resize((2, n_out), resample=Image.NEAREST))[:, 0]
        assert np.array_equal(raster.nearest_table(n_in, n_out), want), (n_in, n_out)


def test_pose_preprocessing_host_logic_matches_reference():
    """The driving clip's preprocessing in front of the rasteriser, per person and frame (host arithmetic on 137 points): the five-frame temporal
    smoothing (dataset/smooth_pose_keypoint.py:85-114; the golden equals the file the reference ships for clip 00164), the shift into the crop
    and the limb re-scaling of opposite-sex pairs (keypoint2img_posenorm.py:70-226, both directions) -- EQUAL to the reference's outputs -- and
    the reader of the smoothing script's json_tricks files."""
    meta, z = _golden()
    for clip, m in meta["clips"].items():
        crop = m["crop"]
        assert np.array_equal(pre.smooth_clip(z[f"{clip}_pts"]), z[f"{clip}_smooth"])
        assert np.array_equal(pre.shift_into_crop(z[f"{clip}_fm_pts_in"][0], crop), z[f"{clip}_pts_redraw"][0])
        for mode in ("fm", "mf"):
            pin, pout = z[f"{clip}_{mode}_pts_in"], z[f"{clip}_{mode}_pts_out"]
            for p, q in zip(pin, pout):
                got = pre.pose_limb_rescale(pre.shift_into_crop(p, crop), mode, crop[3] - crop[1])
                assert np.array_equal(got, q), (clip, mode)
            assert np.abs(pout - np.stack([pre.shift_into_crop(p, crop) for p in pin])).max() > 5      # the re-scaling really moves limbs
    with pytest.raises(ValueError):
        pre.pose_limb_rescale(z["00110_pts"][0], "ff", 100)


def test_read_smooth_openpose_file(tmp_path):
    meta, z = _golden()
    sm = z["00164_smooth"]
    enc = lambda a: {"__ndarray__": a.tolist(), "dtype": "float64", "shape": list(a.shape), "Corder": True}      # json_tricks' array encoding
    doc = {"pose_keypoints_2d": enc(sm[:, :25]), "face_keypoints_2d": enc(sm[:, 25:95]), "hand_left_keypoints_2d": enc(sm[:, 95:116]),
           "hand_right_keypoints_2d": enc(sm[:, 116:]), "name": ["frame%06d" % i for i in range(sm.shape[0])]}
    path = tmp_path / "00164.json"
    path.write_text(json.dumps(doc))
    assert np.array_equal(pre.read_smooth_openpose(str(path)), sm)


def _device_check(lib, dev):
    meta, z = _golden()
    r = raster.PoseRasteriser(dev, lib=lib)
    report = {}
    for clip, m in meta["clips"].items():
        size, crop = tuple(m["size"]), tuple(m["crop"])
        cw, ch = crop[2] - crop[0], crop[3] - crop[1]
        pts, want = z[f"{clip}_pts"], z[f"{clip}_cls_crop"]
        got = r.rasterise(list(pts), size, crop).cpu().numpy()
        diff = got != want
        report[clip] = dict(frames=int(pts.shape[0]), label_pixels=int((want != 0).sum()), differing=int(diff.sum()))
        assert np.array_equal(got, want), report
        # the 'pts' path of the driving frames: drawn directly at crop size (border clamping against the crop)
        re = r.rasterise(list(z[f"{clip}_pts_redraw"]), (cw, ch)).cpu().numpy()
        report[clip]["redraw_differing"] = int((re != z[f"{clip}_cls_redraw"]).sum())
        assert np.array_equal(re, z[f"{clip}_cls_redraw"]), report
        # opposite-sex pairs: the label drawn from the re-scaled points (host) at crop size
        for mode in ("fm", "mf"):
            pin, want_m = z[f"{clip}_{mode}_pts_in"], z[f"{clip}_{mode}_cls"]
            moved = [pre.pose_limb_rescale(pre.shift_into_crop(p, crop), mode, ch) for p in pin]
            got_m = r.rasterise(moved, (cw, ch)).cpu().numpy()
            report[clip][f"{mode}_differing"] = int((got_m != want_m).sum())
            assert np.array_equal(got_m, want_m), report
        # integer work downstream of the class map: equal, given the reference's own class map
        ref_cls = torch.from_numpy(want)
        box = r.bbox(ref_cls)
        assert np.array_equal(box.cpu().numpy() != 0, _bits(z, f"{clip}_bbox_crop", cw) != 0)
        assert set(np.unique(box.cpu().numpy())) <= {0, 255}
        sq = r.to_square(ref_cls).cpu().numpy()
        assert sq.dtype == np.float32 and np.array_equal(sq, z[f"{clip}_cls_256"].astype(np.float32))
        sqb = r.to_square(box, binarise=True).cpu().numpy()
        assert np.array_equal(sqb, _bits(z, f"{clip}_bbox_256", 256).astype(np.float32))
        # and the whole chain from the points, as the clip harness runs it
        cls256, box256, win = r.clip_labels(list(pts), size)
        assert tuple(win) == crop
        report[clip]["chain_cls_256_differing"] = int((cls256.cpu().numpy() != z[f"{clip}_cls_256"]).sum())
        report[clip]["chain_bbox_256_differing"] = int((box256.cpu().numpy() != _bits(z, f"{clip}_bbox_256", 256)).sum())
        assert report[clip]["chain_cls_256_differing"] == 0 and report[clip]["chain_bbox_256_differing"] == 0
    print("[raster-pose] " + json.dumps(report))
    return report


def test_pose_kernels_emulated(emu_lib):
    _device_check(emu_lib, "cpu")


def test_pose_kernel_flags_and_edge_cases(emu_lib):
    """basic_point_only / remove_face_labels, points at the frame border (brush clamping), coincident and missing points, a frame with no
    usable point at all -- against the oracle restatement."""
    r = raster.PoseRasteriser("cpu", lib=emu_lib)
    rng = np.random.default_rng(5)
    size = (160, 200)
    frames = []
    for f in range(6):
        p = rng.uniform(5, 155, (137, 2))                # generic doubles: no sample ordinate within the fit's error of an integer
        p[:25, 1] = rng.uniform(2, 198, 25)
        frames.append(p)
    frames[1][[3, 40, 100]] = 0                      # missing points
    frames[2][4] = frames[2][3]                      # coincident points: an empty stroke
    frames[3][:25, 0] = rng.uniform(0.2, 3, 25)      # hugging the left border
    frames[4][:] = 0                                 # nothing usable
    for flags in ((False, False), (True, False), (False, True)):
        got = r.rasterise(frames, size, None, basic_point_only=flags[0], remove_face_labels=flags[1]).numpy()
        for f, p in enumerate(frames):
            want = PO.skeleton_classes(p, size, basic_point_only=flags[0], remove_face_labels=flags[1])
            assert np.array_equal(got[f], want), (flags, f, int((got[f] != want).sum()))
    assert got[4].max() == 0
    box = r.bbox(torch.from_numpy(got)).numpy()
    assert box[4].max() == 0                          # no label pixel: empty mask (the reference raises there)
    for f in (0, 1, 2, 3, 5):
        assert np.array_equal(box[f], PO.label_bbox(got[f]))
    win = (10, 20, 90, 180)                          # a window: the crop of the full drawing
    sub = r.rasterise(frames, size, win).numpy()
    full = r.rasterise(frames, size, None).numpy()
    assert np.array_equal(sub, full[:, 20:180, 10:90])
    with pytest.raises(RuntimeError):
        r.rasterise(frames, size, (0, 0, 161, 10))


@pytest.mark.gpu
def test_pose_kernels_gpu():
    rep = _device_check(None, "cuda")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/raster_pose_report.json", "w") as f:
        json.dump(rep, f)
