"""key points -> edge map, bounding-box mask, one-hot label.
  * the oracle restatement (oracle/raster_oracle.py) reproduces the REAL reference's maps on every frame of the demo clips bit for bit
    (tests/golden/g7_raster_face.npz, captured by oracle/capture_raster_goldens.py from the imported reference);
  * the host fit (csrc/lmfit.hpp through tsnet_fit_face_curves: MINPACK's lmdif restated) must return scipy.optimize.curve_fit's
    coefficients BIT FOR BIT on every piece of every frame -- the fitted ordinate at an integer key point is 217.99.. or 218.00.. by the
    optimiser's last bits, and the reference truncates it;
  * the device kernels (csrc/raster.hpp; CPU emulation build here, the HIP library in the gpu-marked test): edge map, bounding box and
    one-hot are integer / byte work and must EQUAL the reference's on all 78 frames of its demo clips."""
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


def test_host_fit_equals_curve_fit_bit_for_bit(emu_lib):
    """tsnet_fit_face_curves (host code of the library: the emulation build compiles the same csrc/lmfit.hpp) against scipy's curve_fit called
    exactly as utils/keypoint2img.py:319-337 calls it, on all 34 x 78 pieces of the demo clips: kind, coefficients and sample range."""
    import warnings
    from scipy.optimize import curve_fit
    meta, z = _golden()
    n_fits = 0
    for clip in meta["clips"]:
        kps = np.ascontiguousarray(z[f"{clip}_keypoints"])
        F = kps.shape[0]
        rec = np.full((F, 34, 8), np.nan)
        assert emu_lib.tsnet_fit_face_curves(kps.ctypes.data, F, rec.ctypes.data) == 0
        for f in range(F):
            for e, se in enumerate(RO.sub_edges()):
                x, y = kps[f][se, 0], kps[f][se, 1]
                swap = abs(x[:-1] - x[1:]).max() < abs(y[:-1] - y[1:]).max()
                if swap:
                    x, y = y, x
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    popt, _ = curve_fit(RO._linear if len(x) < 3 else RO._quadratic, x, y)
                n_fits += 1
                if len(x) == 3 and abs(popt[0]) > 1:
                    assert rec[f, e, 0] == 0
                    continue
                assert rec[f, e, 0] == 1 + 2 * swap + 4 * (len(x) == 3)
                want = [0.0, popt[0], popt[1]] if len(x) < 3 else list(popt)
                assert [float(v) for v in rec[f, e, 1:4]] == [float(v) for v in want], (clip, f, e, rec[f, e], popt)       # == on doubles: bit for bit
                assert rec[f, e, 4] == min(x[0], x[-1]) and rec[f, e, 5] == max(x[0], x[-1])
    assert n_fits == 34 * 78


def test_degenerate_pieces_follow_the_reference(emu_lib):
    """Integer landmarks make pieces whose points share an abscissa along the fitting axis, repeat, or collapse to one point (a
    closed-form parabola gives NaN there).  The host fit runs the reference's optimiser on them too: same `popt` bits as scipy's curve_fit
    (or the same decision to drop the piece), and the maps the device draws from them equal the oracle's."""
    import warnings
    from scipy.optimize import curve_fit
    g = np.random.default_rng(7)
    kps = g.integers(30, 260, size=(6, 68, 2)).astype(np.float64)
    subs = RO.sub_edges()
    for f in range(6):                                   # force the degenerate shapes into every frame
        for e in g.choice(len(subs), size=8, replace=False):
            se = subs[e]
            mode = g.integers(0, 4)
            if mode == 0:
                kps[f, se[1]] = kps[f, se[0]]                              # repeated point
            elif mode == 1:
                kps[f, se[1], 0] = kps[f, se[0], 0]; kps[f, se[1], 1] = kps[f, se[0], 1] + 40     # same x, far apart in y
            elif mode == 2:
                kps[f, se, :] = kps[f, se[0]]                              # the whole piece is one point
            else:
                kps[f, se[-1], 1] = kps[f, se[0], 1]; kps[f, se[-1], 0] = kps[f, se[0], 0] + 1
    kps = np.ascontiguousarray(kps)
    rec = np.full((6, 34, 8), np.nan)
    assert emu_lib.tsnet_fit_face_curves(kps.ctypes.data, 6, rec.ctypes.data) == 0
    checked = 0
    for f in range(6):
        for e, se in enumerate(subs):
            x, y = kps[f][se, 0], kps[f][se, 1]
            swap = abs(x[:-1] - x[1:]).max() < abs(y[:-1] - y[1:]).max()
            if swap:
                x, y = y, x
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                popt, _ = curve_fit(RO._linear if len(x) < 3 else RO._quadratic, x, y)
            if len(x) == 3 and abs(popt[0]) > 1:
                assert rec[f, e, 0] == 0
                continue
            want = [0.0, popt[0], popt[1]] if len(x) < 3 else list(popt)
            got = [float(v) for v in rec[f, e, 1:4]]
            assert all((a == b) or (np.isnan(a) and np.isnan(b)) for a, b in zip(got, [float(v) for v in want])), (f, e, x, y, got, want)
            checked += 1
    assert checked > 150
    r = raster.FaceRasteriser("cpu", lib=emu_lib)
    crop = (0, 292, 0, 292)
    edges, bbox, _, bw = r.rasterise(list(kps), crop)
    for f in range(6):
        want = RO.face_edge_map(kps[f], (292, 292), bw)
        assert np.array_equal(edges[f].numpy(), want), f


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
        report[clip] = dict(frames=int(got_e.shape[0]), edge_pixels=int((want_e > 0).sum()), hamming_total=int(ham.sum()), frames_exact=int((ham == 0).sum()))
        assert np.array_equal(got_e, want_e), report[clip]                           # the reference's edge map, pixel for pixel, on every frame
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
