"""Host side of the device rasterisation: key-point files of a clip -> the label tensors
`set_test_input` takes.  The reference does this per frame on the CPU in its data loader
(dataset/dataset_video_face.py:283-330, FaceDatasetTest.__getitem__); here the key points of a whole clip go to the device once
and three kernels (csrc/raster.hpp) produce every frame's edge map, bounding-box mask and one-hot label.

Only the crop arithmetic (a handful of integer operations per clip, dataset_video_face.py:507-518) stays on the host.

Pose clips (dataset/dataset_video_pose.py:304-461, PoseDatasetTestVideo.__getitem__): `PoseRasteriser` takes the OpenPose points of a clip to the
device once; the colour-coded skeleton (as class indices), its crop, the bounding-box mask, the nearest-neighbour resize to 128 x 256, the
padding to 256 x 256 and the one-hot label are kernels.  The host keeps what is per-person arithmetic on 137 points: the confidence
thresholds, the choice of the person, the crop rectangle, the temporal smoothing of a driving clip, the limb re-scaling of opposite-sex pairs and the
two index tables of the resize.  (The loader-side preprocessing of a driving clip -- five-frame smoothing, shift into the crop, limb re-scaling of
opposite-sex pairs: SURVEY.md section 2 marks it out of scope -- lives in tools/pose_preprocess.py, outside the product package.)"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


def read_keypoints(path: str) -> np.ndarray:
    """68 x 2 landmark file of the demo clips: one 'x,y' pair per line (FaceDatasetTest.read_data, :457: np.loadtxt(delimiter=','))."""
    return np.loadtxt(path, delimiter=",")


def crop_coords(keypoints: np.ndarray) -> Tuple[int, int, int, int]:
    """FaceDatasetTest.get_crop_coords (:507-518): a square of twice the face width around the landmarks -> (min_y, max_y, min_x, max_x)."""
    min_y, max_y = int(keypoints[:, 1].min()), int(keypoints[:, 1].max())
    min_x, max_x = int(keypoints[:, 0].min()), int(keypoints[:, 0].max())
    x_cen, y_cen = (min_x + max_x) // 2, (min_y + max_y) // 2
    side = max_x - min_x
    x0 = x_cen - side
    y0 = y_cen - side * 1.25
    return int(y0), int(y0 + side * 2), int(x0), int(x0 + side * 2)


class FaceRasteriser:
    """Edge maps / bounding-box masks of a clip at crop resolution, and one-hot labels, on `device`.

    lib: tests pass the CPU emulation build; product code leaves it None (the in-tree HIP library, no fallback)."""

    def __init__(self, device, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.device = torch.device(device)

    def _stream(self) -> Optional[int]:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    def rasterise(self, keypoints: Sequence[np.ndarray], crop: Optional[Tuple[int, int, int, int]] = None):
        """keypoints: F arrays (68,2) in frame coordinates.  crop: (min_y, max_y, min_x, max_x); default = crop_coords of the first
        frame (fix_crop_pos=True, :294-299).  Returns (edges (F,h,w) uint8, bbox (F,h,w) uint8, crop, bw)."""
        kp = np.stack([np.asarray(k, dtype=np.float64) for k in keypoints])
        if kp.ndim != 3 or kp.shape[1:] != (68, 2):
            raise ValueError(f"expected F x 68 x 2 key points, got {kp.shape}")
        if crop is None:
            crop = crop_coords(kp[0])
        kp = kp.copy()
        kp[:, :, 0] -= crop[2]                                     # read_keypoints (:497-505)
        kp[:, :, 1] -= crop[0]
        h, w = crop[1] - crop[0], crop[3] - crop[2]
        bw = max(1, h // 256)                                      # :295
        F = kp.shape[0]
        kp = np.ascontiguousarray(kp)
        # interp_points' Levenberg-Marquardt fits run on the host (34 per frame, microseconds each; csrc/lmfit.hpp reproduces scipy's
        # curve_fit to the last bit); the device samples and draws the fitted curves
        curves = np.empty((F, 34, 8), dtype=np.float64)
        rc = self.lib.tsnet_fit_face_curves(kp.ctypes.data, F, curves.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"tsnet_fit_face_curves failed ({rc}): {self.lib.tsnet_op_last_error().decode()}")
        kd = torch.from_numpy(kp).to(self.device)
        cd = torch.from_numpy(curves).to(self.device)
        edges = torch.empty((F, h, w), dtype=torch.uint8, device=self.device)
        bbox = torch.empty((F, h, w), dtype=torch.uint8, device=self.device)
        ctx = torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()
        with ctx:
            rc = self.lib.tsnet_raster_face(kd.data_ptr(), cd.data_ptr(), F, h, w, bw, edges.data_ptr(), bbox.data_ptr(), self._stream())
        if rc != 0:
            raise RuntimeError(f"tsnet_raster_face failed ({rc}): {self.lib.tsnet_op_last_error().decode()}")
        self._keep = (kd, cd)
        return edges, bbox, crop, bw

    def vl2ch(self, labels: torch.Tensor, num_classes: int) -> torch.Tensor:
        """utils/misc.py vl2ch (:50-67): (B,H,W) class indices -> (B,num_classes,H,W) one-hot float32."""
        lab = labels.to(self.device, dtype=torch.float32).contiguous()
        B, H, W = lab.shape
        out = torch.empty((B, num_classes, H, W), dtype=torch.float32, device=self.device)
        ctx = torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()
        with ctx:
            rc = self.lib.tsnet_vl2ch(lab.data_ptr(), B, H * W, num_classes, out.data_ptr(), self._stream())
        if rc != 0:
            raise RuntimeError(f"tsnet_vl2ch failed ({rc}): {self.lib.tsnet_op_last_error().decode()}")
        self._keep_l = lab
        return out


# ---------------------------------------------------------------------------------------------------------------- pose clips
_POSE_GROUPS = (("pose_keypoints_2d", 25), ("face_keypoints_2d", 70), ("hand_left_keypoints_2d", 21), ("hand_right_keypoints_2d", 21))
_FACE_RUNS = ((0, 17), (17, 22), (22, 27), (31, 36), (48, 55))           # consecutive runs of face_list (keypoint2img_posenorm.py:439-447)
_FACE_CHAINS = ((28, 31), (35, 28), (36, 37, 38, 39), (39, 40, 41, 36), (42, 43, 44, 45), (45, 46, 47, 42), (54, 55, 56, 57, 58, 59, 48))
_FINGERS = tuple((0,) + tuple(range(4 * i + 1, 4 * i + 5)) for i in range(5))


def _usable(points: np.ndarray) -> np.ndarray:
    """extract_valid_keypoints (keypoint2img_posenorm.py:242-262): a face / hand chain is kept only if ALL its points are confident
    (> 0.1 / > 0.01); body points are kept one by one (> 0.01).  (n,3) -> (n,2), dropped points zero."""
    conf, out = points[:, 2], np.zeros((points.shape[0], 2))
    if points.shape[0] == 25:
        keep = conf > 0.01
        out[keep] = points[keep, :2]
        return out
    chains = [tuple(range(a, b)) for a, b in _FACE_RUNS] + list(_FACE_CHAINS) if points.shape[0] == 70 else _FINGERS
    thr = 0.1 if points.shape[0] == 70 else 0.01
    for chain in chains:
        idx = list(chain)
        if np.all(conf[idx] > thr):
            out[idx] = points[idx, :2]
    return out


def read_openpose(path_or_text: str) -> np.ndarray:
    """One OpenPose frame file (or its text) -> (137,2) float64: body 25 | face 70 | left hand 21 | right hand 21 of the person with the
    largest vertical extent (read_keypoints_posenorm, keypoint2img_posenorm.py:11-41); unusable points are zero."""
    import json
    import os
    text = open(path_or_text, encoding="utf-8").read() if os.path.exists(path_or_text) else path_or_text
    chosen, extent = np.zeros((137, 2)), 0.0
    for person in json.loads(text)["people"]:
        groups = [_usable(np.asarray(person[key], dtype=np.float64).reshape(n, 3)) for key, n in _POSE_GROUPS]
        span = groups[0][:, 1].max() - groups[0][:, 1].min()
        if span > extent:
            extent, chosen = span, np.concatenate(groups)
    return chosen


def pose_crop_coords(points: np.ndarray, size: Tuple[int, int], scale: float = 1.5) -> Tuple[int, int, int, int]:
    """PoseDatasetTestVideo.get_crop_coords (dataset_video_pose.py:554-588, no random offset, aspect_ratio 0.5): a box of `scale` body heights,
    half as wide, clamped into the (w, h) frame -> (xs, ys, xe, ye).  points: (137,2) or (25,2)."""
    w, h = size
    body = np.asarray(points, dtype=np.float64)[:25]
    seen = body[body[:, 0] != 0]
    if seen.shape[0]:
        x_cen = int(seen[:, 0].min() + seen[:, 0].max()) // 2
        top = max(seen[:, 1].min(), min(body[15, 1], body[16, 1]))          # not above the eyes
        bottom = max(body[11, 1], body[14, 1])                               # the ankles, if seen
        if bottom == 0:
            bottom = seen[:, 1].max()
        y_cen, height = int(top + bottom) // 2, bottom - top
    else:
        x_cen, y_cen, height = w // 2, h // 2, h // 2
    half_h = int(min(h, max(h // 4, height * scale))) // 2
    half_w = int(half_h * 0.5)
    x_cen = max(half_w, min(w - half_w, x_cen))
    y_cen = max(half_h, min(h - half_h, y_cen))
    return x_cen - half_w, y_cen - half_h, x_cen + half_w, y_cen + half_h


def nearest_table(n_in: int, n_out: int) -> np.ndarray:
    """Source index of every output index of PIL's Image.resize(.., NEAREST) along one axis: the sampling position starts at scale / 2 and is
    ADVANCED by `scale` per output pixel in double precision (libImaging Geometry.c, ImagingScaleAffine), then truncated -- the running sum,
    not (i + 0.5) * scale, decides ties.  tests/test_raster_pose.py checks the table against PIL itself."""
    scale = float(n_in) / float(n_out)
    pos, tab = scale * 0.5, np.empty(n_out, dtype=np.int32)
    for i in range(n_out):
        tab[i] = min(int(pos), n_in - 1)
        pos += scale
    return tab


class PoseRasteriser:
    """Class-index skeleton labels, bounding-box masks and the model's 256 x 256 label tensors of a pose clip on `device`.

    lib: tests pass the CPU emulation build; product code leaves it None (the in-tree HIP library, no fallback)."""

    def __init__(self, device, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self.device = torch.device(device)
        self._keep = []

    def _stream(self) -> Optional[int]:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    def _ctx(self):
        return torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.tsnet_op_last_error().decode()}")

    def rasterise(self, points: Sequence[np.ndarray], size: Tuple[int, int], window: Optional[Tuple[int, int, int, int]] = None,
                  basic_point_only: bool = False, remove_face_labels: bool = False) -> torch.Tensor:
        """points: F arrays (137,2) in the coordinates of the (w, h) = size frame.  window (xs, ys, xe, ye): the crop that is kept
        (crop_person_region); default the whole frame.  Returns (F, ye-ys, xe-xs) uint8 class indices 0..24 on the device:
        im2vl(connect_keypoints(..)) of the reference."""
        pts = np.stack([np.asarray(p, dtype=np.float64) for p in points])
        if pts.ndim != 3 or pts.shape[1:] != (137, 2):
            raise ValueError(f"expected F x 137 x 2 points, got {pts.shape}")
        w, h = size
        xs, ys, xe, ye = window if window is not None else (0, 0, w, h)
        F = pts.shape[0]
        n = F * (ye - ys) * (xe - xs)
        pts = np.ascontiguousarray(pts)
        flags = (1 if basic_point_only else 0) | (2 if remove_face_labels else 0)
        curves = np.empty((F, 118, 8), dtype=np.float64)              # every stroke's line, fitted on the host as the reference fits it (csrc/lmfit.hpp)
        self._check(self.lib.tsnet_fit_pose_curves(pts.ctypes.data, F, flags, curves.ctypes.data), "tsnet_fit_pose_curves")
        pd = torch.from_numpy(pts).to(self.device)
        cd = torch.from_numpy(curves).to(self.device)
        buf = torch.empty((n + 3) // 4 * 4, dtype=torch.uint8, device=self.device)      # whole 32-bit words: the kernel raises bytes with word atomics
        with self._ctx():
            rc = self.lib.tsnet_raster_pose(pd.data_ptr(), cd.data_ptr(), F, h, w, xs, ys, xe, ye, flags, buf.data_ptr(), self._stream())
        self._check(rc, "tsnet_raster_pose")
        self._keep = [pd, cd, buf]
        return buf[:n].view(F, ye - ys, xe - xs)

    def bbox(self, labels: torch.Tensor) -> torch.Tensor:
        """get_bbox_image (dataset_video_pose.py:590-607) of (F,h,w) uint8 labels -> (F,h,w) uint8 0/255"""
        lab = labels.to(self.device).contiguous()
        F, h, w = lab.shape
        out = torch.empty_like(lab)
        with self._ctx():
            rc = self.lib.tsnet_label_bbox(lab.data_ptr(), F, h, w, out.data_ptr(), self._stream())
        self._check(rc, "tsnet_label_bbox")
        self._keep += [lab]
        return out

    def to_square(self, maps: torch.Tensor, img_size: Tuple[int, int] = (128, 256), binarise: bool = False) -> torch.Tensor:
        """Image.resize(img_size, NEAREST) + resize_square (dataset_video_pose.py:425-432, :471-477) of (F,h,w) uint8 maps -> (F,S,S) float32 with
        S = max(img_size); binarise gives the `!= 0` mask of :441."""
        m = maps.to(self.device).contiguous()
        F, h, w = m.shape
        ow, oh = img_size
        S = max(ow, oh)
        yt = torch.from_numpy(nearest_table(h, oh)).to(self.device)
        xt = torch.from_numpy(nearest_table(w, ow)).to(self.device)
        out = torch.empty((F, S, S), dtype=torch.float32, device=self.device)
        with self._ctx():
            rc = self.lib.tsnet_resize_pad(m.data_ptr(), F, h, w, yt.data_ptr(), xt.data_ptr(), oh, ow, (S - oh) // 2, (S - ow) // 2, S, S,
                                           int(binarise), out.data_ptr(), self._stream())
        self._check(rc, "tsnet_resize_pad")
        self._keep += [m, yt, xt]
        return out

    def clip_labels(self, points: Sequence[np.ndarray], size: Tuple[int, int], window=None, img_size=(128, 256)):
        """The label tensors of a clip as the data loader hands them to the model: class maps (F,256,256) float (vl2ch makes them one-hot),
        bounding-box masks (F,256,256) float 0/1, and the crop used (from the first frame when not given, :331-334)."""
        if window is None:
            window = pose_crop_coords(points[0], size)
        cls = self.rasterise(points, size, window)
        box = self.bbox(cls)
        return self.to_square(cls, img_size), self.to_square(box, img_size, binarise=True), window


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
