"""Loader-side preprocessing of a pose driving clip -- OUTSIDE the product package (`dataset/smooth_pose_keypoint.py` is
an offline tool and the pose-norm reader is data-loader code; neither is on the section 8-f list).  Kept as a tool because the clip harness
(tools/demo_pose_clip.py) can apply it in front of the device rasteriser, and pinned to the reference's outputs by tests/test_raster_pose.py:
  shift_into_crop      dataset/dataset_video_pose.py:554-588 (the crop shift of read_pts_posenorm's input)
  pose_limb_rescale    utils/keypoint2img_posenorm.py:70-226 (read_pts_posenorm with ref_pts_length "fm" / "mf")
  smooth_points / smooth_clip / read_smooth_openpose   dataset/smooth_pose_keypoint.py:85-114 and the json_tricks files it writes"""
from __future__ import annotations

from typing import Tuple

import numpy as np

_POSE_GROUPS = (("pose_keypoints_2d", 25), ("face_keypoints_2d", 70), ("hand_left_keypoints_2d", 21), ("hand_right_keypoints_2d", 21))
_FINGERS = tuple((0,) + tuple(range(4 * i + 1, 4 * i + 5)) for i in range(5))
_POSE_EDGES = ((17, 15), (15, 0), (0, 16), (16, 18), (0, 1), (1, 8), (1, 2), (2, 3), (3, 4), (1, 5), (5, 6), (6, 7),
               (8, 9), (9, 10), (10, 11), (8, 12), (12, 13), (13, 14), (11, 24), (11, 22), (22, 23), (14, 21), (14, 19), (19, 20))
_HAND_SEGS = tuple((f[j], f[j + 1]) for f in _FINGERS for j in range(4))


def shift_into_crop(points: np.ndarray, crop: Tuple[int, int, int, int]) -> np.ndarray:
    """First step of read_pts_posenorm (keypoint2img_posenorm.py:70-74): points with both coordinates non-zero move into the crop's frame;
    points with a zero coordinate (= missing) stay as they are.  (137,2) -> (137,2)."""
    out = np.array(points, dtype=np.float64, copy=True)
    live = (out[:, 0] != 0) & (out[:, 1] != 0)
    out[live] -= np.array([crop[0], crop[1]], dtype=np.float64)
    return out


def pose_limb_rescale(points: np.ndarray, mode: str, height: int) -> np.ndarray:
    """The body-shape transfer read_pts_posenorm applies to opposite-sex pairs (keypoint2img_posenorm.py:90-226; `mode` = ref_pts_length:
    "fm" driving female -> source male, "mf" the reverse; PoseDatasetTestVideo.__getitem__ :306-314).  On the 137 points of one frame, in crop
    coordinates (`height` = crop height): the torso 1 -> 8 is scaled by 0.85 / 1.2 and the shoulders 1 -> 2, 1 -> 5 by 0.9 / 1.2; elbows and
    wrists keep their offsets from the moved parent; hips keep their offsets from the moved pelvis; the knees keep their thighs' share of
    the distance to the bottom edge; ankles and feet stay; each hand is re-attached to its (moved) wrist finger joint by finger joint; the face
    stays.  Arithmetic on 137 points per frame: host work."""
    if mode not in ("fm", "mf"):
        raise ValueError("mode must be 'fm' or 'mf'")
    pts = np.array(points, dtype=np.float64, copy=True)
    body = pts[:25]
    missing = lambda p: p[0] == 0 or p[1] == 0
    length = {}
    for a, b in _POSE_EDGES:
        length[(a, b)] = 0.0 if (missing(body[a]) or missing(body[b])) else float(np.linalg.norm(body[a] - body[b]))
    torso = length[(1, 8)]
    torso_new = torso * (0.85 if mode == "fm" else 1.2)
    shoulder = 0.9 if mode == "fm" else 1.2
    new = body.copy()
    for i in (2, 5):
        if not missing(body[i]):
            new[i] = new[1] + (body[i] - body[1]) * shoulder
    for parent, child in ((2, 3), (5, 6), (3, 4), (6, 7)):           # upper arms, then forearms: children follow their moved parents
        if length[(parent, child)]:
            new[child] = new[parent] + (body[child] - body[parent])
    with np.errstate(divide="ignore", invalid="ignore"):              # a missing neck or pelvis gives what the reference gives: non-finite points
        new[8] = body[1] + torso_new * ((body[8] - body[1]) / torso)
    for i in (9, 12):
        new[i] = new[8] + (body[i] - body[8])
    for hip, knee in ((9, 10), (12, 13)):
        if length[(hip, knee)]:
            with np.errstate(divide="ignore", invalid="ignore"):
                thigh = (height - new[hip][1]) * (length[(hip, knee)] / (height - body[hip][1]))
                new[knee] = new[hip] + thigh * ((body[knee] - body[hip]) / length[(hip, knee)])
    pts[:25] = new
    for off, wrist in ((95, 7), (116, 4)):                            # left hand <- left wrist (7), right hand <- right wrist (4)
        hand = pts[off:off + 21].copy()
        seg_len = {sg: (0.0 if (missing(hand[sg[0]]) or missing(hand[sg[1]])) else float(np.linalg.norm(hand[sg[0]] - hand[sg[1]]))) for sg in _HAND_SEGS}
        moved = hand.copy()
        moved[0] = new[wrist]
        for j in range(4):                                            # joint level by joint level, outwards from the wrist
            for parent, child in (sg for f in _FINGERS for sg in ((f[j], f[j + 1]),)):
                if seg_len[(parent, child)]:
                    moved[child] = moved[parent] + (hand[child] - hand[parent])
        pts[off:off + 21] = moved
    return pts


def smooth_points(seq: np.ndarray) -> np.ndarray:
    """dataset/smooth_pose_keypoint.py smooth_points (:85-114) on one point group of a clip, (F,n,2) -> (F,n,2): per point a moving average over
    frames j-2 .. j+2 that divides the SUM OF ALL five positions (missing = zero) by the number of frames in which the point is present;
    first and last frame unchanged, frames 1 / 2 / F-2 use the windows 0..2 / 0..4 / F-3..F-1.  (The reference's closing
    `new[valid == 0] = 0` compares a Python list with 0 and never selects anything: missing points keep their averaged value.  Reproduced.)"""
    seq = np.asarray(seq, dtype=np.float64)
    F = seq.shape[0]
    if F < 6:
        raise ValueError("smooth_points needs at least six frames")
    csum = np.cumsum(seq, axis=0)
    present = ((seq[:, :, 0] != 0) & (seq[:, :, 1] != 0)).astype(np.int64)
    cnum = np.cumsum(present, axis=0)
    out = np.zeros_like(seq)
    for p in range(seq.shape[1]):
        s, c, x = csum[:, p], cnum[:, p], seq[:, p]
        out[0, p] = x[0]
        out[1, p] = s[2] / c[2] if c[2] else x[1]
        out[2, p] = s[4] / c[4] if c[4] else x[2]
        for j in range(3, F - 2):
            n = c[j + 2] - c[j - 3]
            out[j, p] = (s[j + 2] - s[j - 3]) / n if n else x[j]
        n = c[-1] - c[-4]
        out[F - 2, p] = (s[-1] - s[-4]) / n if n else x[F - 2]
        out[F - 1, p] = x[-1]
    return out


def smooth_clip(points: np.ndarray) -> np.ndarray:
    """smooth_points on the four groups of a clip's (F,137,2) points, as smooth_pose_keypoint.py's main does (:153-160)"""
    pts = np.asarray(points, dtype=np.float64)
    return np.concatenate([smooth_points(pts[:, a:b]) for a, b in ((0, 25), (25, 95), (95, 116), (116, 137))], axis=1)


def read_smooth_openpose(path: str) -> np.ndarray:
    """A file written by smooth_pose_keypoint.py (json_tricks: every array is {"__ndarray__": nested lists, "dtype", "shape"}), read by
    PoseDatasetTestVideo.__getitem__ (:355-359) -> (F,137,2)"""
    import json
    d = json.load(open(path, encoding="utf-8"))
    arr = lambda v: np.asarray(v["__ndarray__"] if isinstance(v, dict) else v, dtype=np.float64)
    return np.concatenate([arr(d[k]) for k, _ in _POSE_GROUPS], axis=1)
