"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's pose-label rasterisation (pose clips).
Only tests/ may import this module.

Follows, function by function (test mode, opt.isTrain = False):
  valid_points       utils/keypoint2img_posenorm.py:242-262  extract_valid_keypoints (confidence thresholds per point group)
  select_person      :26-40                                  read_keypoints_posenorm: the person with the largest vertical extent
  crop_coords        dataset/dataset_video_pose.py:554-588   get_crop_coords (offset 0, aspect_ratio 0.5)
  skeleton_classes   keypoint2img_posenorm.py:265-311        connect_keypoints, drawing CLASS INDICES instead of colours: utils/misc.py im2vl
                                                             (:27-47) maps every colour of define_edge_lists (:396-448) to one index, so the
                                                             colour image never needs to exist
  label_bbox         dataset_video_pose.py:590-607           get_bbox_image
  resize_square      :425-432, :471-477                      Image.resize(img_size, NEAREST) + centred zero padding to a square (PIL itself is
                                                             called, as the reference does)
interp_points / the brush come from oracle/raster_oracle.py (the same functions of utils/keypoint2img.py; keypoint2img_posenorm.py:469-516 are
copies of them) and call scipy.optimize.curve_fit exactly as the reference does.  Pinned by tests/golden/g9_raster_pose.npz, captured by
oracle/capture_raster_pose_goldens.py from the imported reference (tests/test_raster_pose.py: equal on all 60 demo frames)."""
from __future__ import annotations

import json

import numpy as np
from PIL import Image, ImageOps

from oracle.raster_oracle import interp_points

POSE_EDGES = [[17, 15], [15, 0], [0, 16], [16, 18], [0, 1], [1, 8], [1, 2], [2, 3], [3, 4], [1, 5], [5, 6], [6, 7],
              [8, 9], [9, 10], [10, 11], [8, 12], [12, 13], [13, 14]]
FOOT_EDGES = [[11, 24], [11, 22], [22, 23], [14, 21], [14, 19], [19, 20]]
POSE_CLASSES = list(range(1, 19)) + [15, 15, 15, 18, 18, 18]       # the feet reuse the colours of edges 14 and 17 (:417-420)
HAND_FINGERS = [[0, 1, 2, 3, 4], [0, 5, 6, 7, 8], [0, 9, 10, 11, 12], [0, 13, 14, 15, 16], [0, 17, 18, 19, 20]]
HAND_CLASSES = [19, 20, 21, 22, 23]
FACE_CLASS = 24
FACE_LIST = [[list(range(0, 17))], [list(range(17, 22))], [list(range(22, 27))], [[28, 31], list(range(31, 36)), [35, 28]],
             [[36, 37, 38, 39], [39, 40, 41, 36]], [[42, 43, 44, 45], [45, 46, 47, 42]], [list(range(48, 55)), [54, 55, 56, 57, 58, 59, 48]]]


def valid_points(pts: np.ndarray) -> np.ndarray:
    """(n,3) x, y, confidence -> (n,2) with the points the drawing may use; the rest zero"""
    n = pts.shape[0]
    out = np.zeros((n, 2))
    if n == 70:
        for edge_list in FACE_LIST:
            for edge in edge_list:
                if (pts[edge, 2] > 0.1).all():
                    out[edge] = pts[edge, :2]
    elif n == 21:
        for edge in HAND_FINGERS:
            if (pts[edge, 2] > 0.01).all():
                out[edge] = pts[edge, :2]
    else:
        ok = pts[:, 2] > 0.01
        out[ok] = pts[ok, :2]
    return out


def select_person(json_text: str) -> np.ndarray:
    """OpenPose frame file -> (137,2): pose | face | left hand | right hand of the tallest person"""
    best, best_len = np.zeros((137, 2)), 0
    for person in json.loads(json_text)["people"]:
        groups = [valid_points(np.array(person[k]).reshape(n, 3)) for k, n in
                  (("pose_keypoints_2d", 25), ("face_keypoints_2d", 70), ("hand_left_keypoints_2d", 21), ("hand_right_keypoints_2d", 21))]
        y = groups[0][:, 1]
        if y.max() - y.min() > best_len:
            best_len = y.max() - y.min()
            best = np.concatenate(groups, axis=0)
    return best


def crop_coords(pose_pts: np.ndarray, size, scale=None):
    w, h = size
    valid = pose_pts[:, 0] != 0
    x, y = pose_pts[valid, 0], pose_pts[valid, 1]
    x_cen = int(x.min() + x.max()) // 2 if x.shape[0] else w // 2
    if y.shape[0]:
        y_min = max(y.min(), min(pose_pts[15, 1], pose_pts[16, 1]))
        y_max = max(pose_pts[11, 1], pose_pts[14, 1])
        if y_max == 0:
            y_max = y.max()
        y_cen = int(y_min + y_max) // 2
        y_len = y_max - y_min
    else:
        y_cen = y_len = h // 2
    if scale is None:
        scale = 1.5
    bh = int(min(h, max(h // 4, y_len * scale))) // 2
    bw = int(bh * 0.5)
    x_cen = max(bw, min(w - bw, x_cen))
    y_cen = max(bh, min(h - bh, y_cen))
    return [x_cen - bw, y_cen - bh, x_cen + bw, y_cen + bh], scale


def _stroke(im, x, y, bw, cls, end_points):
    if x is None or not x.size:
        return
    h, w = im.shape
    for i in range(-bw, bw):
        for j in range(-bw, bw):
            im[np.clip(y + i, 0, h - 1), np.clip(x + j, 0, w - 1)] = cls
    if end_points:
        for i in range(-2 * bw, 2 * bw):
            for j in range(-2 * bw, 2 * bw):
                if i * i + j * j < 4 * bw * bw:
                    im[np.clip(np.array([y[0], y[-1]]) + i, 0, h - 1), np.clip(np.array([x[0], x[-1]]) + j, 0, w - 1)] = cls


def skeleton_classes(pts: np.ndarray, size, basic_point_only=False, remove_face_labels=False) -> np.ndarray:
    """(137,2) points, frame size (w, h) -> (h, w) uint8 class indices = im2vl(connect_keypoints(...))"""
    pose, face, hands = pts[:25], pts[25:95], (pts[95:116], pts[116:137])
    w, h = size
    im = np.zeros((h, w), np.uint8)
    ph = int(pose[:, 1].max() - pose[:, 1].min())
    bw = min(max(1, ph // 150), 5)
    edges = POSE_EDGES + ([] if basic_point_only else FOOT_EDGES)
    for i, edge in enumerate(edges):
        x, y = pose[edge, 0], pose[edge, 1]
        if 0 not in x:
            cx, cy = interp_points(x, y)
            _stroke(im, cx, cy, bw, POSE_CLASSES[i], True)
    if not basic_point_only:
        bw = min(max(1, ph // 450), 3)
        for hand in hands:
            for i, finger in enumerate(HAND_FINGERS):
                for j in range(len(finger) - 1):
                    x, y = hand[finger[j:j + 2], 0], hand[finger[j:j + 2], 1]
                    if 0 not in x:
                        cx, cy = interp_points(x, y)
                        _stroke(im, cx, cy, bw, HAND_CLASSES[i], False)
        if not remove_face_labels:
            for edge_list in FACE_LIST:
                for edge in edge_list:
                    for i in range(0, max(1, len(edge) - 1)):
                        x, y = face[edge[i:i + 2], 0], face[edge[i:i + 2], 1]
                        if 0 not in x:
                            try:
                                cx, cy = interp_points(x, y)
                            except RuntimeError:
                                continue
                            _stroke(im, cx, cy, bw, FACE_CLASS, False)
    return im


def label_bbox(cls_map: np.ndarray) -> np.ndarray:
    ys, xs = np.nonzero(cls_map)
    h, w = cls_map.shape
    out = np.zeros((h, w), np.uint8)
    x0, x1 = int(max(0.0, xs.min() - w // 16)), int(min(w, xs.max() + w // 16))
    y0, y1 = int(max(0.0, ys.min() - h // 16)), int(min(h, ys.max() + h // 16))
    out[y0:y1, x0:x1] = 255
    return out


def resize_square(arr: np.ndarray, img_size=(128, 256)) -> np.ndarray:
    im = Image.fromarray(arr).resize(img_size, resample=Image.NEAREST)
    w, h = im.size
    s = max(w, h)
    dw, dh = s - w, s - h
    return np.asarray(ImageOps.expand(im, (dw // 2, dh // 2, dw - dw // 2, dh - dh // 2)))
