"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's input rasterisation.
Only tests/ may import this module.

Follows, function by function:
  crop_coords        dataset/dataset_video_face.py:507-518  FaceDatasetTest.get_crop_coords
  crop_keypoints     :497-505                               read_keypoints (the file parse is np.loadtxt(delimiter=','), :457)
  sub_edges          :271-280 (part_list) and :473-477      the split of every face-part polyline into 3-point pieces
  interp_points      utils/keypoint2img.py:319-354          scipy curve_fit of x -> a x^2 + b x + c (3 points) or a x + b (2 points),
                                                            fitted along the axis with the larger extent, sampled at
                                                            np.linspace(x0, xn, ceil(xn - x0)) and truncated by astype(int)
  draw_edge          utils/keypoint2img.py:298-316          a [-bw, bw) x [-bw, bw) brush, clipped to the image
  face_edge_map      dataset_video_face.py:466-481          get_face_image
  bbox_mask          :483-495                               get_bbox_image
  vl2ch              utils/misc.py:50-67                    one-hot labels

The arithmetic that decides pixels lives in scipy.optimize.curve_fit (Levenberg-Marquardt from p0 = ones): this restatement calls
it exactly as the reference does, so it reproduces the reference bit for bit (pinned by tests/golden/g7_raster_face.npz, captured
by oracle/capture_raster_goldens.py from the imported reference).  The step after these functions in the reference's data loader,
skimage.transform.resize + img_as_bool to 256 x 256 (:316-317), is restated in oracle/skimage_resize.py -- PARITY UNPINNED: skimage is
absent from this image, so there is nothing to pin that one against."""
from __future__ import annotations

import math
import warnings

import numpy as np
from scipy.optimize import curve_fit

# FaceDatasetTest.part_list (dataset_video_face.py:271-280): polylines over the 68 landmarks
PART_LIST = [[list(range(0, 17))], [list(range(17, 22))], [list(range(22, 27))], [[28, 31], list(range(31, 36)), [35, 28]],
             [[36, 37, 38, 39], [39, 40, 41, 36]], [[42, 43, 44, 45], [45, 46, 47, 42]],
             [list(range(48, 55)), [54, 55, 56, 57, 58, 59, 48], list(range(60, 65)), [64, 65, 66, 67, 60]]]


def sub_edges(edge_len: int = 3):
    """every polyline cut into pieces of edge_len points that share their end points (:473-477)"""
    out = []
    for edge_list in PART_LIST:
        for edge in edge_list:
            for i in range(0, max(1, len(edge) - 1), edge_len - 1):
                out.append(list(edge[i:i + edge_len]))
    return out


def crop_coords(keypoints: np.ndarray):
    min_y, max_y = int(keypoints[:, 1].min()), int(keypoints[:, 1].max())
    min_x, max_x = int(keypoints[:, 0].min()), int(keypoints[:, 0].max())
    x_cen, y_cen = (min_x + max_x) // 2, (min_y + max_y) // 2
    w = h = max_x - min_x
    min_x = x_cen - w
    min_y = y_cen - h * 1.25
    max_x = min_x + w * 2
    max_y = min_y + h * 2
    return int(min_y), int(max_y), int(min_x), int(max_x)


def crop_keypoints(keypoints: np.ndarray, crop):
    kp = np.array(keypoints, dtype=np.float64, copy=True)
    kp[:, 0] -= crop[2]
    kp[:, 1] -= crop[0]
    return kp


def _quadratic(x, a, b, c):
    return a * x ** 2 + b * x + c


def _linear(x, a, b):
    return a * x + b


def interp_points(x, y):
    if abs(x[:-1] - x[1:]).max() < abs(y[:-1] - y[1:]).max():
        cy, cx = interp_points(y, x)
        if cy is None:
            return None, None
        return cx, cy
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if len(x) < 3:
            popt, _ = curve_fit(_linear, x, y)
        else:
            popt, _ = curve_fit(_quadratic, x, y)
            if abs(popt[0]) > 1:
                return None, None
    if x[0] > x[-1]:
        x = list(reversed(x))
        y = list(reversed(y))
    curve_x = np.linspace(x[0], x[-1], math.ceil(x[-1] - x[0]))
    curve_y = _linear(curve_x, *popt) if len(x) < 3 else _quadratic(curve_x, *popt)
    return curve_x.astype(int), curve_y.astype(int)


def draw_edge(im: np.ndarray, x, y, bw: int = 1):
    if x is None or not x.size:
        return
    h, w = im.shape
    for i in range(-bw, bw):
        for j in range(-bw, bw):
            yy = np.maximum(0, np.minimum(h - 1, y + i))
            xx = np.maximum(0, np.minimum(w - 1, x + j))
            im[yy, xx] = 255


def face_edge_map(keypoints: np.ndarray, size, bw: int) -> np.ndarray:
    w, h = size
    im = np.zeros((h, w), np.uint8)
    for se in sub_edges():
        cx, cy = interp_points(keypoints[se, 0], keypoints[se, 1])
        draw_edge(im, cx, cy, bw=bw)
    return im


def bbox_mask(keypoints: np.ndarray, size) -> np.ndarray:
    w, h = size
    im = np.zeros((h, w), np.uint8)
    x_min, x_max = keypoints[:, 0].min(), keypoints[:, 0].max()
    y_min, y_max = keypoints[:, 1].min(), keypoints[:, 1].max()
    x_margin, y_margin = w // 16, h // 16
    x_min = int(max(0.0, x_min - x_margin))
    x_max = int(min(w, x_max + x_margin))
    y_min = int(max(0.0, y_min - y_margin))
    y_max = int(min(h, y_max + y_margin))
    im[y_min:y_max, x_min:x_max] = 255
    return im


def vl2ch(labels: np.ndarray, num_classes: int) -> np.ndarray:
    """(B,H,W) class indices -> (B,num_classes,H,W) one-hot float32 (utils/misc.py:50-67: 2 classes for faces, 25 for poses)"""
    out = np.zeros((labels.shape[0], num_classes) + labels.shape[1:], np.float32)
    for c in range(num_classes):
        out[:, c] = labels == c
    return out
