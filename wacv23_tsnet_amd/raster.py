"""Host side of the device rasterisation: key-point files of a clip -> the label tensors
`set_test_input` takes.  The reference does this per frame on the CPU in its data loader
(dataset/dataset_video_face.py:283-330, FaceDatasetTest.__getitem__); here the key points of a whole clip go to the device once
and three kernels (csrc/raster.hpp) produce every frame's edge map, bounding-box mask and one-hot label.

Only the crop arithmetic (a handful of integer operations per clip, dataset_video_face.py:507-518) stays on the host."""
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
        kd = torch.from_numpy(kp).to(self.device)
        edges = torch.empty((F, h, w), dtype=torch.uint8, device=self.device)
        bbox = torch.empty((F, h, w), dtype=torch.uint8, device=self.device)
        ctx = torch.cuda.device(self.device) if self.device.type == "cuda" else _Null()
        with ctx:
            rc = self.lib.tsnet_raster_face(kd.data_ptr(), F, h, w, bw, edges.data_ptr(), bbox.data_ptr(), self._stream())
        if rc != 0:
            raise RuntimeError(f"tsnet_raster_face failed ({rc}): {self.lib.tsnet_op_last_error().decode()}")
        self._keep = kd
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


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
