"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the last step of the reference's face / pose data loaders:
    np.asarray(img_as_bool(resize(label_map, (256, 256))), dtype=np.uint8)            dataset/dataset_video_face.py:104-106, 325-326, 397-398
with skimage.transform.resize and skimage.img_as_bool of scikit-image 0.18.3 (the reference's requirements.txt:7).

PARITY UNPINNED.  scikit-image is a third-party dependency that is absent from this image and from /root/reference, so there is nothing
to run the restatement against and the reference's tests hold no vector for it.  What follows restates the PUBLISHED algorithm of that
version (skimage/transform/_warps.py: resize, warp; skimage/transform/_warps_cy.pyx: _warp_fast; skimage/_shared/interpolation.pxd:
bilinear_interpolation, coord_map; skimage/util/dtype.py: img_as_bool; scipy/ndimage/filters.py: gaussian_filter, gaussian_filter1d,
correlate1d) as I read it:

  1. anti_aliasing defaults to True for a non-bool input; sigma = max(0, (in / out - 1) / 2) per axis.  The Gaussian filter runs on the
     image AS PASSED -- a uint8 array of 0 / 255 -- so scipy writes every 1-D pass back into a uint8 array with a C cast (truncation):
     `filter_on_uint8=True`.  (scikit-image 0.19 moved the float conversion in front of the filter; `filter_on_uint8=False` is that
     reading.)  Kernel radius int(4 sigma + 0.5): for crops below 320 pixels (sigma < 0.125) the kernel is [1] and the filter is the identity
     -- every demo clip of the reference (crop 292) is in that case, so the two readings coincide there.  Boundary 'mirror'
     (np.pad's 'reflect').
  2. warp() converts to float64 in [0, 1] (/ 255), then _warp_fast samples output pixel (r, c) at input coordinates
     (fr (r + 0.5) - 0.5, fc (c + 0.5) - 0.5), f = in / out, by bilinear interpolation between floor and ceil, out-of-range neighbours
     mirrored about the edge pixel centres (mode 'reflect'); clip to the input's range.
  3. img_as_bool of a float image: value > 0.5.

The affine map in step 2 comes out of AffineTransform.estimate in the reference (a least-squares solve whose result equals the scale /
offset above to a few ulp); at the exact-integer coordinates where an ulp could change floor(), bilinear interpolation is continuous, so
the sample changes by ~1e-16 -- a pixel whose sample is within 1e-12 of 0.5 is reported by `ties()` instead of being decided silently."""
from __future__ import annotations

import math

import numpy as np


def _mirror(i: np.ndarray, n: int) -> np.ndarray:
    """coord_map(mode='R'): reflect about the edge pixel centres, any distance"""
    if n == 1:
        return np.zeros_like(i)
    cmax = n - 1
    i = np.abs(i)
    q, r = np.divmod(i, cmax)
    return np.where(q % 2 == 1, cmax - r, r)


def _gaussian1d_uint8(a: np.ndarray, sigma: float, axis: int, keep_float: bool) -> np.ndarray:
    """scipy.ndimage.gaussian_filter1d(order=0, mode='mirror', truncate=4.0) along one axis; output in the input's dtype unless keep_float"""
    lw = int(4.0 * sigma + 0.5)
    x = np.arange(-lw, lw + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    w = phi / phi.sum()
    n = a.shape[axis]
    src = np.moveaxis(a, axis, 0).astype(np.float64)
    idx = np.arange(n)
    out = src * w[lw]                                            # correlate1d's symmetric form: centre, then pairs outwards
    for j in range(1, lw + 1):
        out = out + (src[_mirror(idx + j, n)] + src[_mirror(idx - j, n)]) * w[lw + j]
    if not keep_float:
        out = np.trunc(out).astype(a.dtype)                      # the C cast of NI_LineBufferToArray
    return np.moveaxis(out, 0, axis)


def resize_float(img: np.ndarray, out_shape=(256, 256), filter_on_uint8: bool = True) -> np.ndarray:
    """skimage.transform.resize(img, out_shape) for a 2-D uint8 image with the defaults the reference uses -> float64 in [0, 1]"""
    assert img.ndim == 2 and img.dtype == np.uint8
    H, W = img.shape
    OH, OW = out_shape
    f = (H / OH, W / OW)
    work = img if filter_on_uint8 else img.astype(np.float64) / 255.0
    for axis in (0, 1):
        sigma = max(0.0, (f[axis] - 1.0) / 2.0)
        if sigma > 1e-15:
            work = _gaussian1d_uint8(work, sigma, axis, keep_float=not filter_on_uint8)
    im = work.astype(np.float64) / 255.0 if filter_on_uint8 else work
    r = f[0] * (np.arange(OH) + 0.5) - 0.5
    c = f[1] * (np.arange(OW) + 0.5) - 0.5
    r0, c0 = np.floor(r), np.floor(c)
    r1, c1 = np.ceil(r), np.ceil(c)
    dr, dc = (r - r0)[:, None], (c - c0)[None, :]
    ir0, ir1 = _mirror(r0.astype(np.int64), H), _mirror(r1.astype(np.int64), H)
    ic0, ic1 = _mirror(c0.astype(np.int64), W), _mirror(c1.astype(np.int64), W)
    top = (1 - dc) * im[ir0][:, ic0] + dc * im[ir0][:, ic1]
    bot = (1 - dc) * im[ir1][:, ic0] + dc * im[ir1][:, ic1]
    out = (1 - dr) * top + dr * bot
    return np.clip(out, im.min(), im.max())


def resize_bool(img: np.ndarray, out_shape=(256, 256), filter_on_uint8: bool = True) -> np.ndarray:
    """np.asarray(img_as_bool(resize(img, out_shape)), dtype=np.uint8): 0 / 1"""
    return (resize_float(img, out_shape, filter_on_uint8) > 0.5).astype(np.uint8)


def ties(img: np.ndarray, out_shape=(256, 256), filter_on_uint8: bool = True, eps: float = 1e-12) -> int:
    """number of output pixels whose sample lies within eps of the 0.5 threshold (their value depends on the last ulp of the reference's map)"""
    return int((np.abs(resize_float(img, out_shape, filter_on_uint8) - 0.5) <= eps).sum())
