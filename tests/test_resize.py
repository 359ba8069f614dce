"""CPU tier: the 256 x 256 resize of the reference's loaders (skimage.transform.resize + img_as_bool, dataset_video_face.py:316-317).
PARITY UNPINNED: scikit-image 0.18.3 cannot be run here, so these tests hold only what can be held -- the product-side helper
(wacv23_tsnet_amd.demo.resize_label -> tsnet_resize_label, HIP kernels run here through the emulation build) equals the oracle restatement (oracle/skimage_resize.py) bit for bit, the restatement has the
properties the published algorithm implies, and in the interior it equals torch's bilinear resampling, which uses the same
(o + 0.5) f - 0.5 sampling grid."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import skimage_resize as SR
from wacv23_tsnet_amd import demo


def edge_map(h, w, seed):
    g = np.random.default_rng(seed)
    img = np.zeros((h, w), np.uint8)
    for _ in range(12):                                   # random thick polylines, like the rasterised face edges
        y, x = g.integers(4, h - 4), g.integers(4, w - 4)
        for _ in range(60):
            y = int(np.clip(y + g.integers(-2, 3), 1, h - 2)); x = int(np.clip(x + g.integers(-2, 3), 1, w - 2))
            img[y - 1:y + 1, x - 1:x + 1] = 255
    return img


@pytest.mark.parametrize("h,w", [(292, 292), (256, 256), (300, 281), (512, 512), (640, 400), (200, 230), (1100, 900)])
def test_product_kernels_equal_the_restatement(emu_lib, h, w):
    """tsnet_resize_label (the emulation build runs the same kernels) against oracle/skimage_resize.py, two frames per call"""
    imgs = np.stack([edge_map(h, w, seed) for seed in range(2)])
    got = demo.resize_label(torch.from_numpy(imgs), lib=emu_lib).numpy()
    assert got.shape == (2, 256, 256)
    for f in range(2):
        assert np.array_equal(got[f], SR.resize_bool(imgs[f]).astype(np.float32)), f


def test_identity_and_small_crops_do_not_blur():
    img = edge_map(256, 256, 3)
    assert np.array_equal(SR.resize_bool(img), (img > 0).astype(np.uint8))             # same size: the image itself
    img = edge_map(292, 292, 4)                                                        # the reference's demo clips: sigma 0.07, kernel [1]
    assert np.array_equal(SR.resize_float(img, filter_on_uint8=True), SR.resize_float(img, filter_on_uint8=False))
    assert SR.ties(img) == 0                                                           # 73 / 64 never samples midway between two pixels
    t = F.interpolate(torch.from_numpy(img)[None, None].double() / 255.0, size=(256, 256), mode="bilinear", align_corners=False)[0, 0].numpy()
    assert np.abs(SR.resize_float(img)[2:-2, 2:-2] - t[2:-2, 2:-2]).max() < 1e-12    # the same sampling grid away from the borders


def test_large_crops_are_filtered_on_the_byte_image():
    img = edge_map(512, 512, 5)
    a, b = SR.resize_float(img, filter_on_uint8=True), SR.resize_float(img, filter_on_uint8=False)
    assert 0 < np.abs(a - b).max() < 2.5 / 255           # the two readings differ by the truncation of two 1-D passes, at most
    assert SR.resize_bool(img).sum() > 0


def test_mirrored_borders():
    img = np.zeros((300, 300), np.uint8); img[0, :] = 255; img[:, -1] = 255
    out = SR.resize_float(img)
    assert out[0].min() > 0.4 and out[:, -1].min() > 0.4 and out[5:-5, 5:-5].max() == 0


@pytest.mark.gpu
def test_product_kernels_on_the_gpu():
    """the same comparison through the HIP library on cuda:0 (three crop sizes: no filter, a 5-tap and a 15-tap Gaussian)"""
    for h, w in ((292, 292), (512, 640), (1100, 900)):
        imgs = np.stack([edge_map(h, w, seed) for seed in range(3)])
        got = demo.resize_label(torch.from_numpy(imgs).cuda()).cpu().numpy()
        for f in range(3):
            assert np.array_equal(got[f], SR.resize_bool(imgs[f]).astype(np.float32)), (h, w, f)
