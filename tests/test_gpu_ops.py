"""GPU tier: operator-level parity of the HIP kernels against PyTorch CPU fp32 ops, through the C ABI."""
import pytest
import torch

import op_cases as oc

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-5     # fp32 reassociation noise on O(1) values


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    from wacv23_tsnet_amd import _lib
    return _lib.load()       # raises if the HIP extension is missing: no fallback


# shapes that select each of the four tile configurations on the real dispatch heuristic
@pytest.mark.parametrize("shape", [(4, 32, 32, 64, 512), (4, 32, 32, 64, 128), (2, 9, 7, 16, 40), (1, 32, 32, 8, 3), (12, 32, 32, 512, 512)])
def test_conv_tiles(lib, shape):
    N, H, W, Cin, Cout = shape
    assert oc.conv_case(lib, DEV, N, H, W, Cin, Cout, 3, 1, 1, True, norm=True) < TOL


@pytest.mark.parametrize("k,stride,pad,reflect", [(7, 1, 3, True), (3, 2, 1, False), (3, 1, 1, True), (1, 1, 0, False)])
@pytest.mark.parametrize("norm", [False, True])
def test_conv_kinds(lib, k, stride, pad, reflect, norm):
    assert oc.conv_case(lib, DEV, 2, 24, 20, 8, 24, k, stride, pad, reflect, norm=norm) < TOL


def test_conv_stem_shape(lib):
    assert oc.conv_case(lib, DEV, 2, 64, 64, 8, 64, 7, 1, 3, True) < TOL


def test_conv_head_tanh(lib):
    assert oc.conv_case(lib, DEV, 1, 32, 32, 64, 3, 7, 1, 3, True, norm=True, act=1) < TOL


def test_conv_fuse_shape(lib):
    assert oc.conv_case(lib, DEV, 2, 16, 16, 1024, 1024, 3, 1, 1, True, norm=True) < 1e-4


def test_conv_ragged_m_and_no_bias(lib):
    assert oc.conv_case(lib, DEV, 3, 5, 7, 32, 130, 3, 1, 1, True, bias=False) < TOL


@pytest.mark.parametrize("C,H,W", [(8, 6, 5), (64, 64, 64), (24, 9, 3), (512, 32, 32), (1024, 8, 8)])
@pytest.mark.parametrize("relu,resid", [(True, False), (False, True)])
def test_instnorm(lib, C, H, W, relu, resid):
    assert oc.instnorm_case(lib, DEV, 2, H, W, C, relu, resid) < TOL


def test_instnorm_large_mean(lib):
    assert oc.instnorm_case(lib, DEV, 1, 16, 16, 8, False, False, offset=300.0) < 2e-3


@pytest.mark.parametrize("norm", [False, True])
def test_upsample(lib, norm):
    assert oc.upsample_case(lib, DEV, 2, 16, 12, 64, norm) < TOL


@pytest.mark.parametrize("mask_mode", ["bernoulli", "ones", "zeros", "soft"])
def test_flow_masks(lib, mask_mode):
    df, dw = oc.flow_case(lib, DEV, 2, 8, 8, 512, mask_mode)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_full_size(lib):
    df, dw = oc.flow_case(lib, DEV, 2, 32, 32, 512, "bernoulli", spike=True)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_ragged_positions(lib):
    df, dw = oc.flow_case(lib, DEV, 1, 7, 9, 32, "bernoulli", spike=True)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_warp_out_of_range(lib):
    assert oc.warp_case(lib, DEV, 2, 16, 12, 128) < TOL


# ---- bf16x3 conv kernel (conv_x3.hpp): same cases, same fp32-class tolerance as the fp32-MFMA kernel
@pytest.mark.parametrize("tile", [16, 17, 18])
def test_conv_x3_tiles(lib, tile):
    assert oc.conv_x3_case(lib, DEV, 4, 32, 32, 64, 256, 3, 1, 1, True, tile=tile) < TOL
    assert oc.conv_x3_case(lib, DEV, 2, 24, 20, 32, 128, 3, 2, 1, False, tile=tile) < TOL


@pytest.mark.parametrize("k,stride,pad,reflect,cin", [(7, 1, 3, True, 8), (7, 1, 3, True, 32), (1, 1, 0, False, 128), (3, 1, 1, True, 16)])
def test_conv_x3_kinds(lib, k, stride, pad, reflect, cin):
    assert oc.conv_x3_case(lib, DEV, 2, 24, 20, cin, 64, k, stride, pad, reflect) < TOL


def test_conv_x3_patch_kernel(lib):
    """conv_x3p.hpp x3q (tiles 14, 15): LDS-resident input patch, weights in registers; shapes of the residual, fusion and decoder layers"""
    assert oc.conv_x3_case(lib, DEV, 12, 32, 32, 512, 512, 3, 1, 1, True, tile=14) < TOL
    assert oc.conv_x3_case(lib, DEV, 12, 32, 32, 512, 512, 3, 1, 1, True, tile=15) < TOL
    assert oc.conv_x3_case(lib, DEV, 4, 32, 32, 1024, 256, 3, 1, 1, False, tile=14) < 1e-4
    assert oc.conv_x3_case(lib, DEV, 2, 128, 128, 256, 128, 3, 1, 1, False, tile=14) < TOL
    assert oc.conv_x3_case(lib, DEV, 2, 64, 64, 128, 64, 3, 1, 1, False, tile=15) < TOL
    assert oc.conv_x3_case(lib, DEV, 1, 4, 256, 16, 128, 3, 1, 1, True, tile=14, bias=False) < TOL
    assert oc.conv_x3_tiles_bitwise(lib, DEV, 12, 32, 32, 256, 512, (14, 15)) == 0.0


def test_conv_x3r_register_staged(lib):
    """conv_x3r.hpp (tiles 16, 17, 18): stems, stride-2 and 1x1 layers; the tiles of the family are bit-identical"""
    assert oc.conv_x3_case(lib, DEV, 2, 128, 128, 8, 64, 7, 1, 3, True, tile=17) < TOL
    assert oc.conv_x3_case(lib, DEV, 4, 64, 64, 128, 256, 3, 2, 1, False, tile=16) < TOL
    assert oc.conv_x3_case(lib, DEV, 4, 32, 32, 512, 512, 1, 1, 0, False, tile=17) < TOL
    assert oc.conv_x3_case(lib, DEV, 3, 21, 19, 32, 130, 3, 1, 1, True, tile=16, bias=False) < TOL
    assert oc.conv_x3_tiles_bitwise(lib, DEV, 4, 32, 32, 256, 256, (16, 17, 18)) == 0.0


def test_conv_x3_big_layers(lib):
    assert oc.conv_x3_case(lib, DEV, 12, 32, 32, 512, 512, 3, 1, 1, True) < TOL
    assert oc.conv_x3_case(lib, DEV, 2, 16, 16, 1024, 1024, 3, 1, 1, True) < 1e-4
    assert oc.conv_x3_case(lib, DEV, 3, 5, 7, 32, 130, 3, 1, 1, True, bias=False) < TOL


# ---- fp16x2 patch kernel with the producer's IN + ReLU fused into the staging (conv_h2.hpp); error relative to max|fp64 reference|
@pytest.mark.parametrize("norm", [False, True])
def test_conv_h2_layers(lib, norm):
    """the shapes it runs on in the forward: ResnetBlock, FuseNet, decoder up-convolutions (3 products, 64-wide tiles)"""
    assert oc.conv_h2_case(lib, DEV, 12, 32, 32, 512, 512, True, norm=norm) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 4, 32, 32, 1024, 1024, True, norm=norm) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 2, 128, 128, 256, 128, True, norm=norm) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 1, 256, 256, 128, 64, True, norm=norm, bias=False) < 1e-6


def test_conv_h2_variants(lib):
    """zero padding, odd slab counts, 128-wide tiles, four products, operand magnitudes far from 1"""
    assert oc.conv_h2_case(lib, DEV, 3, 8, 64, 48, 96, False, norm=True) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 4, 32, 32, 512, 256, True, norm=True, tile_n=128) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 4, 32, 32, 512, 256, True, norm=True, nprod=4) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 2, 32, 32, 256, 128, True, scale=300.0) < 1e-6
    assert oc.conv_h2_case(lib, DEV, 2, 32, 32, 256, 128, True, scale=1e-4) < 1e-6


def test_conv_h2_tile_widths_bitwise(lib):
    """32-, 64- and 128-wide tiles of conv_h2 run the same chains per output element: the launcher's choice never changes a result"""
    import torch
    ys = []
    for bn in (32, 64, 128):
        ys.append(oc.conv_h2_case(lib, DEV, 4, 32, 32, 256, 256, True, norm=True, tile_n=bn, return_output=True))
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2])


def test_conv_h2r_layers(lib):
    """the encoder's downsampling convolutions and stems at their real shapes"""
    assert oc.conv_h2r_case(lib, DEV, 2, 256, 256, 64, 128, 3, norm=True) < 1e-6
    assert oc.conv_h2r_case(lib, DEV, 4, 64, 64, 256, 512, 3, norm=True) < 1e-6
    assert oc.conv_h2r_case(lib, DEV, 2, 256, 256, 8, 64, 7) < 1e-6
    assert oc.conv_h2r_case(lib, DEV, 1, 256, 256, 32, 64, 7, bias=False) < 1e-6
