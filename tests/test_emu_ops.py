"""CPU tier: every HIP kernel of the hot path executed under the fiber emulator (tests/emu) through
the C ABI and compared with PyTorch CPU ops.  This checks indexing / padding / MFMA fragment layout /
reduction logic; numerics on real hardware are the GPU tier's job."""
import pytest

import op_cases as oc

TOL = 2e-5


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
def test_conv_tiles(emu_lib, tile, monkeypatch):
    """every tile of the fp32 LDS-DMA kernel (conv_dma) and, with the in-loader IN+ReLU, of the register-staged one"""
    monkeypatch.setenv("TSNET_DMA_TILE", str(tile))
    monkeypatch.setenv("TSNET_CONV_TILE", str(tile))
    for norm in (False, True):
        assert oc.conv_case(emu_lib, "cpu", 2, 11, 13, 16, 256, 3, 1, 1, True, norm=norm) < TOL
        assert oc.conv_case(emu_lib, "cpu", 1, 20, 13, 32, 128, 3, 2, 1, False, norm=norm) < TOL
        assert oc.conv_case(emu_lib, "cpu", 1, 9, 9, 8, 128, 7, 1, 3, True, norm=norm) < TOL


@pytest.mark.parametrize("k,stride,pad,reflect", [(7, 1, 3, True), (3, 2, 1, False), (3, 1, 1, True), (1, 1, 0, False)])
@pytest.mark.parametrize("norm", [False, True])
def test_conv_kinds(emu_lib, k, stride, pad, reflect, norm):
    assert oc.conv_case(emu_lib, "cpu", 2, 12, 10, 8, 24, k, stride, pad, reflect, norm=norm) < TOL


def test_conv_head_tanh(emu_lib):
    assert oc.conv_case(emu_lib, "cpu", 1, 8, 8, 64, 3, 7, 1, 3, True, norm=True, act=1) < TOL


def test_conv_ragged_m_and_no_bias(emu_lib):
    assert oc.conv_case(emu_lib, "cpu", 3, 5, 7, 32, 130, 3, 1, 1, True, bias=False) < TOL


@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("reflect", [True, False])
def test_conv_h2(emu_lib, norm, reflect):
    """fp16x2 patch convolution (conv_h2.hpp): slab parity (1, 2 and 3 slabs), zero / reflection padding, fused IN + ReLU, 3 and 4 products"""
    for cin in (16, 32, 48):
        assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, cin, 64, reflect, norm=norm) < 2e-6
    assert oc.conv_h2_case(emu_lib, "cpu", 2, 8, 32, 32, 96, reflect, norm=norm, bias=False, nprod=4) < 2e-6


def test_conv_h2_wide_tile_and_scales(emu_lib):
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 64, 32, 128, True, norm=True, tile_n=128) < 2e-6
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 8, 32, 32, 96, False, norm=True, tile_n=32) < 2e-6           # small-M tiles: four waves over the four patch rows
    # operands far from 1: the power-of-two scales keep the fp16 planes in range (results relative to max|ref|)
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 16, 64, True, scale=300.0) < 2e-6
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 16, 64, True, scale=1e-4) < 2e-6


def test_conv_h2r(emu_lib):
    """conv_h2r: 3x3 / stride-2 (with and without the fused IN + ReLU) and the 7x7 stems (8 channels: two taps per k-group; 32 channels)"""
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 32, 16, 16, 64, 3, norm=True) < 2e-6
    assert oc.conv_h2r_case(emu_lib, "cpu", 2, 16, 32, 32, 128, 3, norm=False, bias=False) < 2e-6
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 16, 8, 64, 7) < 2e-6
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 16, 8, 32, 64, 7) < 2e-6


def test_conv_h2d_stride2_patch_kernel(emu_lib):
    """3x3 / stride-2 layers whose output splits into 4 x 32 rectangles run on the patch kernel (conv_h2.hpp h2d): one, two and four slabs
    (stage parity; the entry takes power-of-two input widths; the kernel takes layers with at least 512 output channels), zero padding at
    all borders, with and without the fused IN + ReLU, two images / two tiles"""
    for cin in (16, 32, 64):
        assert oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 64, cin, 512, 3, norm=True, seed=cin) < 2e-6
    assert oc.conv_h2r_case(emu_lib, "cpu", 2, 16, 64, 32, 512, 3, norm=False, bias=False) < 2e-6
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 128, 16, 576, 3, norm=True) < 2e-6                 # nine N-tiles, two M-tiles


def test_conv_h2s_stem_patch_kernel(emu_lib):
    """8-channel 7x7 stems on whole 4 x 32 rectangles run on the patch kernel (conv_h2.hpp h2s): reflection at all four borders (one tile
    high / several tiles), two images, no bias"""
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 4, 32, 8, 64, 7) < 2e-6
    assert oc.conv_h2r_case(emu_lib, "cpu", 2, 8, 64, 8, 64, 7, bias=False, seed=3) < 2e-6


@pytest.mark.parametrize("C,H,W", [(8, 6, 5), (64, 16, 16), (24, 9, 3), (1024, 2, 2)])
@pytest.mark.parametrize("relu,resid", [(True, False), (False, True)])
def test_instnorm(emu_lib, C, H, W, relu, resid):
    assert oc.instnorm_case(emu_lib, "cpu", 2, H, W, C, relu, resid) < TOL


def test_instnorm_large_mean(emu_lib):
    # mean >> std: E[x^2]-mean^2 in fp32 would cancel; the fp64 partials must not
    assert oc.instnorm_case(emu_lib, "cpu", 1, 16, 16, 8, False, False, offset=300.0) < 2e-3


@pytest.mark.parametrize("norm", [False, True])
def test_upsample(emu_lib, norm):
    assert oc.upsample_case(emu_lib, "cpu", 2, 5, 7, 16, norm) < TOL


@pytest.mark.parametrize("mask_mode", ["bernoulli", "ones", "zeros", "soft"])
def test_flow_masks(emu_lib, mask_mode):
    df, dw = oc.flow_case(emu_lib, "cpu", 2, 4, 6, 64, mask_mode)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_ragged_positions_and_spike(emu_lib):
    # P = 7*9 = 63 (not a multiple of the 32-position tiles); spiky attention forces rescales
    df, dw = oc.flow_case(emu_lib, "cpu", 1, 7, 9, 32, "bernoulli", spike=True)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_many_tiles(emu_lib):
    df, dw = oc.flow_case(emu_lib, "cpu", 1, 16, 16, 16, "ones", spike=True)   # 8 source tiles: 2 per wave
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_warp_out_of_range(emu_lib):
    assert oc.warp_case(emu_lib, "cpu", 2, 5, 6, 16) < TOL


def test_conv_x3_patch_kernel(emu_lib):
    """conv_x3p.hpp x3q (weights fetched into registers, input patch resident in LDS, taps as shifted views): tiles 14 (128 x 128) and
    15 (128 x 64); reflection and zero padding, one and several 16-channel slabs, several tiles per row / per image / per batch.
    (The superseded generations -- LDS-DMA tiles 0..10, x3p 11..13 -- are only in the tools build, tools/x3_ablate.py.)"""
    assert oc.conv_x3_case(emu_lib, "cpu", 2, 8, 32, 16, 128, 3, 1, 1, True, tile=14) < TOL
    assert oc.conv_x3_case(emu_lib, "cpu", 1, 4, 64, 64, 256, 3, 1, 1, False, tile=14) < TOL
    assert oc.conv_x3_case(emu_lib, "cpu", 1, 8, 32, 32, 64, 3, 1, 1, True, tile=15, bias=False) < TOL
    assert oc.conv_x3_case(emu_lib, "cpu", 1, 4, 32, 32, 192, 3, 1, 1, False, tile=15) < TOL
    # same arithmetic per output element whatever the launch shape
    assert oc.conv_x3_tiles_bitwise(emu_lib, "cpu", 3, 4, 32, 32, 256, (14, 15)) == 0.0


def test_conv_x3r_register_staged(emu_lib):
    """conv_x3r.hpp (tiles 16, 17, 18 = 128 / 64 / 32 wide): A tile staged through registers, weights fetched into registers; every
    kind of layer the patch kernels do not take; the tiles of the family are bit-identical"""
    for tile in (16, 17, 18):
        assert oc.conv_x3_case(emu_lib, "cpu", 2, 11, 13, 16, 256, 3, 1, 1, True, tile=tile) < TOL
        assert oc.conv_x3_case(emu_lib, "cpu", 1, 20, 13, 32, 128, 3, 2, 1, False, tile=tile) < TOL
        assert oc.conv_x3_case(emu_lib, "cpu", 1, 9, 9, 8, 128, 7, 1, 3, True, tile=tile) < TOL
        assert oc.conv_x3_case(emu_lib, "cpu", 3, 6, 5, 128, 192, 1, 1, 0, False, tile=tile, bias=False) < TOL
    assert oc.conv_x3_tiles_bitwise(emu_lib, "cpu", 2, 9, 11, 32, 128, (16, 17, 18)) == 0.0


def test_conv_x3_1x1_and_ragged(emu_lib):
    assert oc.conv_x3_case(emu_lib, "cpu", 3, 6, 5, 128, 160, 1, 1, 0, False) < TOL
    assert oc.conv_x3_case(emu_lib, "cpu", 3, 5, 7, 32, 130, 3, 1, 1, True, bias=False) < TOL
