"""CPU tier: every HIP kernel of the hot path executed under the fiber emulator (tests/emu) through
the C ABI and compared with PyTorch CPU ops.  This checks indexing / padding / MFMA fragment layout /
reduction logic; numerics on real hardware are the GPU tier's job."""
import pytest
import torch

import op_cases as oc

TOL = 2e-5
REL = 2e-6     # conv error relative to max|fp64 reference|: fp32-class (two-level accumulation of exact fp16 x 2 products)


@pytest.mark.parametrize("k,stride,pad,reflect", [(7, 1, 3, True), (3, 2, 1, False), (3, 1, 1, True), (1, 1, 0, False)])
@pytest.mark.parametrize("norm", [False, True])
def test_conv_kinds_general_kernel(emu_lib, k, stride, pad, reflect, norm):
    """every layer kind of the forward on frames that do NOT split into 4 x 32 rectangles -> the general implicit GEMM (conv_h2r.hpp):
    ragged image tiles (12 x 10 = 120 positions per image), 8 input channels (two taps per k-group) and 16 / 48 (any multiple of 16)"""
    for cin in ((8, 48) if k != 1 else (16, 48)):
        assert oc.conv_case(emu_lib, "cpu", 2, 12, 10, cin, 24, k, stride, pad, reflect, norm=norm) < REL


def test_conv_general_kernel_tiles_and_shapes(emu_lib):
    """two tiles per image with a ragged second one, 128-wide tiles, more than one N tile, no bias, a transform without ReLU; the two tile
    widths of the general kernel run the same chains (bit-identical)"""
    assert oc.conv_case(emu_lib, "cpu", 2, 13, 11, 32, 130, 3, 1, 1, True, bias=False) < REL
    assert oc.conv_case(emu_lib, "cpu", 1, 20, 13, 32, 128, 3, 2, 1, False, norm=True, relu=False) < REL
    import torch
    a = oc.conv_case(emu_lib, "cpu", 1, 9, 17, 16, 256, 3, 1, 1, True, norm=True, kernel=1, tile=64, return_output=True)
    b = oc.conv_case(emu_lib, "cpu", 1, 9, 17, 16, 256, 3, 1, 1, True, norm=True, kernel=1, tile=128, return_output=True)
    assert torch.equal(a, b)


def test_conv_g64_equals_conv_h2r_bitwise(emu_lib):
    """the 64-deep-step form of the general kernel (round 6) on every layer kind the forward sends it: equal bits, and < REL of the reference"""
    assert oc.conv_g64_cases(emu_lib, "cpu") < REL


def test_conv_h2s32_pose_stem_equals_conv_h2r_bitwise(emu_lib):
    """the pose model's 32-channel 7 x 7 stems on their own patch kernel (round 6): equal bits to the general kernel, < REL of the reference"""
    assert oc.conv_h2s32_cases(emu_lib, "cpu") < REL


def test_conv_cat_on_load(emu_lib):
    """dec.map_conv: 1x1 on cat(pg, sg) formed on load from two tensors (TSNet.py:163); also a 3x3 and a shared second tensor"""
    assert oc.conv_cat_case(emu_lib, "cpu", 2, 4, 8, 32, 32, 48) < REL
    assert oc.conv_cat_case(emu_lib, "cpu", 3, 5, 6, 16, 48, 24, k=3, shared=True) < REL


@pytest.mark.parametrize("C", [8, 16, 32])
def test_head(emu_lib, C):
    """RGB head (head_conv.hpp): the narrow-net form (C = 8: half a 16-channel stage; C = 16) and head_conv3 (C = 32), with the composite"""
    assert oc.head_case(emu_lib, "cpu", 1, 16, 20, C) < TOL
    if C == 16:
        assert oc.head_case(emu_lib, "cpu", 1, 8, 256, C, norm=False, composite=True) < TOL


def test_head_tile_rows_give_the_same_bits(emu_lib):
    """head_conv3 with 8-, 16- and 32-row tiles (the launcher picks by the number of workgroups: 8 rows for one frame, 32 from B = 4 on):
    a pixel's sums do not depend on its tile -- torch.equal; a ragged frame (40 rows: tiles hanging over the edge in every form), the
    fused InstanceNorm + ReLU, the composite"""
    for (N, H, W, C, norm, comp) in ((1, 40, 36, 16, True, False), (2, 32, 256, 16, False, True)):
        ys = [oc.head_case(emu_lib, "cpu", N, H, W, C, norm=norm, composite=comp, rows=r, return_output=True) for r in (8, 16, 32, 0)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2]) and torch.equal(ys[2], ys[3])
        assert oc.head_case(emu_lib, "cpu", N, H, W, C, norm=norm, composite=comp, rows=32) < TOL


@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("reflect", [True, False])
def test_conv_h2(emu_lib, norm, reflect):
    """fp16x2 patch convolution (conv_h2.hpp): slab parity (1, 2 and 3 slabs), zero / reflection padding, fused IN + ReLU, 3 and 4 products"""
    for cin in (16, 32, 48):
        assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, cin, 64, reflect, norm=norm) < REL
    assert oc.conv_h2_case(emu_lib, "cpu", 2, 8, 32, 32, 96, reflect, norm=norm, bias=False, nprod=4, tile_n=64) < REL


def test_conv_h2_tile_shapes_and_scales(emu_lib):
    import torch
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 64, 32, 128, True, norm=True, tile_n=128) < REL
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 8, 32, 32, 96, False, norm=True, tile_n=32) < REL           # small-M tiles: four waves over the four patch rows
    assert oc.conv_h2_case(emu_lib, "cpu", 2, 6, 64, 32, 128, False, norm=True, tile_n=2128) < REL        # 2-row tiles (6 rows: not a multiple of 4)
    # every tile shape runs the same chains per output element
    ys = [oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 48, 128, True, norm=True, tile_n=t, return_output=True) for t in (32, 64, 128, 2128)]
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    # two K groups (20032 / 20064: single-frame launches; eight waves, group g folds the chains of slabs g, g + 2, ...; total = P0 + P1, weight
    # fragments eight steps ahead): the same bits in both tile
    # shapes, fp32-class accuracy, NOT the bits of the one-group tiles (another association of the same chains)
    zs = [oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 64, 128, True, norm=True, tile_n=t, return_output=True) for t in (64, 20032, 20064)]
    assert torch.equal(zs[1], zs[2]) and not torch.equal(zs[0], zs[1]) and (zs[0] - zs[1]).abs().max().item() < 1e-5 * zs[0].abs().max().item()
    assert oc.conv_h2_case(emu_lib, "cpu", 2, 8, 32, 32, 64, False, tile_n=20064) < REL
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 64, 96, 96, True, norm=True, tile_n=20032) < REL       # six slabs: three per group (odd local count)
    # operands far from 1: the power-of-two scales keep the fp16 planes in range (results relative to max|ref|)
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 16, 64, True, scale=300.0) < REL
    assert oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 16, 64, True, scale=1e-4) < REL


def test_conv_h2d_stride2_patch_kernel(emu_lib):
    """3x3 / stride-2 layers whose output splits into 4 x 32 rectangles run on the patch kernel (conv_h2.hpp h2d): one, two and four slabs
    (stage parity), zero padding at all borders, with and without the fused IN + ReLU, two images / two tiles, both workgroup shapes
    (four waves x 64 columns, eight waves x 128 columns, four waves x two rows x 128 columns) -- bit-identical to each other"""
    import torch
    for cin in (16, 32, 64):
        assert oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 64, cin, 128, 3, norm=True, seed=cin, kernel=2) < REL
    assert oc.conv_h2r_case(emu_lib, "cpu", 2, 16, 64, 32, 64, 3, norm=False, bias=False, kernel=2) < REL
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 128, 16, 320, 3, norm=True, kernel=2, tile=64) < REL      # five N-tiles, two M-tiles
    a = oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 64, 32, 256, 3, norm=True, kernel=2, tile=64, return_output=True)
    b = oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 64, 32, 256, 3, norm=True, kernel=2, tile=128, return_output=True)
    c = oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 64, 32, 256, 3, norm=True, kernel=2, tile=2128, return_output=True)       # two rows x 128, four waves
    assert torch.equal(a, b) and torch.equal(a, c)
    assert oc.conv_h2r_case(emu_lib, "cpu", 2, 12, 64, 48, 128, 3, norm=True, kernel=2, tile=2128) < REL     # six output rows: three 2-row tiles per image
    # the deep schedule of the 2 x 128 tile (12128: weights eight steps ahead, the three staging rounds in flight together -- what a launch of
    # at most two workgroups per CU runs, so every emulator-sized launch when the choice is the launcher's): the same bits, odd and even slab
    # counts, with and without the fused transform
    d = oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 64, 32, 256, 3, norm=True, kernel=2, tile=12128, return_output=True)
    assert torch.equal(a, d)
    for (cin, norm) in ((16, True), (48, False), (64, True)):
        ys = [oc.conv_h2r_case(emu_lib, "cpu", 2, 16, 64, cin, 128, 3, norm=norm, kernel=2, tile=t, seed=cin, return_output=True) for t in (2128, 12128, 0)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2])


def test_conv_h2s_stem_patch_kernel(emu_lib):
    """8-channel 7x7 stems on whole 4 x 32 rectangles run on the patch kernel (conv_h2.hpp h2s): reflection at all four borders (one tile
    high / several tiles), two images, no bias; the general kernel on the same layer agrees to rounding (different K order)"""
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 4, 32, 8, 64, 7, kernel=2) < REL
    assert oc.conv_h2r_case(emu_lib, "cpu", 2, 8, 64, 8, 64, 7, bias=False, seed=3, kernel=2) < REL
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 4, 32, 8, 64, 7, kernel=1) < REL
    assert oc.conv_h2r_case(emu_lib, "cpu", 1, 8, 32, 32, 64, 7, kernel=0) < REL                    # 32-channel stem (pose model): general kernel


def test_bf16_operand_convs(emu_lib):
    """bf16-operand mode of every kernel: one plane, one product; error of bf16 rounding (2^-9 per operand)"""
    assert oc.conv_case(emu_lib, "cpu", 1, 4, 32, 32, 64, 3, 1, 1, True, norm=True, nprod=1) < 2e-2
    assert oc.conv_case(emu_lib, "cpu", 1, 8, 64, 16, 128, 3, 2, 1, False, norm=True, nprod=1) < 2e-2
    assert oc.conv_case(emu_lib, "cpu", 1, 4, 32, 8, 64, 7, 1, 3, True, nprod=1) < 2e-2
    assert oc.conv_case(emu_lib, "cpu", 1, 5, 7, 16, 24, 1, 1, 0, False, nprod=1) < 2e-2


def test_bf16_wide_tile_side_by_side_waves(emu_lib):
    """the bf16 4 x 128 patch tile with its four waves side by side (tile code 3128; the bf16 layer's own tile since round 6): the A rows
    double-buffered by tap column, every weight fragment loaded once -- same K order and chains, so THE SAME BITS as the 2 x 2 wave grid (128)
    and as the 64-wide tile; reflection and zero padding, one and several tiles, two slab counts, with and without the fused transform"""
    for (N, H, W, Ci, Co, refl, norm) in ((1, 4, 32, 32, 128, True, True), (2, 8, 64, 48, 256, False, True), (1, 8, 32, 16, 128, True, False)):
        ys = [oc.conv_h2_case(emu_lib, "cpu", N, H, W, Ci, Co, refl, norm=norm, nprod=1, tile_n=t, return_output=True) for t in (64, 128, 3128, 0)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2]) and torch.equal(ys[2], ys[3])
        assert oc.conv_h2_case(emu_lib, "cpu", N, H, W, Ci, Co, refl, norm=norm, nprod=1, tile_n=3128) < 2e-2
    with pytest.raises(AssertionError, match="bf16 4 x 128"):
        oc.conv_h2_case(emu_lib, "cpu", 1, 4, 32, 32, 128, True, nprod=3, tile_n=3128)


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
    for (h, w) in ((1, 3), (3, 1), (1, 1), (2, 2)):          # one-row / one-column maps: every clamp of the 3 x 3 window (round 6: a thread per 2 x 2 output block)
        assert oc.upsample_case(emu_lib, "cpu", 1, h, w, 8, norm, seed=h * 10 + w) < TOL


@pytest.mark.parametrize("mask_mode", ["bernoulli", "ones", "zeros", "soft"])
def test_flow_masks(emu_lib, mask_mode):
    df, dw = oc.flow_case(emu_lib, "cpu", 2, 4, 6, 64, mask_mode)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_ragged_positions_and_spike(emu_lib):
    # P = 7*9 = 63 (not a multiple of the 32-position tiles); spiky attention forces rescales
    df, dw = oc.flow_case(emu_lib, "cpu", 1, 7, 9, 32, "bernoulli", spike=True)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_k_small_map_and_argument_errors(emu_lib):
    """tsnet_op_flow_k on a small map (flow_kernel: one workgroup per (source, batch element, 64 targets)) with K = 3 sources; bad arguments
    are refused with a message, not run."""
    import torch
    assert oc.flow_k_case(emu_lib, "cpu", 2, 3, 6, 8, 16, "bernoulli") < 5e-5
    z = torch.zeros(8 * 6 * 8 * 16)
    for K, variant, word in ((9, 0, "sources"), (0, 0, "sources"), (1, 3, "variant")):
        rc = emu_lib.tsnet_op_flow_k(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, K, 6, 8, 16, 48, 64, z.data_ptr(), variant, 1, None, None)
        assert rc != 0 and word in emu_lib.tsnet_op_last_error().decode()
    assert emu_lib.tsnet_flow_plan(0, 6, 8, 16) == -1 and emu_lib.tsnet_flow_plan(1, 6, 8, 12) == -1 and emu_lib.tsnet_flow_plan(1, 6, 8, 16) == 0


def test_flow_large_map_persistent_target_tiles(emu_lib):
    """flow_kernel_p (maps of >= 2048 positions): 32 x 64 positions, one batch element -> 32 target tiles x G = 4 workgroups, every wave one
    source pair per slice; K = 2 sources swept with the target tile kept; partial states merged by the last workgroup to arrive.  Second
    case: four batch elements (G = 2, two pairs per wave and slice) with soft masks."""
    assert emu_lib.tsnet_flow_plan(1, 32, 64, 16) == 4 and emu_lib.tsnet_flow_plan(4, 64, 32, 8) == 2
    assert emu_lib.tsnet_flow_plan(4, 32, 32, 512) == 0            # configs[1]: flow_kernel fills the chip (192 workgroups, 2 pairs per wave)
    assert emu_lib.tsnet_flow_plan(1, 64, 64, 512) == 4            # configs[4] shard: 64 target tiles x 4 = one workgroup per CU
    assert emu_lib.tsnet_flow_plan(4, 64, 64, 512) == 1            # the target tiles alone fill the chip: no split, no cross-workgroup merge
    assert emu_lib.tsnet_flow_plan(1, 64, 64, 1024) == 0 and emu_lib.tsnet_flow_plan(1, 60, 64, 512) == 0   # LDS; ragged
    assert oc.flow_k_case(emu_lib, "cpu", 1, 2, 32, 64, 16, "bernoulli") < 5e-5
    assert oc.flow_k_case(emu_lib, "cpu", 4, 1, 64, 32, 8, "soft") < 5e-5
    # the same frame alone (G = 4), in a batch of four (G = 2) and of eight (G = 1, no cross-workgroup merge): the same bits
    import torch
    f1, f4, f8 = oc.flow_k_batch_independence_case(emu_lib, "cpu", (1, 4, 8), 2, 32, 64, 8)
    assert emu_lib.tsnet_flow_plan(8, 32, 64, 8) == 1 and torch.equal(f1, f4) and torch.equal(f1, f8)


def test_flow_many_tiles(emu_lib):
    df, dw = oc.flow_case(emu_lib, "cpu", 1, 16, 16, 16, "ones", spike=True)   # 4 pairs of 32-source blocks: waves 0..3 sweep one each
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient
    df, dw = oc.flow_case(emu_lib, "cpu", 1, 24, 24, 16, "bernoulli", spike=True)   # 9 pairs: wave 0 sweeps two; 9 workgroups of 64 targets
    assert df < 5e-5 and dw < 4e-3
    df, dw = oc.flow_case(emu_lib, "cpu", 1, 8, 10, 1024, "bernoulli", spike=True)  # 1024 channels: 64 targets' planes exceed the LDS -> 32 per workgroup (flow_kernel<1>)
    assert df < 5e-5 and dw < 4e-3


def test_warp_out_of_range(emu_lib):
    assert oc.warp_case(emu_lib, "cpu", 2, 5, 6, 16) < TOL


def test_conv_split_worst_case_dynamic_range(emu_lib):
    """The fp16 x 2 split under an adversarial dynamic range inside ONE image (1 % of the activations at amax, the bulk 2^-12 .. 2^-24
    below; weights likewise): the absolute error stays within 3 x that of the exact-fp32 chain (torch fp32 conv) on the same data, and
    outputs built from the tiny tiers alone keep >= 19 bits relative to their own magnitude (the split's floor is 2^-40 amax per element:
    DESIGN.md section 4.1).  GPU tier: tests/test_gpu_ops.py at the ResnetBlock shape."""
    for corner in (False, True):
        for wt in (True, False):
            e, e32, eq, rq, _, _ = oc.conv_split_worstcase_case(emu_lib, "cpu", 1, 16, 32, 128, 64, corner=corner, weights_too=wt)
            assert e <= 3.0 * e32, (corner, wt, e, e32)
            assert eq <= rq * 2.0 ** -19, (corner, wt, eq, rq)
    e, e32, _, _, _, _ = oc.conv_split_worstcase_case(emu_lib, "cpu", 1, 16, 32, 128, 64, tiers=(-6, -9, -12))
    assert e <= 3.0 * e32


def test_conv_w1_chunks_of_tiles_bitwise(emu_lib):
    """A conv_w1 workgroup runs a chunk of 1, 2 or 3 consecutive spatial tiles of one channel tile, the next tile's V(0), V(1) produced under the current tile's last two
    periods and the epilogue of a tile with a successor confined to one stage (two exchange passes).  Same chains per output element: the
    chunk size never changes a bit.  Cases: chunks that cross from one image into the next under a fused InstanceNorm (the second transform
    table), zero padding (the per-tile padding masks), a raw input, an odd slab count (the all-zero last slab), a second channel tile."""
    import torch
    for args, kw, chunks in (((6, 16, 32, 128, 128, True), dict(norm=True), (1, 3)), ((4, 16, 32, 128, 128, False), dict(norm=True), (1, 2)),
                             ((6, 16, 32, 112, 128, True), dict(), (1, 3)), ((4, 16, 32, 128, 64, False), dict(norm=True, relu=False), (1, 2))):
        ys = [oc.conv_w1_case(emu_lib, "cpu", *args, chunk=c, return_output=True, **kw) for c in chunks]
        assert all(torch.equal(ys[0], y) for y in ys[1:]), (args, kw)
        assert oc.conv_w1_case(emu_lib, "cpu", *args, chunk=chunks[-1], **kw) < REL


def test_conv_w1_odd_slab_count_stages_zeros_past_cin(emu_lib):
    """an odd slab count (Cin = 48, 80) under a fused InstanceNorm whose first channels carry a large gain -- the staging past
    Cin must be zeros, not another pixel's channels under the wrong affine (inf x 0 = NaN in the MFMA)."""
    assert oc.conv_w1_odd_slab_case(emu_lib, "cpu", 1, 4, 32, 48, 64) < REL
    assert oc.conv_w1_odd_slab_case(emu_lib, "cpu", 2, 8, 32, 80, 64, seed=3) < REL


def test_conv_w1_worst_case_range_and_structured_filters(emu_lib):
    """The kernel that runs 83 % of the forward's FLOPs (conv_w1, tsnet_op_conv2d(kernel = 3)) under the adversarial dynamic range of the test
    above -- it has one bit less operand head-room than the direct kernel and its output transform subtracts -- and on STRUCTURED filters
    (smooth g0 = g2 with |g1| >> |g0|, binomial, Sobel-like antisymmetric, second difference: what a trained checkpoint holds, where the
    Winograd filters U1 / U2 differ most from the taps).  Same tiers and gates as the direct kernel: <= 3 x the exact-fp32 chain, >= 19 bits
    on quiet outputs.  GPU tier: tests/test_gpu_ops.py at 512 -> 512."""
    for corner in (False, True):
        for wt in (True, False):
            e, e32, eq, rq, _, _ = oc.conv_split_worstcase_case(emu_lib, "cpu", 1, 16, 32, 128, 64, corner=corner, weights_too=wt, kernel=3)
            assert e <= 3.0 * e32, (corner, wt, e, e32)
            assert eq <= rq * 2.0 ** -19, (corner, wt, eq, rq)
    e, e32, _, _, _, _ = oc.conv_split_worstcase_case(emu_lib, "cpu", 1, 16, 32, 128, 64, tiers=(-6, -9, -12), kernel=3)
    assert e <= 3.0 * e32
    for kind in ("smooth", "binomial", "sobel", "edge"):
        for norm in (True, False):
            e, e32, rmax = oc.conv_structured_filter_case(emu_lib, "cpu", 1, 8, 32, 128, 64, kind, kernel=3, norm=norm)
            assert e <= 3.0 * e32 and e <= REL * rmax, (kind, norm, e, e32, rmax)


def test_conv_w1_winograd_x_form(emu_lib):
    """3x3 / stride 1 / pad 1 in the Winograd F(2,3)-along-x form (conv_w1.hpp, tsnet_op_conv2d(kernel = 3)): reflection and zero padding,
    with and without the fused IN + ReLU, one / three / five slabs (the odd counts end in an all-zero slab of the last period), two images
    and several tiles, a second 64-channel tile column, bf16 operands.  Accuracy: the direct kernel's class (the transform precedes the
    split; tools/probes/winograd_probe.py).  The forward runs its ResnetBlock / FuseNet / first up-convolution layers in this form."""
    assert oc.conv_w1_case(emu_lib, "cpu", 1, 4, 32, 16, 64, True) < REL
    assert oc.conv_w1_case(emu_lib, "cpu", 2, 8, 64, 48, 96, False, norm=True) < REL
    assert oc.conv_w1_case(emu_lib, "cpu", 1, 4, 32, 80, 64, True, norm=True, relu=False) < REL
    assert oc.conv_w1_case(emu_lib, "cpu", 1, 8, 32, 64, 128, False, bias=False) < REL
    assert oc.conv_w1_case(emu_lib, "cpu", 1, 4, 32, 32, 64, True, scale=300.0) < REL
    assert oc.conv_w1_case(emu_lib, "cpu", 1, 4, 32, 32, 64, True, norm=True, nprod=1) < 2e-2
