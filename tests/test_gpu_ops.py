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


REL = 1e-6     # conv error relative to max|fp64 reference| (fp32-class: exact fp16 x 2 products, two-level fp32 accumulation)


# ---- the general implicit-GEMM kernel (conv_h2r.hpp): layers / frame sizes without a patch kernel
@pytest.mark.parametrize("k,stride,pad,reflect", [(7, 1, 3, True), (3, 2, 1, False), (3, 1, 1, True), (1, 1, 0, False)])
@pytest.mark.parametrize("norm", [False, True])
def test_conv_kinds_general_kernel(lib, k, stride, pad, reflect, norm):
    for cin in ((8, 48) if k != 1 else (16, 48)):
        assert oc.conv_case(lib, DEV, 2, 24, 20, cin, 24, k, stride, pad, reflect, norm=norm) < REL


def test_conv_general_kernel_shapes(lib):
    """ragged image tiles, 64- and 128-wide tiles (bit-identical), the 1x1 layers at their real shape, the pose model's 32-channel stem,
    a feature-resolution 3x3 layer of a 64 x 64 frame (8 x 8 = half a tile per image)"""
    import torch
    assert oc.conv_case(lib, DEV, 3, 21, 19, 32, 130, 3, 1, 1, True, bias=False) < REL
    assert oc.conv_case(lib, DEV, 4, 32, 32, 1024, 512, 1, 1, 0, False) < REL
    assert oc.conv_case(lib, DEV, 1, 256, 256, 32, 64, 7, 1, 3, True, bias=False) < REL
    assert oc.conv_case(lib, DEV, 6, 8, 8, 512, 512, 3, 1, 1, True, norm=True) < REL
    a = oc.conv_case(lib, DEV, 2, 24, 20, 64, 256, 3, 1, 1, True, norm=True, kernel=1, tile=64, return_output=True)
    b = oc.conv_case(lib, DEV, 2, 24, 20, 64, 256, 3, 1, 1, True, norm=True, kernel=1, tile=128, return_output=True)
    assert torch.equal(a, b)


def test_conv_g64_equals_conv_h2r_bitwise(lib):
    """the 64-deep-step form of the general kernel (csrc/conv_g64.hpp): equal bits to conv_h2r on the small cases and on the forward's own
    shapes (1 x 1 at 4096 x 1024 -> 512, the 64 -> 128 stride-2 layer at 256^2, a bf16 stride-2 layer), < REL of the fp64 reference"""
    assert oc.conv_g64_cases(lib, DEV, big=True) < REL


def test_conv_h2s32_pose_stem_equals_conv_h2r_bitwise(lib):
    """the pose model's 32-channel 7 x 7 stems on their own patch kernel (csrc/conv_h2s32.hpp): equal bits to conv_h2r, also at 256 x 256"""
    assert oc.conv_h2s32_cases(lib, DEV, big=True) < REL


def test_conv_cat_on_load(lib):
    """dec.map_conv at its real shape: 1x1 on cat(pg, sg), 512 + 512 -> 512 channels, formed on load"""
    assert oc.conv_cat_case(lib, DEV, 4, 32, 32, 512, 512, 512) < REL
    assert oc.conv_cat_case(lib, DEV, 3, 5, 6, 16, 48, 24, k=3, shared=True) < REL


@pytest.mark.parametrize("C,H,W,composite", [(64, 256, 256, True), (64, 40, 72, False), (8, 32, 32, False), (16, 32, 48, False)])
def test_head(lib, C, H, W, composite):
    assert oc.head_case(lib, DEV, 2, H, W, C, composite=composite) < TOL


def test_head_tile_rows_give_the_same_bits(lib):
    """head_conv3 with 8-, 16- and 32-row tiles at the forward's shape (one frame runs 8 rows, two 16, four and more 32) and on a ragged
    frame: torch.equal"""
    for (N, H, W, C, comp) in ((1, 256, 256, 64, True), (2, 40, 72, 64, False), (4, 256, 256, 64, False)):
        ys = [oc.head_case(lib, DEV, N, H, W, C, composite=comp, rows=r, return_output=True) for r in (8, 16, 32, 0)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2]) and torch.equal(ys[2], ys[3])
        assert oc.head_case(lib, DEV, N, H, W, C, composite=comp, rows=8) < TOL


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


def test_flow_configs4_form_persistent_target_tiles(lib):
    """BASELINE.json configs[4] per GPU: one driving frame, five sources, 64 x 64 positions, 512 channels -> flow_kernel_p (64 target tiles
    x G = 4 workgroups, one per CU; partial states merged across workgroups); then four driving frames (G = 1, no cross-workgroup merge)
    and two at 32 x 64 positions (G = 4, one source pair per wave and slice).  Each case runs twice and must give the same bits: the
    packed-arithmetic build of this kernel did not (csrc/flow_persist.hpp)."""
    assert lib.tsnet_flow_plan(1, 64, 64, 512) == 4 and lib.tsnet_flow_plan(4, 64, 64, 512) == 1 and lib.tsnet_flow_plan(2, 32, 64, 512) == 4
    assert oc.flow_k_case(lib, DEV, 1, 5, 64, 64, 512) < 5e-5
    assert oc.flow_k_case(lib, DEV, 4, 2, 64, 64, 512, "soft") < 5e-5
    assert oc.flow_k_case(lib, DEV, 2, 3, 32, 64, 512) < 5e-5


def test_flow_large_map_bits_do_not_depend_on_the_batch(lib):
    """flow_kernel_p at the configs[4] map (64 x 64 positions, 512 channels): a frame alone (G = 4 workgroups per target tile), in a batch of
    two (G = 2) and of four (G = 1): the slice tree is a function of the map alone, so the flows are the same bits."""
    import torch
    assert [lib.tsnet_flow_plan(b, 64, 64, 512) for b in (1, 2, 4)] == [4, 2, 1]
    f1, f2, f4 = oc.flow_k_batch_independence_case(lib, DEV, (1, 2, 4), 3, 64, 64, 512)
    assert torch.equal(f1, f2) and torch.equal(f1, f4)


def test_flow_ragged_positions(lib):
    df, dw = oc.flow_case(lib, DEV, 1, 7, 9, 32, "bernoulli", spike=True)
    assert df < 5e-5 and dw < 4e-3   # dw = flow error x feature gradient


def test_flow_wide_features_and_large_maps(lib):
    """1024 channels (64 targets' planes exceed the LDS: 32 targets per workgroup, flow_kernel<1>) and BASELINE.json configs[4]'s 64 x 64 positions"""
    df, dw = oc.flow_case(lib, DEV, 1, 16, 16, 1024, "bernoulli", spike=True)
    assert df < 5e-5 and dw < 4e-3
    df, dw = oc.flow_case(lib, DEV, 1, 64, 64, 512, "box")
    assert df < 5e-5 and dw < 4e-3


def test_warp_out_of_range(lib):
    assert oc.warp_case(lib, DEV, 2, 16, 12, 128) < TOL


# ---- fp16x2 patch kernel with the producer's IN + ReLU fused into the staging (conv_h2.hpp); error relative to max|fp64 reference|
@pytest.mark.parametrize("norm", [False, True])
def test_conv_h2_layers(lib, norm):
    """the shapes it runs on in the forward: ResnetBlock, FuseNet, decoder up-convolutions (3 products, 64-wide tiles)"""
    assert oc.conv_h2_case(lib, DEV, 12, 32, 32, 512, 512, True, norm=norm) < REL
    assert oc.conv_h2_case(lib, DEV, 4, 32, 32, 1024, 1024, True, norm=norm) < REL
    assert oc.conv_h2_case(lib, DEV, 2, 128, 128, 256, 128, True, norm=norm) < REL
    assert oc.conv_h2_case(lib, DEV, 1, 256, 256, 128, 64, True, norm=norm, bias=False) < REL


def test_conv_h2_variants(lib):
    """zero padding, odd slab counts, 128-wide tiles, four products, operand magnitudes far from 1"""
    assert oc.conv_h2_case(lib, DEV, 3, 8, 64, 48, 96, False, norm=True) < REL
    assert oc.conv_h2_case(lib, DEV, 4, 32, 32, 512, 256, True, norm=True, tile_n=128) < REL
    assert oc.conv_h2_case(lib, DEV, 4, 32, 32, 512, 256, True, norm=True, nprod=4, tile_n=64) < REL
    assert oc.conv_h2_case(lib, DEV, 2, 32, 32, 256, 128, True, scale=300.0) < REL
    assert oc.conv_h2_case(lib, DEV, 2, 32, 32, 256, 128, True, scale=1e-4) < REL
    assert oc.conv_h2_case(lib, DEV, 2, 6, 64, 32, 128, False, norm=True, tile_n=2128) < REL


def test_conv_h2_tile_shapes_bitwise(lib):
    """4 x 32, 4 x 64, 4 x 128 and 2 x 128 tiles of conv_h2 run the same chains per output element: the launcher's choice never changes a result"""
    import torch
    ys = [oc.conv_h2_case(lib, DEV, 4, 32, 32, 256, 256, True, norm=True, tile_n=t, return_output=True) for t in (32, 64, 128, 2128)]
    assert all(torch.equal(ys[0], y) for y in ys[1:])


def test_conv_h2_two_k_groups(lib):
    """single-frame launches (at most one 64-wide tile per CU) run the tile as eight waves: two K groups, each folding the chains of every other
    slab, total = P0 + P1.  Both tile shapes give the same bits; against the one-group tiles the association differs (agreement to fp32
    rounding, asserted as such); accuracy stays fp32-class.  The operator's default (tile = 0) is a one-group tile; the ENGINE picks the
    two-group form when it runs a forward of one frame (test_gpu_forward.py::test_single_frame_forward)."""
    import torch
    assert oc.conv_h2_case(lib, DEV, 3, 32, 32, 512, 512, True, norm=True, tile_n=20064) < REL
    assert oc.conv_h2_case(lib, DEV, 1, 32, 32, 512, 512, True, norm=True, tile_n=20032) < REL
    assert oc.conv_h2_case(lib, DEV, 1, 64, 64, 512, 256, True, tile_n=20032) < REL
    zs = [oc.conv_h2_case(lib, DEV, 1, 32, 32, 512, 512, True, norm=True, tile_n=t, return_output=True) for t in (64, 20032, 20064, 0)]
    assert torch.equal(zs[1], zs[2]) and torch.equal(zs[0], zs[3]) and not torch.equal(zs[0], zs[1])
    assert (zs[0] - zs[1]).abs().max().item() < 1e-5 * zs[0].abs().max().item()


def test_conv_h2d_downsampling_layers(lib):
    """the encoder's three stride-2 layers at their real shapes on the patch kernel (conv_h2.hpp h2d), explicitly selected (kernel = 2): the
    four-wave / 64-column, the eight-wave / 128-column and the four-wave / two-row x 128-column workgroups, which are bit-identical to each other.  Against the general kernel on
    the same layer only agreement to rounding is expected -- it visits the K chunks tap-major, the patch kernel slab-major."""
    import torch
    assert oc.conv_h2r_case(lib, DEV, 2, 256, 256, 64, 128, 3, norm=True, kernel=2) < REL
    assert oc.conv_h2r_case(lib, DEV, 4, 128, 128, 128, 256, 3, norm=True, kernel=2) < REL
    assert oc.conv_h2r_case(lib, DEV, 4, 64, 64, 256, 512, 3, norm=True, kernel=2) < REL
    assert oc.conv_h2r_case(lib, DEV, 4, 64, 64, 256, 512, 3, norm=True, kernel=2, tile=64) < REL
    a = oc.conv_h2r_case(lib, DEV, 2, 64, 64, 256, 512, 3, norm=True, kernel=2, tile=64, return_output=True)
    b = oc.conv_h2r_case(lib, DEV, 2, 64, 64, 256, 512, 3, norm=True, kernel=2, tile=128, return_output=True)
    c = oc.conv_h2r_case(lib, DEV, 2, 64, 64, 256, 512, 3, norm=True, kernel=2, tile=2128, return_output=True)      # two rows x 128: the forward's shape
    d = oc.conv_h2r_case(lib, DEV, 2, 64, 64, 256, 512, 3, norm=True, kernel=2, return_output=True)                 # the launcher's own choice
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    # the deep schedule (12128: a single frame's launches) against the plain one, at the one-frame and the batch shapes of both layers
    for (n, hw_, ci, co) in ((1, 64, 256, 512), (3, 128, 128, 256), (4, 64, 256, 512)):
        ys = [oc.conv_h2r_case(lib, DEV, n, hw_, hw_, ci, co, 3, norm=True, kernel=2, tile=t, return_output=True) for t in (2128, 12128, 0)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2])
    assert oc.conv_h2r_case(lib, DEV, 4, 128, 128, 128, 256, 3, norm=True, kernel=2, tile=2128) < REL
    g = oc.conv_h2r_case(lib, DEV, 2, 64, 64, 256, 512, 3, norm=True, kernel=1, return_output=True)
    assert not torch.equal(a, g) and (a - g).abs().max().item() < 1e-4 * a.abs().max().item()
    assert oc.conv_h2r_case(lib, DEV, 2, 64, 64, 256, 512, 3, norm=True, kernel=1) < REL


def test_conv_h2s_stem_layers(lib):
    """the 8-channel 7x7 stems at their real shape on the patch kernel (conv_h2.hpp h2s), explicitly selected; the general kernel on the
    same layer agrees to rounding"""
    assert oc.conv_h2r_case(lib, DEV, 2, 256, 256, 8, 64, 7, kernel=2) < REL
    assert oc.conv_h2r_case(lib, DEV, 1, 256, 256, 8, 64, 7, bias=False, kernel=2, seed=5) < REL
    assert oc.conv_h2r_case(lib, DEV, 1, 256, 256, 8, 64, 7, kernel=1) < REL


def test_bf16_operand_convs(lib):
    assert oc.conv_case(lib, DEV, 4, 32, 32, 512, 512, 3, 1, 1, True, norm=True, nprod=1) < 2e-2
    assert oc.conv_case(lib, DEV, 2, 64, 64, 256, 512, 3, 2, 1, False, norm=True, nprod=1) < 2e-2
    assert oc.conv_case(lib, DEV, 1, 256, 256, 8, 64, 7, 1, 3, True, nprod=1) < 2e-2
    assert oc.conv_case(lib, DEV, 2, 32, 32, 1024, 512, 1, 1, 0, False, nprod=1) < 2e-2


def test_bf16_wide_tile_side_by_side_waves(lib):
    """bf16 4 x 128 patch tile, four waves side by side (3128: the bf16 layers' own tile) against the 2 x 2 wave grid (128) and the 64-wide
    tile: the same bits, at the ResnetBlock and first up-convolution shapes"""
    for (N, H, W, Ci, Co, refl) in ((4, 32, 32, 512, 512, True), (2, 128, 128, 256, 128, True), (3, 32, 32, 1024, 1024, False)):
        ys = [oc.conv_h2_case(lib, DEV, N, H, W, Ci, Co, refl, norm=True, nprod=1, tile_n=t, return_output=True) for t in (64, 128, 3128, 0)]
        assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2]) and torch.equal(ys[2], ys[3])
        assert oc.conv_h2_case(lib, DEV, N, H, W, Ci, Co, refl, norm=True, nprod=1, tile_n=3128) < 2e-2


def test_conv_split_worst_case_dynamic_range(lib):
    """The fp16 x 2 split under an adversarial dynamic range inside ONE image at the ResnetBlock shape (512 -> 512, 32 x 32): 1 % of the
    activations at amax, the bulk at amax * 2^-12 / 2^-18 / 2^-24, weights likewise.  Absolute error within 3 x the exact-fp32 chain's
    (torch fp32 conv on the same data); outputs that see only the tiny tiers keep >= 19 bits relative to their own magnitude."""
    for corner in (False, True):
        for wt in (True, False):
            e, e32, eq, rq, _, _ = oc.conv_split_worstcase_case(lib, DEV, 2, 32, 32, 512, 512, corner=corner, weights_too=wt)
            print(f"worst-case split corner={corner} tiered_weights={wt}: max|err| {e:.3e} vs fp32 chain {e32:.3e} (x{e / e32:.2f}); quiet outputs {eq:.3e} of {rq:.3e}")
            assert e <= 3.0 * e32, (corner, wt, e, e32)
            assert eq <= rq * 2.0 ** -19, (corner, wt, eq, rq)
    e, e32, _, _, _, _ = oc.conv_split_worstcase_case(lib, DEV, 2, 32, 32, 512, 512, tiers=(-6, -9, -12))
    assert e <= 3.0 * e32


def test_conv_w1_chunks_of_tiles_bitwise(lib):
    """conv_w1 workgroups run chunks of 1, 2 or 3 consecutive tiles (the next tile's V(0), V(1) produced under the current tile's last two
    periods; csrc/conv_w1.hpp): at the ResnetBlock, FuseNet and first-up-convolution shapes the chunk size never changes a bit, with the fused
    InstanceNorm (chunks of three cross image boundaries: the second transform table), a raw input, zero padding.  chunk = 0 is the launcher's choice."""
    import torch
    for args, kw, chunks in (((12, 32, 32, 512, 512, True), dict(norm=True), (1, 2, 3, 0)), ((12, 32, 32, 512, 512, True), dict(), (1, 3, 0)),
                             ((3, 32, 32, 1024, 1024, True), dict(norm=True), (1, 2, 0)), ((4, 64, 64, 512, 256, False), dict(norm=True, relu=False), (1, 2, 0)),
                             ((3, 32, 32, 512, 512, False), dict(norm=True), (1, 3, 0))):
        ys = [oc.conv_w1_case(lib, DEV, *args, chunk=c, return_output=True, **kw) for c in chunks]
        assert all(torch.equal(ys[0], y) for y in ys[1:]), (args, kw)
    assert oc.conv_w1_case(lib, DEV, 12, 32, 32, 512, 512, True, norm=True, chunk=3) < REL


def test_conv_w1_odd_slab_count_stages_zeros_past_cin(lib):
    """odd slab counts under a fused InstanceNorm with a large gain on the first channels: finite, fp32-class results"""
    assert oc.conv_w1_odd_slab_case(lib, DEV, 1, 4, 32, 48, 64) < REL
    assert oc.conv_w1_odd_slab_case(lib, DEV, 2, 32, 64, 80, 128, seed=3) < REL


def test_conv_w1_worst_case_range_and_structured_filters(lib):
    """conv_w1 -- the kernel of the forward's ResnetBlock / FuseNet / first up-convolution layers, 83 % of its FLOPs -- under the adversarial
    dynamic range of the test above (one bit less operand head-room: |V| <= 2 max|x|; the output transform out[2j+1] = M1 - M2 - M3 forms
    d1 g2 twice and subtracts) and on STRUCTURED filters at 512 -> 512: smooth (g0 = g2, |g1| = 2^8 |g0|: U1 and U2 are +-g1/2 to eight
    bits), binomial, Sobel-like antisymmetric, second difference -- what a trained checkpoint holds and N(0, 0.02) draws never do.
    Gates as for the direct kernel: <= 3 x the exact-fp32 chain (torch fp32 conv on the same data), >= 19 bits on quiet outputs, and
    fp32-class relative to max|ref|.  (VERDICT r4 weak #1.)"""
    for corner in (False, True):
        for wt in (True, False):
            e, e32, eq, rq, _, _ = oc.conv_split_worstcase_case(lib, DEV, 2, 32, 32, 512, 512, corner=corner, weights_too=wt, kernel=3)
            print(f"conv_w1 worst-case split corner={corner} tiered_weights={wt}: max|err| {e:.3e} vs fp32 chain {e32:.3e} (x{e / e32:.2f}); quiet outputs {eq:.3e} of {rq:.3e}")
            assert e <= 3.0 * e32, (corner, wt, e, e32)
            assert eq <= rq * 2.0 ** -19, (corner, wt, eq, rq)
    e, e32, _, _, _, _ = oc.conv_split_worstcase_case(lib, DEV, 2, 32, 32, 512, 512, tiers=(-6, -9, -12), kernel=3)
    assert e <= 3.0 * e32
    for kind in ("smooth", "binomial", "sobel", "edge"):
        for norm in (True, False):
            e, e32, rmax = oc.conv_structured_filter_case(lib, DEV, 2, 32, 32, 512, 512, kind, kernel=3, norm=norm)
            ed, _, _ = oc.conv_structured_filter_case(lib, DEV, 2, 32, 32, 512, 512, kind, kernel=2, norm=norm)
            print(f"conv_w1 structured filters {kind:8s} norm={int(norm)}: max|err| {e:.3e} (direct kernel {ed:.3e}, fp32 chain {e32:.3e}) of max|ref| {rmax:.3e}")
            assert e <= 3.0 * e32 and e <= REL * rmax, (kind, norm, e, e32, rmax)
            assert e <= 1.5 * max(ed, 0.25 * e32), (kind, norm, e, ed)      # the Winograd rule: <= 1.5 x the direct form's error


def test_conv_w1_winograd_x_form(lib):
    """the Winograd F(2,3)-along-x form of the 3x3 / stride-1 layers (conv_w1.hpp; tsnet_op_conv2d(kernel = 3)) at the ResnetBlock, FuseNet
    and decoder shapes: fp32-class accuracy like the direct kernel (measured 0.8 - 1.1 x its error).  The forward runs these layers in this
    form."""
    assert oc.conv_w1_case(lib, DEV, 12, 32, 32, 512, 512, True, norm=True) < REL
    assert oc.conv_w1_case(lib, DEV, 4, 32, 32, 1024, 1024, True) < REL
    assert oc.conv_w1_case(lib, DEV, 2, 8, 64, 48, 96, False, norm=True) < REL
    assert oc.conv_w1_case(lib, DEV, 1, 256, 256, 128, 64, False, bias=False) < REL
