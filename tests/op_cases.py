"""Operator-level parity cases, shared by the CPU-emulation tier (tests/test_emu_*.py) and the GPU
tier (tests/test_gpu_ops.py).  Every case calls through the C ABI (tsnet_op_*) and compares with
plain PyTorch CPU fp32 ops -- the same ATen ops the reference's modules resolve to."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from wacv23_tsnet_amd import prng


def _rand(seed, name, shape, lo=-1.0, hi=1.0):
    return prng.uniform01(seed, name, shape) * (hi - lo) + lo


def _p(t):
    return None if t is None else t.data_ptr()


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def conv_case(lib, dev, N, H, W, Cin, Cout, k, stride, pad, reflect, norm=False, bias=True, seed=0, nprod=3, kernel=0, tile=0, scale=1.0,
              relu=None, return_output=False):
    """nn.Conv2d (+ReflectionPad2d / zero pad), optionally on relu(x*alpha+beta) -- the consumer side of nn.InstanceNorm2d + nn.ReLU --
    vs tsnet_op_conv2d (fp16 x 2 operands, the transform fused into the operand staging).  kernel: 0 = the forward's own choice for the
    layer, 1 = general implicit GEMM, 2 = patch kernel.  Returns max|d| relative to max|fp64 reference| (or the NHWC output)."""
    x = _rand(seed, "x", (N, Cin, H, W)) * scale
    w = _rand(seed, "w", (Cout, Cin, k, k)) * (2.0 / (Cin * k * k) ** 0.5)
    b = _rand(seed, "b", (Cout,)) if bias else None
    xin, al, be = x, None, None
    relu = norm if relu is None else relu
    if norm:
        al = _rand(seed, "al", (N, Cin), 0.5, 1.5)
        be = _rand(seed, "be", (N, Cin), -0.3, 0.3)
        xin = x * al[:, :, None, None] + be[:, :, None, None]
    if relu:
        xin = F.relu(xin)
    b64 = None if b is None else b.double()
    if reflect:
        ref = F.conv2d(F.pad(xin.double(), (pad,) * 4, mode="reflect"), w.double(), b64, stride=stride)
    else:
        ref = F.conv2d(xin.double(), w.double(), b64, stride=stride, padding=pad)
    bound = float(xin.abs().max()) * 1.0001 + 1e-30
    Ho, Wo = ref.shape[2:]
    xd = nhwc(x).to(dev)
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    wd, bd = w.to(dev), (b.to(dev) if bias else None)
    rc = lib.tsnet_op_conv2d(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), _p(bd), Cout, k, stride, pad, int(reflect),
                             _p(ald), _p(bed), int(relu), bound, nprod, kernel, tile, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    if return_output:
        return y.cpu()
    return ((nchw(y.cpu()).double() - ref).abs().max() / ref.abs().max()).item()


def conv_h2_case(lib, dev, N, H, W, Cin, Cout, reflect, norm=False, bias=True, nprod=3, tile_n=0, seed=0, scale=1.0, return_output=False):
    """3x3 / stride 1 / pad 1 on the PATCH kernel (conv_h2.hpp h2_tile); tile_n: 0, 32, 64, 128 (4-row tiles), 2128 (2 rows x 128), 3128 (bf16 operands: 4 x 128, four waves side by side), 20032 / 20064 (two K groups, 4 x 32 / 4 x 64)."""
    return conv_case(lib, dev, N, H, W, Cin, Cout, 3, 1, 1, reflect, norm=norm, bias=bias, seed=seed, nprod=nprod, kernel=2, tile=tile_n,
                     scale=scale, return_output=return_output)


def conv_w1_case(lib, dev, N, H, W, Cin, Cout, reflect, norm=False, bias=True, nprod=3, seed=0, scale=1.0, relu=None, return_output=False, chunk=0):
    """3x3 / stride 1 / pad 1 in the Winograd F(2,3)-along-x form (conv_w1.hpp; kernel = 3): same contract as conv_h2_case.  chunk: tiles per
    workgroup (1, 2, 3; 0 = the launcher's choice)"""
    return conv_case(lib, dev, N, H, W, Cin, Cout, 3, 1, 1, reflect, norm=norm, bias=bias, seed=seed, nprod=nprod, kernel=3, tile=chunk, scale=scale,
                     relu=relu, return_output=return_output)


def conv_h2r_case(lib, dev, N, H, W, Cin, Cout, ksize, norm=False, bias=True, seed=0, kernel=0, tile=0, return_output=False):
    """the encoder's stem (7x7, reflection pad 3) and downsampling (3x3, stride 2, zero pad 1) convolutions; kernel as conv_case."""
    if ksize == 7:
        return conv_case(lib, dev, N, H, W, Cin, Cout, 7, 1, 3, True, norm=norm, bias=bias, seed=seed, kernel=kernel, tile=tile, return_output=return_output)
    return conv_case(lib, dev, N, H, W, Cin, Cout, 3, 2, 1, False, norm=norm, bias=bias, seed=seed, kernel=kernel, tile=tile, return_output=return_output)


def conv_cat_case(lib, dev, N, H, W, C1, C2, Cout, k=1, seed=0, shared=False):
    """convolution of torch.cat((x, x2), dim=1) with the concat formed on load (dec.map_conv on cat(pg, sg), TSNet.py:163); `shared`: x2 has
    one image that every image of x is concatenated with (image index n % x2_nmod)."""
    x = _rand(seed, "x", (N, C1, H, W))
    x2 = _rand(seed, "x2", (1 if shared else N, C2, H, W)) * 3.0
    w = _rand(seed, "w", (Cout, C1 + C2, k, k)) * (2.0 / ((C1 + C2) * k * k) ** 0.5)
    b = _rand(seed, "b", (Cout,))
    xin = torch.cat([x, x2.expand(N, -1, -1, -1)], dim=1)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=k // 2)
    y = torch.full((N, H, W, Cout), float("nan"), device=dev)
    xd, x2d, wd, bd = nhwc(x).to(dev), nhwc(x2).to(dev), w.to(dev), b.to(dev)
    rc = lib.tsnet_op_conv2d_cat(xd.data_ptr(), x2d.data_ptr(), N, H, W, C1, C2, x2.shape[0], wd.data_ptr(), bd.data_ptr(), Cout, k, 1, k // 2, 0,
                                 float(xin.abs().max()) * 1.0001, 3, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return ((nchw(y.cpu()).double() - ref).abs().max() / ref.abs().max()).item()


def head_case(lib, dev, N, H, W, C, norm=True, composite=False, seed=0, rows=0, return_output=False):
    """the decoder's RGB head: ReflectionPad2d(3) + Conv2d(C -> 3, 7x7) + bias + Tanh on relu(IN(x)) (TSNet.py:151-152), optional pose
    composite (TSNet_pose.py:416-417) vs tsnet_op_head.  rows: tile rows of head_conv3 to force (8, 16, 32; 0 = the launcher's choice).
    Returns max|d| (or the output)."""
    x = _rand(seed, "x", (N, C, H, W))
    w = _rand(seed, "w", (3, C, 7, 7)) * (2.0 / (C * 49) ** 0.5)
    b = _rand(seed, "b", (3,))
    xin, al, be = x, None, None
    if norm:
        al = _rand(seed, "al", (N, C), 0.5, 1.5)
        be = _rand(seed, "be", (N, C), -0.3, 0.3)
        xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    ref = torch.tanh(F.conv2d(F.pad(xin, (3,) * 4, mode="reflect"), w, b))
    bg = [0.25, -0.5, 0.75]
    if composite:
        for c in range(3):
            ref[:, c, :, :64] = bg[c]
            ref[:, c, :, 192:] = bg[c]
    import ctypes
    y = torch.full((N, 3, H, W), float("nan"), device=dev)
    xd, wd, bd = nhwc(x).to(dev), w.to(dev), b.to(dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    rc = lib.tsnet_op_head(xd.data_ptr(), N, H, W, C, _p(ald), _p(bed), wd.data_ptr(), bd.data_ptr(), int(composite) | (rows << 8), (ctypes.c_float * 3)(*bg),
                           y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    if return_output:
        return y.cpu()
    return (y.cpu() - ref).abs().max().item()


def instnorm_case(lib, dev, N, H, W, C, relu, resid, seed=0, offset=0.0):
    """InstanceNorm2d(eps=1e-5, biased var) [+ReLU] [+residual] vs stats + norm_act kernels."""
    x = _rand(seed, "x", (N, C, H, W), -2, 2) + offset
    ref = F.instance_norm(x, eps=1e-5)
    if relu:
        ref = F.relu(ref)
    r = _rand(seed, "r", (N, C, H, W)) if resid else None
    if resid:
        ref = r + ref
    xd = nhwc(x).to(dev)
    al = torch.empty(N * C, device=dev)
    be = torch.empty(N * C, device=dev)
    rc = lib.tsnet_op_instnorm_stats(xd.data_ptr(), N, H * W, C, al.data_ptr(), be.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    y = torch.empty_like(xd)
    rd = nhwc(r).to(dev) if resid else None
    rc = lib.tsnet_op_norm_act(xd.data_ptr(), al.data_ptr(), be.data_ptr(), int(relu), _p(rd), N, H * W, C, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (nchw(y.cpu()) - ref).abs().max().item()


def upsample_case(lib, dev, N, H, W, C, norm, seed=0):
    """nn.Upsample(x2, bilinear, align_corners=False) [after IN+ReLU] vs tsnet_op_upsample2x."""
    x = _rand(seed, "x", (N, C, H, W), -2, 2)
    xin, al, be = x, None, None
    if norm:
        al = _rand(seed, "al", (N, C), 0.5, 1.5)
        be = _rand(seed, "be", (N, C), -0.3, 0.3)
        xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    ref = F.interpolate(xin, scale_factor=2, mode="bilinear", align_corners=False)
    xd = nhwc(x).to(dev)
    y = torch.empty((N, 2 * H, 2 * W, C), device=dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    rc = lib.tsnet_op_upsample2x(xd.data_ptr(), _p(ald), _p(bed), int(norm), N, H, W, C, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (nchw(y.cpu()) - ref).abs().max().item()


def flow_case(lib, dev, B, h, w, C, mask_mode="bernoulli", seed=0, spike=False):
    """Transformation branch up to the flow (TSNet.py:319-365) + grid_sample (:366) vs the oracle's
    transformation_branch.  spike=True plants one dominant source per target (forces online-softmax
    rescales in every tile order).  Returns (max|d flow|, max|d warped|)."""
    from oracle import tsnet_oracle as O
    H, W = h * 8, w * 8
    tar = F.relu(_rand(seed, "tar", (B, C, h, w), -1, 1))
    src = _rand(seed, "src", (B, C, h, w), -1, 1) * 3
    if spike:
        P = h * w
        perm = torch.arange(P - 1, -1, -1)
        srcf = src.view(B, C, P)
        tarf = tar.view(B, C, P)
        srcf[:, :, perm] = srcf[:, :, perm] * 0.2 + 4.0 * tarf    # source perm[t] ~ aligned with target t
    if mask_mode == "ones":
        mt, ms = torch.ones(B, H, W), torch.ones(B, H, W)
    elif mask_mode == "zeros":
        mt, ms = torch.zeros(B, H, W), torch.zeros(B, H, W)
    elif mask_mode == "soft":
        mt, ms = prng.uniform01(seed, "mt", (B, H, W)), prng.uniform01(seed, "ms", (B, H, W))
    else:
        mt, ms = prng.bernoulli(seed, "mt", (B, H, W)), prng.bernoulli(seed, "ms", (B, H, W))
    warped_ref, flow_ref = O.transformation_branch(tar, src, mt.unsqueeze(1), ms.unsqueeze(1))
    tard, srcd = nhwc(tar).to(dev), nhwc(src).to(dev)
    mtd, msd = mt.to(dev), ms.to(dev)
    flow = torch.empty((B, h, w, 2), device=dev)
    rc = lib.tsnet_op_flow(tard.data_ptr(), srcd.data_ptr(), mtd.data_ptr(), msd.data_ptr(), B, h, w, C, H, W, flow.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    # a second run must give the same bits (the packed-fp32 build of the large-map kernel did not: csrc/flow_persist.hpp; both flow kernels
    # are now compiled without the SLP vectorizer and both are held to this)
    flow2 = torch.full((B, h, w, 2), float("nan"), device=dev)
    rc = lib.tsnet_op_flow(tard.data_ptr(), srcd.data_ptr(), mtd.data_ptr(), msd.data_ptr(), B, h, w, C, H, W, flow2.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    assert torch.equal(flow, flow2)
    warped = torch.empty((B, h, w, C), device=dev)
    rc = lib.tsnet_op_warp(srcd.data_ptr(), flow.data_ptr(), B, h, w, C, warped.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (flow.cpu() - flow_ref).abs().max().item(), (nchw(warped.cpu()) - warped_ref).abs().max().item()


def flow_k_case(lib, dev, B, K, h, w, C, mask_mode="bernoulli", seed=0, spike=True):
    """K sources per driving frame (the model's loop, TSNet.py:336-366) through tsnet_op_flow_k vs the oracle's transformation_branch run
    once per source.  Maps of >= 2048 positions take flow_kernel_p (csrc/flow_persist.hpp): a workgroup keeps its target tile for the K
    sources, G workgroups share a (source, target tile) and the last to arrive merges their partial softmax states.  Both calls start from
    fresh scratch memory: a partial state read before it is visible would show as a difference.  Returns max|d flow| over all sources."""
    from oracle import tsnet_oracle as O
    H, W = h * 8, w * 8
    P = h * w
    tar = F.relu(_rand(seed, "tar", (B, C, h, w), -1, 1))
    mt = prng.bernoulli(seed, "mt", (B, H, W)) if mask_mode == "bernoulli" else prng.uniform01(seed, "mt", (B, H, W))
    srcs, mss, refs = [], [], []
    for k in range(K):
        src = _rand(seed + 1 + k, "src", (B, C, h, w), -1, 1) * 3
        if spike:                                                         # one dominant source per target, somewhere else for every k
            perm = (torch.arange(P) * (2 * k + 3) + 7 * k) % P if P % (2 * k + 3) else torch.arange(P - 1, -1, -1)
            srcf, tarf = src.view(B, C, P), tar.view(B, C, P)
            srcf[:, :, perm] = srcf[:, :, perm] * 0.2 + 4.0 * tarf
        ms = prng.bernoulli(seed + 1 + k, "ms", (B, H, W)) if mask_mode == "bernoulli" else prng.uniform01(seed + 1 + k, "ms", (B, H, W))
        _, flow_ref = O.transformation_branch(tar, src, mt.unsqueeze(1), ms.unsqueeze(1))
        srcs.append(nhwc(src)); mss.append(ms); refs.append(flow_ref)
    tard, srcd = nhwc(tar).to(dev), torch.cat(srcs, 0).contiguous().to(dev)
    mtd, msd = mt.to(dev), torch.cat(mss, 0).contiguous().to(dev)
    flow = torch.empty((K * B, h, w, 2), device=dev)
    rc = lib.tsnet_op_flow_k(tard.data_ptr(), srcd.data_ptr(), mtd.data_ptr(), msd.data_ptr(), B, K, h, w, C, H, W, flow.data_ptr(), 0, 1, None, None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    # a second call must give the same bits (arrival counters back at zero, merge order fixed)
    flow2 = torch.empty_like(flow)
    rc = lib.tsnet_op_flow_k(tard.data_ptr(), srcd.data_ptr(), mtd.data_ptr(), msd.data_ptr(), B, K, h, w, C, H, W, flow2.data_ptr(), 0, 2, None, None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    assert torch.equal(flow, flow2)
    return (flow.cpu() - torch.cat(refs, 0)).abs().max().item()


def flow_k_batch_independence_case(lib, dev, batches, K, h, w, C, seed=0):
    """flow_kernel_p splits every source image into a number of slices that depends on the MAP alone and merges them in slice order; how
    many workgroups share a target tile (G: 4, 2, 1 as the batch grows) only decides who computes which slice.  The flows of the first
    batch element must therefore be the same BITS in every batch size (the tree used to be built on G).  Returns the list of
    flows of element 0, one per batch size."""
    H, W, P = h * 8, w * 8, h * w
    Bmax = max(batches)
    tar = F.relu(_rand(seed, "tar", (Bmax, C, h, w), -1, 1))
    mt = prng.bernoulli(seed, "mt", (Bmax, H, W))
    srcs = [_rand(seed + 1 + k, "src", (Bmax, C, h, w), -1, 1) * 3 for k in range(K)]
    for k in range(K):
        perm = (torch.arange(P) * (2 * k + 3) + 7 * k) % P if P % (2 * k + 3) else torch.arange(P - 1, -1, -1)
        srcs[k].view(Bmax, C, P)[:, :, perm] = srcs[k].view(Bmax, C, P)[:, :, perm] * 0.2 + 4.0 * tar.view(Bmax, C, P)
    mss = [prng.bernoulli(seed + 1 + k, "ms", (Bmax, H, W)) for k in range(K)]
    out = []
    for B in batches:
        tard = nhwc(tar[:B]).to(dev)
        srcd = torch.cat([nhwc(s[:B]) for s in srcs], 0).contiguous().to(dev)
        mtd, msd = mt[:B].contiguous().to(dev), torch.cat([m[:B] for m in mss], 0).contiguous().to(dev)
        flow = torch.empty((K * B, h, w, 2), device=dev)
        rc = lib.tsnet_op_flow_k(tard.data_ptr(), srcd.data_ptr(), mtd.data_ptr(), msd.data_ptr(), B, K, h, w, C, H, W, flow.data_ptr(), 0, 1, None, None)
        assert rc == 0, lib.tsnet_op_last_error().decode()
        _sync(dev)
        out.append(flow.cpu().view(K, B, h, w, 2)[:, 0].clone())
    return out


def warp_case(lib, dev, B, h, w, C, seed=0):
    """F.grid_sample(bilinear, zeros, align_corners=False) with flows that leave [-1,1] (zero padding
    and border blending exercised) vs tsnet_op_warp."""
    src = _rand(seed, "src", (B, C, h, w), -2, 2)
    flow = _rand(seed, "flow", (B, h, w, 2), -1.3, 1.3)
    ref = F.grid_sample(src, flow, align_corners=False)
    srcd, fd = nhwc(src).to(dev), flow.to(dev)
    out = torch.empty((B, h, w, C), device=dev)
    rc = lib.tsnet_op_warp(srcd.data_ptr(), fd.data_ptr(), B, h, w, C, out.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (nchw(out.cpu()) - ref).abs().max().item()


def conv_split_worstcase_case(lib, dev, N, H, W, Cin, Cout, tiers=(-12, -18, -24), weights_too=True, corner=False, seed=0, kernel=2):
    """kernel: 2 = the direct patch kernel (conv_h2), 3 = the Winograd-along-x kernel (conv_w1: the forward's ResnetBlock / FuseNet layers --
    one bit less operand head-room, |V| <= 2 max|x|, and the output transform's own cancellation M1 - M2 - M3).
    Adversarial dynamic range INSIDE one image for the fp16 x 2 operand split (conv_common.hpp): 1 % of the activations sit at the
    image's maximum, the rest in equal shares at amax * 2^t for t in `tiers` (the operand scale is derived from amax alone, so the bulk
    lands deep in fp16's lower range, the last tier in its subnormals); the weights likewise.  corner=True confines the amax values to the
    top-left 8 x 8 pixels, so most outputs are sums of tiny terms only.
    Returns (max|y - ref64|, max|conv_fp32 - ref64|, max|y - ref64| over outputs that see no amax activation, their max|ref64|, amax_x, amax_w):
    the exact-fp32 chain on the same data is the yardstick (torch fp32 conv on the CPU)."""
    def tiered(name, shape, amax, spatial=None):
        u = prng.uniform01(seed, name + "_u", shape)                     # magnitude within a tier: [0.5, 1)
        sgn = torch.where(prng.uniform01(seed, name + "_s", shape) < 0.5, -1.0, 1.0)
        sel = prng.uniform01(seed, name + "_t", shape)
        e = torch.zeros(shape)
        nt = len(tiers)
        for i, t in enumerate(tiers):
            e = torch.where((sel >= 0.01 + 0.99 * i / nt) & (sel < 0.01 + 0.99 * (i + 1) / nt), float(t), e)
        top = sel < 0.01
        if spatial is not None:
            top = top & spatial
            e = torch.where((sel < 0.01) & ~spatial, float(tiers[0]), e)
        e = torch.where(top, 0.0, e)
        v = sgn * (0.5 + 0.5 * u) * torch.exp2(e) * amax
        if top.any():
            v.view(-1)[top.view(-1).nonzero()[0]] = amax                 # the maximum itself is present
        return v.float(), top

    sp = None
    if corner:
        sp = torch.zeros(N, Cin, H, W, dtype=torch.bool)
        sp[:, :, :8, :8] = True
    x, xtop = tiered("x", (N, Cin, H, W), 3.0, sp)
    if weights_too:
        w, _ = tiered("w", (Cout, Cin, 3, 3), 2.0 / (Cin * 9) ** 0.5)
    else:
        w = _rand(seed, "w", (Cout, Cin, 3, 3)) * (2.0 / (Cin * 9) ** 0.5)
    xp = F.pad(x, (1,) * 4, mode="reflect")
    ref = F.conv2d(xp.double(), w.double())
    y32 = F.conv2d(xp, w)
    bound = float(x.abs().max()) * 1.0001
    xd, wd = nhwc(x).to(dev), w.to(dev)
    y = torch.full((N, H, W, Cout), float("nan"), device=dev)
    rc = lib.tsnet_op_conv2d(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), None, Cout, 3, 1, 1, 1, None, None, 0, bound, 3, kernel, 0, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    yc = nchw(y.cpu()).double()
    e_h2 = (yc - ref).abs().max().item()
    e_32 = (y32.double() - ref).abs().max().item()
    # outputs whose 3 x 3 receptive field holds no amax activation in any channel
    seen = F.max_pool2d(F.pad(xtop.any(dim=1, keepdim=True).float(), (1,) * 4, mode="reflect"), 3, 1)      # (N,1,H,W)
    quiet = (seen == 0).expand(-1, Cout, -1, -1)
    e_quiet = (yc - ref)[quiet].abs().max().item() if quiet.any() else 0.0
    r_quiet = ref[quiet].abs().max().item() if quiet.any() else 0.0
    return e_h2, e_32, e_quiet, r_quiet, float(x.abs().max()), float(w.abs().max())


def conv_structured_filter_case(lib, dev, N, H, W, Cin, Cout, kind, kernel=3, norm=True, seed=0):
    """STRUCTURED 3 x 3 filters, as a trained checkpoint holds them (the op tests' and the seed sweep's weights are i.i.d. N(0, 0.02) / uniform):
    every (output, input, tap row) draws one amplitude a and the three taps along x follow a fixed profile --
      "smooth"    a * (2^-8, 1, 2^-8):  g0 = g2, |g1| >> |g0|: the Winograd filters U1 = (g0+g1+g2)/2 and U2 = (g0-g1+g2)/2 are +-g1/2 to eight bits
                  and the small taps live in their difference;
      "binomial"  a * (1/2, 1, 1/2):    a low-pass filter, g0 = g2;
      "sobel"     a * (-1, 0, 1):       antisymmetric: g0 + g2 = 0 and g1 = 0, so U1 = U2 = 0 and the whole output comes from M0 = V0 g0 and
                  M3 = V3 g2;
      "edge"      a * (1, -2, 1):       second difference: U1 = 0, U2 = 2 a -- out[2j+1] = M1 - M2 - M3 is all M2.
    Input: relu(x * alpha + beta) with the fused transform when `norm` (the ResnetBlock's second convolution), zero-free uniform data
    otherwise.  Returns (max|y - ref64|, max|conv_fp32 - ref64|, max|ref64|): the exact-fp32 chain (torch fp32 conv on the CPU) is the yardstick."""
    prof = {"smooth": (2.0 ** -8, 1.0, 2.0 ** -8), "binomial": (0.5, 1.0, 0.5), "sobel": (-1.0, 0.0, 1.0), "edge": (1.0, -2.0, 1.0)}[kind]
    a = _rand(seed, "wa", (Cout, Cin, 3, 1)) * (2.0 / (Cin * 9) ** 0.5)
    w = (a * torch.tensor(prof, dtype=torch.float32).view(1, 1, 1, 3)).contiguous()
    x = _rand(seed, "x", (N, Cin, H, W))
    xin, al, be = x, None, None
    if norm:
        al = _rand(seed, "al", (N, Cin), 0.5, 1.5)
        be = _rand(seed, "be", (N, Cin), -0.3, 0.3)
        xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    xp = F.pad(xin, (1,) * 4, mode="reflect")
    ref = F.conv2d(xp.double(), w.double())
    y32 = F.conv2d(xp, w)
    bound = float(xin.abs().max()) * 1.0001
    xd, wd = nhwc(x).to(dev), w.to(dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    y = torch.full((N, H, W, Cout), float("nan"), device=dev)
    rc = lib.tsnet_op_conv2d(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), None, Cout, 3, 1, 1, 1, _p(ald), _p(bed), int(norm), bound, 3, kernel, 0, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    yc = nchw(y.cpu()).double()
    return (yc - ref).abs().max().item(), (y32.double() - ref).abs().max().item(), ref.abs().max().item()


def conv_w1_odd_slab_case(lib, dev, N=1, H=4, W=32, Cin=48, Cout=64, seed=0):
    """conv_w1 with an ODD number of 16-channel slabs and a fused InstanceNorm: the last period stages 16 channels past Cin.  They
    used to be the next pixel's channels 0..15 transformed with the affine of channels 0..3 -- not bounded by the operand bound; beyond
    fp16's range they became inf, and inf x (zero weights) = NaN reached real outputs.  Here channels 0..3 carry a gain of 1e3 on tiny
    values (so the layer's bound stays ~1) while the next pixel's channels 16..19 are O(1): the old staging saw 1e3.  Now the lanes past Cin
    read zeros and the table holds zeros there.  Returns max|d| relative to max|fp64 reference| (NaN-safe: a NaN fails the comparison)."""
    x = _rand(seed, "x", (N, Cin, H, W))
    x[:, :4] *= 1e-3
    al = _rand(seed, "al", (N, Cin), 0.5, 1.5)
    al[:, :4] = 1e3
    be = _rand(seed, "be", (N, Cin), -0.3, 0.3)
    w = _rand(seed, "w", (Cout, Cin, 3, 3)) * (2.0 / (Cin * 9) ** 0.5)
    xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    ref = F.conv2d(F.pad(xin.double(), (1,) * 4, mode="reflect"), w.double())
    bound = float(xin.abs().max()) * 1.0001
    xd, wd, ald, bed = nhwc(x).to(dev), w.to(dev), al.contiguous().to(dev), be.contiguous().to(dev)
    y = torch.full((N, H, W, Cout), float("nan"), device=dev)
    rc = lib.tsnet_op_conv2d(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), None, Cout, 3, 1, 1, 1, ald.data_ptr(), bed.data_ptr(), 1, bound, 3, 3, 0, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    yc = nchw(y.cpu()).double()
    assert torch.isfinite(yc).all(), "non-finite outputs"
    return ((yc - ref).abs().max() / ref.abs().max()).item()


def conv_g64_cases(lib, dev, big=False):
    """conv_g64 (csrc/conv_g64.hpp: the general implicit GEMM in 64-deep K steps) against conv_h2r (16-deep steps) on the layer kinds the
    forward sends it -- the two kernels run the same chains, so the outputs must be EQUAL BITS -- and against the fp64 reference.  Shapes:
    1 x 1 (fuse_net.conv / dec.map_conv), 3 x 3 stride 2 with the producer's InstanceNorm + ReLU on load (the 64 -> 128 layer; with bf16
    operands every stride-2 layer), 3 x 3 stride 1 with reflection on a ragged frame, 192 channels (three steps per tap), both tile heights;
    the concat formed on load (own channel split per tensor, shared second image).  Returns the worst relative error of the fp16 x 2 cases."""
    worst = 0.0
    shapes = [(2, 8, 8, 128, 128, 1, 1, 0, False, False, 3, 64), (1, 16, 16, 64, 128, 3, 2, 1, False, True, 3, 128), (2, 10, 12, 64, 256, 3, 1, 1, True, True, 3, 128),
              (1, 16, 16, 128, 128, 3, 2, 1, False, True, 1, 128), (1, 9, 7, 192, 128, 3, 2, 1, False, False, 3, 128)]
    if big:      # the forward's own shapes at the headline batch
        shapes += [(4, 32, 32, 1024, 512, 1, 1, 0, False, False, 3, 64), (3, 256, 256, 64, 128, 3, 2, 1, False, True, 3, 128), (2, 64, 64, 256, 512, 3, 2, 1, False, True, 1, 128)]
    for (N, H, W, Cin, Cout, k, st, pad, refl, norm, nprod, ref_tile) in shapes:
        a = conv_case(lib, dev, N, H, W, Cin, Cout, k, st, pad, refl, norm=norm, nprod=nprod, kernel=1, tile=ref_tile, return_output=True)
        for tile in (3064, 3128):
            b = conv_case(lib, dev, N, H, W, Cin, Cout, k, st, pad, refl, norm=norm, nprod=nprod, kernel=1, tile=tile, return_output=True)
            assert torch.equal(a, b), (N, H, W, Cin, Cout, k, st, nprod, tile, (a - b).abs().max().item())
        if nprod == 3:
            worst = max(worst, conv_case(lib, dev, N, H, W, Cin, Cout, k, st, pad, refl, norm=norm, nprod=nprod, kernel=1, tile=3064))
        c = conv_case(lib, dev, N, H, W, Cin, Cout, k, st, pad, refl, norm=norm, nprod=nprod, kernel=1, tile=0, return_output=True)    # the launcher's own choice
        assert torch.equal(a, c)
    worst = max(worst, conv_cat_case(lib, dev, 2, 8, 8, 64, 64, 128), conv_cat_case(lib, dev, 2, 8, 8, 128, 64, 256, shared=True))
    if big:
        worst = max(worst, conv_cat_case(lib, dev, 4, 32, 32, 512, 512, 512))
    return worst


def conv_h2s32_cases(lib, dev, big=False):
    """The pose model's 7 x 7 stems (32 padded input channels) on their patch kernel (csrc/conv_h2s32.hpp; kernel = 0: the layer's own) against
    the general kernel (kernel = 1, conv_h2r) -- same K order and chains: EQUAL BITS -- and against the fp64 reference: one tile, several tiles
    and images with reflection at all four borders, a narrow net (8 output channels), three channel tiles with a ragged last one, bf16 operands."""
    worst = 0.0
    shapes = [(1, 4, 32, 64, 3, True, 0), (2, 8, 64, 64, 3, False, 3), (1, 12, 32, 8, 3, True, 5), (1, 4, 64, 130, 3, True, 7), (1, 8, 32, 64, 1, True, 9)]
    if big:
        shapes += [(2, 256, 256, 64, 3, True, 11)]
    for (N, H, W, Cout, nprod, bias, seed) in shapes:
        a = conv_case(lib, dev, N, H, W, 32, Cout, 7, 1, 3, True, nprod=nprod, bias=bias, seed=seed, kernel=1, tile=64, return_output=True)
        b = conv_case(lib, dev, N, H, W, 32, Cout, 7, 1, 3, True, nprod=nprod, bias=bias, seed=seed, kernel=0, return_output=True)
        assert torch.equal(a, b), (N, H, W, Cout, nprod, (a - b).abs().max().item())
        if nprod == 3:
            worst = max(worst, conv_case(lib, dev, N, H, W, 32, Cout, 7, 1, 3, True, nprod=nprod, bias=bias, seed=seed, kernel=2))
    return worst
