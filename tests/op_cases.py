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


def conv_case(lib, dev, N, H, W, Cin, Cout, k, stride, pad, reflect, norm=False, act=0, bias=True, seed=0):
    """nn.Conv2d (+ReflectionPad2d / zero pad, + consumer-side IN+ReLU) vs tsnet_op_conv2d. Returns max|d|."""
    x = _rand(seed, "x", (N, Cin, H, W))
    w = _rand(seed, "w", (Cout, Cin, k, k)) * (2.0 / (Cin * k * k) ** 0.5)
    b = _rand(seed, "b", (Cout,)) if bias else None
    xin, al, be = x, None, None
    if norm:
        al = _rand(seed, "al", (N, Cin), 0.5, 1.5)
        be = _rand(seed, "be", (N, Cin), -0.3, 0.3)
        xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    if reflect:
        ref = F.conv2d(F.pad(xin, (pad,) * 4, mode="reflect"), w, b, stride=stride)
    else:
        ref = F.conv2d(xin, w, b, stride=stride, padding=pad)
    if act:
        ref = torch.tanh(ref)
    Ho, Wo = ref.shape[2:]
    xd = nhwc(x).to(dev)
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    wd, bd = w.to(dev), (b.to(dev) if bias else None)
    rc = lib.tsnet_op_conv2d(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), _p(bd), Cout, k, stride, pad, int(reflect),
                             _p(ald), _p(bed), 1 if norm else 0, act, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (nchw(y.cpu()) - ref).abs().max().item()



def conv_h2_case(lib, dev, N, H, W, Cin, Cout, reflect, norm=False, bias=True, nprod=3, tile_n=0, seed=0, scale=1.0, return_output=False):
    """nn.Conv2d 3x3 / stride 1 (+ReflectionPad2d(1) or zero pad 1), optionally on relu(x*alpha+beta) -- the consumer side of
    nn.InstanceNorm2d + nn.ReLU -- vs tsnet_op_conv2d_h2 (fp16x2 patch kernel, transform fused into the patch staging).
    Returns max|d| relative to max|ref|."""
    x = _rand(seed, "x", (N, Cin, H, W)) * scale
    w = _rand(seed, "w", (Cout, Cin, 3, 3)) * (2.0 / (Cin * 9) ** 0.5)
    b = _rand(seed, "b", (Cout,)) if bias else None
    xin, al, be = x, None, None
    if norm:
        al = _rand(seed, "al", (N, Cin), 0.5, 1.5)
        be = _rand(seed, "be", (N, Cin), -0.3, 0.3)
        xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    if reflect:
        ref = F.conv2d(F.pad(xin.double(), (1,) * 4, mode="reflect"), w.double(), None if b is None else b.double())
    else:
        ref = F.conv2d(xin.double(), w.double(), None if b is None else b.double(), padding=1)
    bound = float(xin.abs().max()) * 1.0001 + 1e-30
    xd = nhwc(x).to(dev)
    y = torch.full((N, H, W, Cout), float("nan"), device=dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    wd, bd = w.to(dev), (b.to(dev) if bias else None)
    rc = lib.tsnet_op_conv2d_h2(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), _p(bd), Cout, int(reflect), _p(ald), _p(bed),
                                1 if norm else 0, bound, nprod, tile_n, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    if return_output:
        return y.cpu()
    return ((nchw(y.cpu()).double() - ref).abs().max() / ref.abs().max()).item()


def conv_h2r_case(lib, dev, N, H, W, Cin, Cout, ksize, norm=False, bias=True, seed=0):
    """the encoder's stem (7x7, reflection pad 3) and downsampling (3x3, stride 2, zero pad 1) convolutions, optionally on
    relu(x*alpha+beta), vs tsnet_op_conv2d_h2r (fp16 x 2 implicit GEMM with the transform fused).  Returns max|d| / max|ref|."""
    x = _rand(seed, "x", (N, Cin, H, W))
    w = _rand(seed, "w", (Cout, Cin, ksize, ksize)) * (2.0 / (Cin * ksize * ksize) ** 0.5)
    b = _rand(seed, "b", (Cout,)) if bias else None
    xin, al, be = x, None, None
    if norm:
        al = _rand(seed, "al", (N, Cin), 0.5, 1.5)
        be = _rand(seed, "be", (N, Cin), -0.3, 0.3)
        xin = F.relu(x * al[:, :, None, None] + be[:, :, None, None])
    bd64 = None if b is None else b.double()
    if ksize == 7:
        ref = F.conv2d(F.pad(xin.double(), (3,) * 4, mode="reflect"), w.double(), bd64)
    else:
        ref = F.conv2d(xin.double(), w.double(), bd64, stride=2, padding=1)
    bound = float(xin.abs().max()) * 1.0001 + 1e-30
    xd = nhwc(x).to(dev)
    y = torch.full((N, ref.shape[2], ref.shape[3], Cout), float("nan"), device=dev)
    ald = al.contiguous().to(dev) if norm else None
    bed = be.contiguous().to(dev) if norm else None
    wd, bd = w.to(dev), (b.to(dev) if bias else None)
    rc = lib.tsnet_op_conv2d_h2r(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), _p(bd), Cout, ksize, _p(ald), _p(bed), 1 if norm else 0, bound, 3,
                                 y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return ((nchw(y.cpu()).double() - ref).abs().max() / ref.abs().max()).item()

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
    warped = torch.empty((B, h, w, C), device=dev)
    rc = lib.tsnet_op_warp(srcd.data_ptr(), flow.data_ptr(), B, h, w, C, warped.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (flow.cpu() - flow_ref).abs().max().item(), (nchw(warped.cpu()) - warped_ref).abs().max().item()


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


def conv_x3_tiles_bitwise(lib, dev, N, H, W, Cin, Cout, tiles, reflect=True, seed=0):
    """Largest |difference| between the outputs of several tile configurations of one kernel family on the same
    3x3 / stride-1 layer (0.0 = bit-identical: tile choice must not change the arithmetic)."""
    x = F.relu(_rand(seed, "x", (N, Cin, H, W), -1.0, 2.0))
    w = _rand(seed, "w", (Cout, Cin, 3, 3)) * (2.0 / (Cin * 9) ** 0.5)
    b = _rand(seed, "b", (Cout,))
    xd, wd, bd = nhwc(x).to(dev), w.to(dev), b.to(dev)
    outs = []
    for t in tiles:
        y = torch.full((N, H, W, Cout), float("nan"), device=dev)
        rc = lib.tsnet_op_conv2d_x3(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), bd.data_ptr(), Cout, 3, 1, 1, int(reflect), t, y.data_ptr(), None)
        assert rc == 0, lib.tsnet_op_last_error().decode()
        _sync(dev)
        outs.append(y.cpu())
    return max((o - outs[0]).abs().max().item() for o in outs[1:])


def conv_x3_case(lib, dev, N, H, W, Cin, Cout, k, stride, pad, reflect, tile=-1, bias=True, seed=0):
    """nn.Conv2d (+padding) on the bf16x3 kernel (3-way bf16 operand split on the bf16 MFMA) vs PyTorch fp32."""
    x = F.relu(_rand(seed, "x", (N, Cin, H, W), -1.0, 2.0))
    w = _rand(seed, "w", (Cout, Cin, k, k)) * (2.0 / (Cin * k * k) ** 0.5)
    b = _rand(seed, "b", (Cout,)) if bias else None
    if reflect:
        ref = F.conv2d(F.pad(x, (pad,) * 4, mode="reflect"), w, b, stride=stride)
    else:
        ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    Ho, Wo = ref.shape[2:]
    xd = nhwc(x).to(dev)
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device=dev)
    wd, bd = w.to(dev), (b.to(dev) if bias else None)
    rc = lib.tsnet_op_conv2d_x3(xd.data_ptr(), N, H, W, Cin, wd.data_ptr(), _p(bd), Cout, k, stride, pad, int(reflect), tile, y.data_ptr(), None)
    assert rc == 0, lib.tsnet_op_last_error().decode()
    _sync(dev)
    return (nchw(y.cpu()) - ref).abs().max().item()
