"""CPU oracle for the TS-Net generator forward path -- TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  The shipped path
(`wacv23_tsnet_amd`) never falls back to it; it raises if the HIP library is missing.

It is a from-scratch restatement, in plain PyTorch CPU fp32 functional ops, of the
reference's test-mode forward (`/root/reference/model/TSNet.py:309-407`,
`is_train=False`) and of the pose variant's compositing epilogue
(`model/TSNet_pose.py:276-280,416-417`).  Each function cites the reference lines
it follows.  All arithmetic lives in third-party PyTorch ATen (reference pin:
torch==1.10.1+cu102, requirements.txt:9); here it runs on this image's torch 2.10 CPU
kernels, whose documented semantics for these ops are unchanged.

Parity pin: the reference has no tests or golden vectors, so
the pin is `oracle/capture_goldens.py`, which imports the real reference in the
authoring container, loads PRNG-generated weights into it and stores its outputs
under `tests/golden/`; `tests/test_oracle_golden.py` checks this restatement against
those vectors (bit-exact at the capture thread count, <=2e-5 otherwise).

Weights are a flat dict keyed exactly like the reference checkpoints
(`'<net>.<state_dict key>'`, nets = img_enc, lbl_enc, fuse_net, dec;
train_face.py:350-355).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

POSE_MEAN = (101.84807705937696, 112.10832843463207, 111.65973036298041)  # TSNet_pose.py:215


@dataclass
class TSNetConfig:
    """Constructor arguments of the reference model that shape the forward graph
    (model/TSNet.py:204-210; pose extras model/TSNet_pose.py:214-215)."""
    label_nc: int = 2
    n_blocks: int = 0          # decoder ResnetBlocks
    n_downsampling: int = 3
    n_source: int = 3
    ngf: int = 64
    addcoords: bool = True
    enc_blocks: int = 9        # Encoder default n_blocks (TSNet.py:53); lbl_enc uses 0 (TSNet.py:221)
    fuse_ngf: int = 1024       # hard-coded in the reference (TSNet.py:227)
    fuse_blocks: int = 1
    pose: bool = False         # TSNet_pose variant
    use_mask: bool = True      # pose only
    mean: Sequence[float] = field(default_factory=lambda: POSE_MEAN)

    @property
    def feat_ch(self) -> int:
        return self.ngf * (2 ** self.n_downsampling)


# --------------------------------------------------------------------------- operand rounding (bf16-operand mode)
_ROUND_BF16 = False     # set by tsnet_forward(round_operands="bf16") for the duration of one call


def _r(t: torch.Tensor) -> torch.Tensor:
    """bf16-operand mode of the engine (tsnet_cfg.operand_mode = 1; BASELINE.json configs[2] / [4]): every convolution input and weight is
    rounded to bfloat16 (round to nearest even) where the convolution reads it; products and sums stay fp32.  Identity otherwise."""
    return t.to(torch.bfloat16).to(t.dtype) if _ROUND_BF16 else t


def _conv(x, w, b=None, **kw):
    return F.conv2d(_r(x), _r(w), b, **kw)


_STORE_BF16 = False     # set by tsnet_forward(round_operands="bf16s"): the engine's bf16-STORAGE mode (tsnet_cfg.operand_mode = 2)


def _s(t: torch.Tensor) -> torch.Tensor:
    """bf16 storage: a tensor the engine keeps in bf16 between two kernels is rounded once where it is written"""
    return t.to(torch.bfloat16).to(t.dtype) if _STORE_BF16 else t


def _in_s(x: torch.Tensor) -> torch.Tensor:
    """InstanceNorm of a convolution output that the bf16-storage mode keeps in bf16: the statistics come from the fp32 accumulators (the
    unrounded tensor), the normalised tensor is the stored (rounded) one"""
    if not _STORE_BF16:
        return _in(x)
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (_s(x) - mean) * torch.rsqrt(var + 1e-5)


class bf16_operands:
    """context manager: the layer functions of this module (encoder, fuse_net, decoder, ...) round their convolution operands to
    bf16 inside it (storage=True: and keep the large activations in bf16, operand_mode 2) -- for checks of one stage of the engine's bf16
    modes in isolation"""

    def __init__(self, storage: bool = False):
        self.storage = storage

    def __enter__(self):
        global _ROUND_BF16, _STORE_BF16
        self._old, _ROUND_BF16 = (_ROUND_BF16, _STORE_BF16), True
        _STORE_BF16 = self.storage
        return self

    def __exit__(self, *a):
        global _ROUND_BF16, _STORE_BF16
        _ROUND_BF16, _STORE_BF16 = self._old
        return False


# --------------------------------------------------------------------------- layers
def coord_conv(x: torch.Tensor) -> torch.Tensor:
    """Append xx, yy, rr channels (Encoder.coord_conv, TSNet.py:107-125).

    xx = 2*j/(w-1)-1, yy = 2*i/(h-1)-1 computed in float32 as arange/(n-1) then 2*.-1,
    rr = sqrt(xx^2+yy^2); channel order (x, xx, yy, rr)."""
    bs, _, h, w = x.shape
    xs = torch.arange(w, dtype=x.dtype) / (w - 1)
    ys = torch.arange(h, dtype=x.dtype) / (h - 1)
    # materialised (contiguous) like the reference's matmul outputs, so torch.pow takes the same
    # vectorised x*x path as in the reference (an expanded view goes through scalar powf: 1-ulp differences)
    xx = (2 * xs - 1).view(1, 1, 1, w).expand(bs, 1, h, w).contiguous()
    yy = (2 * ys - 1).view(1, 1, h, 1).expand(bs, 1, h, w).contiguous()
    rr = torch.sqrt(torch.pow(xx, 2) + torch.pow(yy, 2))
    return torch.cat((x, xx, yy, rr), dim=1)


def _in(x: torch.Tensor) -> torch.Tensor:
    """nn.InstanceNorm2d defaults: affine=False, no running stats, eps=1e-5, biased var."""
    return F.instance_norm(x, eps=1e-5)


def resnet_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """x + IN(conv3(reflpad1(relu(IN(conv3(reflpad1(x))))))) (ResnetBlock, TSNet.py:15-49)."""
    y = _conv(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[prefix + "conv_block.1.weight"], sd[prefix + "conv_block.1.bias"])
    y = F.relu(_in(y))
    y = _conv(F.pad(y, (1, 1, 1, 1), mode="reflect"), sd[prefix + "conv_block.5.weight"], sd[prefix + "conv_block.5.bias"])
    return x + _in(y)


def encoder(x: torch.Tensor, sd: Dict[str, torch.Tensor], net: str, cfg: TSNetConfig, n_blocks: int) -> torch.Tensor:
    """Encoder.forward (TSNet.py:88-105), non-debug Sequential layout (TSNet.py:65-86):
    idx 0-3 stem [pad3, conv7, IN, ReLU]; 3 per downsample [conv3 s2 p1, IN, ReLU];
    then one index per ResnetBlock."""
    if cfg.addcoords:
        x = coord_conv(x)
    p = net + ".model."
    x = _conv(F.pad(x, (3, 3, 3, 3), mode="reflect"), sd[p + "1.weight"], sd[p + "1.bias"])
    x = F.relu(_in_s(x) if cfg.n_downsampling > 0 else _in(x))      # bf16 storage: the stem's and all but the last down-convolution's outputs
    idx = 4
    for l in range(cfg.n_downsampling):
        x = _conv(x, sd[p + f"{idx}.weight"], sd[p + f"{idx}.bias"], stride=2, padding=1)
        x = F.relu(_in_s(x) if l + 1 < cfg.n_downsampling else _in(x))
        idx += 3
    for _ in range(n_blocks):
        x = resnet_block(x, sd, p + f"{idx}.")
        idx += 1
    return x


def fuse_net(src_fea: torch.Tensor, tar_fea: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: TSNetConfig, head: bool = True) -> torch.Tensor:
    """FuseNet.forward (TSNet.py:195-200): cat -> ResnetBlock(s) -> 1x1 conv (head=False: stop before the 1x1 conv)."""
    x = torch.cat((src_fea, tar_fea), dim=1)
    for i in range(cfg.fuse_blocks):
        x = resnet_block(x, sd, f"fuse_net.model.{i}.")
    if not head:
        return x
    return F.conv2d(x, sd["fuse_net.conv.weight"], sd["fuse_net.conv.bias"])


def decoder(prop: torch.Tensor, syn: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: TSNetConfig, stages: Optional[dict] = None):
    """Decoder.forward with return_fea=True (TSNet.py:162-171; ctor :136-155):
    map_conv(cat) -> model0..model{n_blocks-1} ResnetBlocks -> 3x [up x2 bilinear,
    reflpad1, conv3, IN, ReLU] -> [reflpad3, conv7, tanh]."""
    x = _conv(torch.cat([prop, syn], dim=1), sd["dec.map_conv.weight"], sd["dec.map_conv.bias"])
    if stages is not None:
        stages["dec_map"] = x
    n = 0
    for _ in range(cfg.n_blocks):
        x = resnet_block(x, sd, f"dec.model{n}.0.")
        n += 1
    for i in range(cfg.n_downsampling):
        x = _s(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False))     # bf16 storage: the upsampled input ...
        x = _conv(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[f"dec.model{n}.2.weight"], sd[f"dec.model{n}.2.bias"])
        x = F.relu(_in_s(x))                                                                # ... and the up-convolution's output
        if stages is not None:
            stages[f"dec_up{i}"] = x
        n += 1
    fea = x
    x = F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), sd[f"dec.model{n}.1.weight"], sd[f"dec.model{n}.1.bias"])
    return torch.tanh(x), fea


def get_grid(b: int, H: int, W: int) -> torch.Tensor:
    """TSNet.get_grid(normalize=True) (TSNet.py:299-307): (b,H,W,2) of (x,y) in [-1,1]."""
    hr = torch.linspace(-1, 1, H)
    wr = torch.linspace(-1, 1, W)
    gy, gx = torch.meshgrid(hr, wr, indexing="ij")
    return torch.stack((gx, gy), dim=-1).unsqueeze(0).repeat(b, 1, 1, 1).float()


def transformation_branch(tar_fea: torch.Tensor, src_fea: torch.Tensor, tar_bbox: torch.Tensor, src_bbox: torch.Tensor):
    """One source iteration of the propagation branch (TSNet.py:319-323, 339-366).

    tar_fea/src_fea: (b,c,h,w); tar_bbox/src_bbox: (b,1,H,W).  Returns
    (warped (b,c,h,w), flow (b,h,w,2))."""
    b, c, h, w = tar_fea.shape
    t = F.normalize(tar_fea, p=2, dim=1).view(b, c, h * w).transpose(1, 2)           # :319-320
    mt = F.interpolate(tar_bbox, (h, w), mode="nearest").view(b, 1, h * w).transpose(1, 2)  # :322-323
    s = F.normalize(src_fea, p=2, dim=1).view(b, c, h * w)                            # :339-341
    ms = F.interpolate(src_bbox, (h, w), mode="nearest").view(b, 1, h * w)            # :347-348
    corr = torch.bmm(t * mt, s * ms) + torch.bmm(t * (1.0 - mt), s * (1.0 - ms))      # :350-358
    att = F.softmax(100 * corr, dim=2)                                                # :359
    flow = torch.matmul(att, get_grid(b, h, w).to(att.dtype).view(b, h * w, 2)).view(b, h, w, 2)  # :362-365
    warped = F.grid_sample(src_fea, flow, align_corners=False)                        # :366
    return warped, flow


def pose_composite(rec: torch.Tensor, cfg: TSNetConfig) -> torch.Tensor:
    """TSNet_pose fixed-background composite (TSNet_pose.py:276-280, 416-417):
    rec*fore + (-mean/255)*(1-fore), fore = columns 64:192 of a 256x256 frame."""
    mean = torch.tensor(cfg.mean, dtype=torch.float32)
    mask_img = ((-mean).view(1, 3, 1, 1).repeat(1, 1, 256, 256) / 255.0).to(rec.dtype)
    fore = torch.zeros((256, 256), dtype=torch.float32)
    fore[:, 64:192] = 1
    fore = fore.view(1, 1, 256, 256)
    return rec * fore + mask_img * (1 - fore)


# --------------------------------------------------------------------------- forward
@torch.no_grad()
def train_extras(src_img: List[torch.Tensor], tar_img: torch.Tensor, flows: List[torch.Tensor],
                 pg: torch.Tensor, sg: torch.Tensor, cfg: "TSNetConfig" = None) -> dict:
    """The forward's training-mode extras, TSNet.py:327-331, 372-390, 402-405.
    src_img / tar_img are already divided by 255 (set_train_input, :267,279).  Per source: the image is cut into
    down x down patches (F.unfold), the patch grid is warped with the source's flow (F.grid_sample), folded back,
    re-normalised to the target image's per-channel mean / unbiased std, and compared with the target (10 * L1).
    loss_align = 1 - mean cosine similarity of the propagated and the synthesised features.
    (The reference folds to a hard-coded 256 (:379); output_size is the image size here, identical at 256.)
    Pose model (cfg.pose, TSNet_pose.py:343-346, 386-404): the same warp + re-normalisation, then the fixed-background composite of
    the warped image BEFORE the L1 (:399-400), and no loss_align (returned as None)."""
    b, _, h, w = pg.shape
    ref_mean = tar_img.view(b, 3, -1).mean(dim=2).view(b, 3, 1, 1)             # :329
    ref_std = tar_img.view(b, 3, -1).std(dim=2).view(b, 3, 1, 1)               # :330
    warp_list, loss_list = [], []
    for i in range(len(src_img)):
        _, _, ori_h, ori_w = src_img[i].size()                                 # :373
        down = ori_h // h
        src_img_down = F.unfold(src_img[i], down, stride=down)                 # :375
        src_img_down_reshape = src_img_down.view(b, -1, h, w)
        pg_down = F.grid_sample(src_img_down_reshape, flows[i], align_corners=False)   # :377
        warp_src_img = F.fold(pg_down.view(b, -1, h * w), (ori_h, ori_w), down, stride=down)   # :379
        gen_mean = warp_src_img.view(b, 3, -1).mean(dim=2).view(b, 3, 1, 1)    # :381
        gen_std = warp_src_img.view(b, 3, -1).std(dim=2).view(b, 3, 1, 1)
        norm_warp_src_img = (warp_src_img - gen_mean) / gen_std
        warp_src_img = norm_warp_src_img * ref_std + ref_mean                  # :384
        if cfg is not None and cfg.pose and cfg.use_mask:
            warp_src_img = pose_composite(warp_src_img, cfg)                   # TSNet_pose.py:399-400
        warp_list.append(warp_src_img)
        loss_list.append(10 * F.l1_loss(warp_src_img, tar_img))                # :386
    loss_warp = sum(loss_list)                                                 # :390
    if cfg is not None and cfg.pose:
        loss_align = None                                                      # TSNet_pose.py has no alignment loss
    else:
        loss_align = 1 - (F.cosine_similarity(pg, sg, dim=1)).mean()           # :403-405
    return {"warp_src_img_list": warp_list, "loss_warp": loss_warp, "loss_align": loss_align}


def tsnet_forward(sd: Dict[str, torch.Tensor], cfg: TSNetConfig,
                  src_img_list: List[torch.Tensor], src_lbl_list: List[torch.Tensor], src_bbox_list: List[torch.Tensor],
                  tar_lbl: torch.Tensor, tar_bbox: torch.Tensor, want_stages: bool = False,
                  tar_img: torch.Tensor = None, round_operands: Optional[str] = None) -> dict:
    """set_test_input + forward of the reference (TSNet.py:283-294, 309-407).

    Inputs exactly as the reference's callers pass them: images (B,3,H,W) *before* the
    /255 of set_test_input (:286), labels (B,L,H,W), bboxes (B,H,W).
    Returns {'rec_tar_img': (B,3,H,W), 'flows': K x (B,h,w,2), [stage tensors]}.

    round_operands="bf16" is NOT the reference: it is the checker of the engine's bf16-operand mode (BASELINE.json configs[2] / [4]).
    Every convolution input and weight is rounded to bfloat16 where the engine rounds it -- all convolutions except the RGB head, and
    `fuse_net.conv` (1x1, linear) applied to the mean over sources like the engine does -- while products, sums, InstanceNorm, the
    transformation branch and the head stay fp32.  What remains between the two is summation order and the bf16 roundings it flips."""
    global _ROUND_BF16, _STORE_BF16
    if round_operands not in (None, "bf16", "bf16s"):
        raise ValueError("round_operands: None, 'bf16' or 'bf16s' (bf16 operands + bf16 storage of the large activations: operand_mode 2)")
    _ROUND_BF16 = round_operands in ("bf16", "bf16s")
    _STORE_BF16 = round_operands == "bf16s"
    try:
        return _forward(sd, cfg, src_img_list, src_lbl_list, src_bbox_list, tar_lbl, tar_bbox, want_stages, tar_img)
    finally:
        _ROUND_BF16 = False
        _STORE_BF16 = False


def _forward(sd, cfg, src_img_list, src_lbl_list, src_bbox_list, tar_lbl, tar_bbox, want_stages, tar_img) -> dict:
    K = cfg.n_source
    src_img = [x / 255.0 for x in src_img_list[:K]]                 # :286
    src_bbox = [x.unsqueeze(1) for x in src_bbox_list[:K]]          # :288
    tbbox = tar_bbox.unsqueeze(1)                                   # :290
    stages: dict = {}
    src_fea = [encoder(torch.cat([src_img[i], src_lbl_list[i]], dim=1), sd, "img_enc", cfg, cfg.enc_blocks) for i in range(K)]  # :311-313
    tar_fea = encoder(tar_lbl, sd, "lbl_enc", cfg, 0)               # :315
    warped, flows = [], []
    for i in range(K):                                              # :336-370
        wpd, fl = transformation_branch(tar_fea, src_fea[i], tbbox, src_bbox[i])
        warped.append(wpd)
        flows.append(fl)
    pg = torch.stack(warped, dim=1).mean(dim=1)                     # :392
    if _ROUND_BF16:     # the engine's order: the linear 1x1 convolution once, on the mean (its input is what gets rounded)
        zbar = torch.stack([fuse_net(src_fea[i], tar_fea, sd, cfg, head=False) for i in range(K)], dim=1).mean(dim=1)
        sg = _conv(zbar, sd["fuse_net.conv.weight"], sd["fuse_net.conv.bias"])
    else:
        sg = torch.stack([fuse_net(src_fea[i], tar_fea, sd, cfg) for i in range(K)], dim=1).mean(dim=1)  # :396-400
    rec, fea = decoder(pg, sg, sd, cfg, stages if want_stages else None)  # :407
    if cfg.pose and cfg.use_mask:
        rec = pose_composite(rec, cfg)                              # TSNet_pose.py:416-417
    out = {"rec_tar_img": rec, "flows": flows}
    if tar_img is not None:                                         # set_train_input + is_train branches of forward
        out["train"] = train_extras(src_img, tar_img / 255.0, flows, pg, sg, cfg)
    if want_stages:
        stages.update({"src_fea": src_fea, "tar_fea": tar_fea, "pg": pg, "sg": sg, "dec_fea": fea})
        out["stages"] = stages
    return out


# --------------------------------------------------------------------------- synthetic weights / inputs
def conv_shapes(cfg: TSNetConfig) -> Dict[str, tuple]:
    """Every parameter of the four generator nets with its OIHW shape, keyed like the
    reference checkpoints (ctor lines TSNet.py:65-77,139-152,187-193)."""
    shp: Dict[str, tuple] = {}

    def conv(key, co, ci, k):
        shp[key + ".weight"] = (co, ci, k, k)
        shp[key + ".bias"] = (co,)

    def enc(net, cin, nb):
        if cfg.addcoords:
            cin += 3
        conv(f"{net}.model.1", cfg.ngf, cin, 7)
        idx, ch = 4, cfg.ngf
        for _ in range(cfg.n_downsampling):
            conv(f"{net}.model.{idx}", ch * 2, ch, 3)
            ch *= 2
            idx += 3
        for _ in range(nb):
            conv(f"{net}.model.{idx}.conv_block.1", ch, ch, 3)
            conv(f"{net}.model.{idx}.conv_block.5", ch, ch, 3)
            idx += 1

    enc("img_enc", 3 + cfg.label_nc, cfg.enc_blocks)
    enc("lbl_enc", cfg.label_nc, 0)
    fc = cfg.fuse_ngf
    for i in range(cfg.fuse_blocks):
        conv(f"fuse_net.model.{i}.conv_block.1", fc, fc, 3)
        conv(f"fuse_net.model.{i}.conv_block.5", fc, fc, 3)
    conv("fuse_net.conv", fc // 2, fc, 1)
    c = cfg.feat_ch
    conv("dec.map_conv", c, 2 * c, 1)
    n = 0
    for _ in range(cfg.n_blocks):
        conv(f"dec.model{n}.0.conv_block.1", c, c, 3)
        conv(f"dec.model{n}.0.conv_block.5", c, c, 3)
        n += 1
    for i in range(cfg.n_downsampling):
        ci = cfg.ngf * 2 ** (cfg.n_downsampling - i)
        conv(f"dec.model{n}.2", ci // 2, ci, 3)
        n += 1
    conv(f"dec.model{n}.1", 3, cfg.ngf, 7)
    return shp


def synth_state_dict(cfg: TSNetConfig, seed: int = 0, bias_std: float = 0.0) -> Dict[str, torch.Tensor]:
    """PRNG weights with init_net's moments: conv weights ~N(0,0.02), biases 0
    (networks.py:82,92).  bias_std>0 gives non-zero biases (trained-checkpoint-like)
    so bias handling is actually exercised by the parity tests."""
    from wacv23_tsnet_amd import prng
    sd = {}
    for k, s in conv_shapes(cfg).items():
        if k.endswith(".weight"):
            sd[k] = prng.normal(seed, k, s, 0.02)
        elif bias_std > 0:
            sd[k] = prng.normal(seed, k, s, bias_std)
        else:
            sd[k] = torch.zeros(s, dtype=torch.float32)
    return sd


def synth_inputs(cfg: TSNetConfig, B: int, H: int, W: int, seed: int = 1, mask_mode: str = "bernoulli"):
    """quick_start1.py:12-29 input recipe on the PRNG: images U[0,1), labels and bboxes
    Bernoulli(0.5).  mask_mode: 'bernoulli' | 'ones' | 'zeros' | 'soft' (non-binary) | 'box'."""
    from wacv23_tsnet_amd import prng

    def mask(name):
        if mask_mode == "ones":
            return torch.ones((B, H, W))
        if mask_mode == "zeros":
            return torch.zeros((B, H, W))
        if mask_mode == "soft":
            return prng.uniform01(seed, name, (B, H, W))
        if mask_mode == "box":
            m = torch.zeros((B, H, W))
            m[:, H // 4: 3 * H // 4, W // 8: 5 * W // 8] = 1
            return m
        return prng.bernoulli(seed, name, (B, H, W))

    K = cfg.n_source
    src_img = [prng.uniform01(seed, f"src_img.{i}", (B, 3, H, W)) for i in range(K)]
    src_lbl = [prng.bernoulli(seed, f"src_lbl.{i}", (B, cfg.label_nc, H, W)) for i in range(K)]
    src_bbox = [mask(f"src_bbox.{i}") for i in range(K)]
    tar_lbl = prng.bernoulli(seed, "tar_lbl", (B, cfg.label_nc, H, W))
    tar_bbox = mask("tar_bbox")
    return src_img, src_lbl, src_bbox, tar_lbl, tar_bbox
