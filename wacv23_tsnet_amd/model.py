"""Drop-in Python surface of the reference generator (model/TSNet.py, model/TSNet_pose.py).

What is kept from the reference: class names, constructor keywords, the sub-module attribute
names (`img_enc`, `lbl_enc`, `fuse_net`, `dec`) and their `state_dict()` keys (so the reference's
checkpoints load with the same four `load_state_dict` calls as demo/demo_face.py:126-129), and
the `set_test_input()` / `set_source_num()` / `forward()` -> `self.rec_tar_img` /
`self.warp_grid2d_list` protocol (TSNet.py:283-297, 309-407).

What is different: the modules only *hold* parameters and describe the graph; `forward()` hands
device pointers to the HIP engine (include/tsnet_abi.h).  Nothing here computes with torch ops, and
there is no fallback: without the HIP library `forward()` raises.

Training (`is_train=True`: discriminators, VGG/GAN losses, optimisers -- TSNet.py:229-255,
409-572) is out of scope for this path and raises NotImplementedError.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from .engine import POSE_MEAN, TSNetEngine


def _init_conv_weights(module: nn.Module, gain: float = 0.02):
    """networks.init_weights('normal') semantics (networks.py:78-92): conv weights ~N(0,gain), bias 0."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, 0.0, gain)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0.0)


class _GraphOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("this module is a parameter container; run the model through TSNet.forward() "
                           "(HIP engine). There is no PyTorch execution path.")


class ResnetBlock(_GraphOnly):
    """x + IN(conv3(reflpad(relu(IN(conv3(reflpad(x))))))); keys conv_block.{1,5}.* (TSNet.py:10-49)."""

    def __init__(self, dim: int):
        super().__init__()
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), nn.InstanceNorm2d(dim), nn.ReLU(True),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), nn.InstanceNorm2d(dim))


class Encoder(_GraphOnly):
    """Stem 7x7 + n_downsampling stride-2 convs + n_blocks ResnetBlocks; keys model.{1,4,7,..}.*,
    model.{idx}.conv_block.{1,5}.* (TSNet.py:52-86)."""

    def __init__(self, input_nc: int, ngf: int = 64, n_downsampling: int = 4, n_blocks: int = 9, addcoords: bool = False):
        super().__init__()
        self.addcoords = addcoords
        self.n_blocks = n_blocks
        cin = input_nc + (3 if addcoords else 0)
        layers: List[nn.Module] = [nn.ReflectionPad2d(3), nn.Conv2d(cin, ngf, 7), nn.InstanceNorm2d(ngf), nn.ReLU(True)]
        ch = ngf
        for _ in range(n_downsampling):
            layers += [nn.Conv2d(ch, ch * 2, 3, stride=2, padding=1), nn.InstanceNorm2d(ch * 2), nn.ReLU(True)]
            ch *= 2
        layers += [ResnetBlock(ch) for _ in range(n_blocks)]
        self.model = nn.Sequential(*layers)


class Decoder(_GraphOnly):
    """map_conv 1x1 + staged model0..modelN (return_fea=True layout); keys map_conv.*,
    model{j}.0.conv_block.{1,5}.*, model{n_blocks+i}.2.*, model{n_blocks+n_down}.1.* (TSNet.py:128-155)."""

    def __init__(self, output_nc: int = 3, ngf: int = 64, n_downsampling: int = 4, n_blocks: int = 0):
        super().__init__()
        c = ngf * 2 ** n_downsampling
        self.n_blocks = n_blocks
        self.map_conv = nn.Conv2d(2 * c, c, 1)
        n = 0
        for _ in range(n_blocks):
            setattr(self, f"model{n}", nn.Sequential(ResnetBlock(c)))
            n += 1
        for i in range(n_downsampling):
            ci = ngf * 2 ** (n_downsampling - i)
            setattr(self, f"model{n}", nn.Sequential(
                nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False), nn.ReflectionPad2d(1),
                nn.Conv2d(ci, ci // 2, 3), nn.InstanceNorm2d(ci // 2), nn.ReLU(True)))
            n += 1
        setattr(self, f"model{n}", nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()))


class FuseNet(_GraphOnly):
    """cat -> ResnetBlock(ngf) -> 1x1 conv to ngf/2; keys model.0.conv_block.{1,5}.*, conv.* (TSNet.py:177-193)."""

    def __init__(self, ngf: int = 1024, n_blocks: int = 1):
        super().__init__()
        if n_blocks != 1:
            raise NotImplementedError("the reference only instantiates FuseNet(n_blocks=1) (TSNet.py:227)")
        self.model = nn.Sequential(*[ResnetBlock(ngf) for _ in range(n_blocks)])
        self.conv = nn.Conv2d(ngf, ngf // 2, 1)


GEN_NETS = ("img_enc", "lbl_enc", "fuse_net", "dec")


class TSNet(nn.Module):
    """Face model (model/TSNet.py:203).  Same keywords as the reference constructor."""

    _pose = False

    def __init__(self, lr=0.0002, beta1=0.5, n_blocks=0, n_source=3,
                 lambda_FML=10.0, lambda_VGG=10.0, lambda_CON=10.0, lambda_GRAD=10.0,
                 is_train=True, getIntermFeat=True, label_nc=5, debug=False, lambda_dec=1.0,
                 addcoords=True, ngf=64, n_downsampling=4, return_flow=False,
                 height=256, width=256, max_batch=None, operands="fp32"):
        super().__init__()
        if operands not in ("fp32", "bf16", "bf16s"):
            raise ValueError("operands must be 'fp32' (default, 1e-3 parity with the fp32 reference), 'bf16' (BASELINE.json configs[2]/[4]: bf16 "
                             "convolution operands) or 'bf16s' (the same + bf16 storage of the large activations)")
        self.operands = operands
        if is_train:
            raise NotImplementedError("training (GAN/VGG losses, optimisers) is outside the MI355X forward path; "
                                      "construct with is_train=False")
        if debug:
            raise NotImplementedError("debug=True changes the state_dict layout and is not supported")
        if ngf * 2 ** n_downsampling * 2 != 1024 and ngf == 64:
            # same constraint the reference has implicitly: FuseNet is hard-wired to 1024 channels (TSNet.py:227)
            raise ValueError("with ngf=64 the reference's FuseNet(ngf=1024) requires n_downsampling=3")
        self.return_flow = return_flow
        self.n_source = n_source
        self.is_train = False
        self.label_nc, self.n_blocks, self.n_downsampling, self.ngf, self.addcoords = label_nc, n_blocks, n_downsampling, ngf, addcoords
        self.height, self.width, self.max_batch = height, width, max_batch
        self.img_enc = Encoder(3 + label_nc, ngf=ngf, n_downsampling=n_downsampling, addcoords=addcoords)
        self.lbl_enc = Encoder(label_nc, ngf=ngf, n_downsampling=n_downsampling, n_blocks=0, addcoords=addcoords)
        self.dec = Decoder(3, ngf=ngf, n_downsampling=n_downsampling, n_blocks=n_blocks)
        self.fuse_net = FuseNet(ngf=2 * ngf * 2 ** n_downsampling, n_blocks=1)
        for net in GEN_NETS:
            _init_conv_weights(getattr(self, net), 0.02)     # networks.init_net (TSNet.py:220-228)
        self.src_img_list = self.src_lbl_list = self.src_bbox_list = None
        self.tar_lbl = self.tar_bbox = None
        self.prev_tar_img = self.prev_tar_lbl = self.prev_tar_bbox = None
        self.rec_tar_img = None
        self.warp_grid2d_list = None
        self.tar_img = None
        self.warp_src_img_list = None
        self.loss_warp = self.loss_align = 0.0
        self._engine: Optional[TSNetEngine] = None
        self._engine_key = None
        self._use_prev = None

    # ------------------------------------------------------------------ reference protocol
    def set_test_input(self, src_img_list, src_lbl_list, src_bbox_list, tar_lbl, tar_bbox,
                       prev_tar_img=None, prev_tar_lbl=None, prev_tar_bbox=None):
        """Same argument meaning as TSNet.set_test_input (TSNet.py:283-294).  The reference moves
        tensors to the GPU and divides images by 255 here; the move happens here too, the /255 is
        fused into the engine's input-packing kernel.  prev_* are stored and unused, as in the reference."""
        dev = self._device()
        mv = lambda t: t.to(dev, dtype=torch.float32, non_blocking=True)
        self.src_img_list = [mv(x) for x in src_img_list]
        self.src_lbl_list = [mv(x) for x in src_lbl_list]
        self.src_bbox_list = [mv(x) for x in src_bbox_list]
        self.tar_lbl = mv(tar_lbl)
        self.tar_bbox = mv(tar_bbox)
        if prev_tar_img is not None:
            self.prev_tar_img, self.prev_tar_lbl, self.prev_tar_bbox = prev_tar_img, prev_tar_lbl, prev_tar_bbox
        # a test-mode input ends any training-mode state: no stale target frame, no per-source divisors
        self.tar_img = None
        self.warp_src_img_list = None
        self.loss_warp = self.loss_align = 0.0
        self._use_prev = None

    def set_train_input(self, src_img_list, src_lbl_list, src_bbox_list, tar_img, tar_lbl, tar_bbox, use_prev=None):
        """TSNet.set_train_input (TSNet.py:266-281): as set_test_input plus the ground-truth target frame.  forward()
        then also fills warp_src_img_list, loss_warp and loss_align -- the training-mode outputs of the FORWARD
        (TSNet.py:327-331, 372-390, 402-405).  The GAN / VGG losses and the backward pass are not part of this build.
        use_prev[i] marks source i as a previously generated frame that is already in [0,1]: it skips the /255 (TSNet.py:269-276)."""
        self.set_test_input(src_img_list, src_lbl_list, src_bbox_list, tar_lbl, tar_bbox)
        self.tar_img = tar_img.to(self._device(), dtype=torch.float32, non_blocking=True)
        self._use_prev = None if use_prev is None else [bool(x) for x in use_prev]

    def set_source_num(self, n_source):
        """TSNet.set_source_num (TSNet.py:296)."""
        self.n_source = n_source

    def forward(self):
        """TSNet.forward (TSNet.py:309-407), test mode: result in self.rec_tar_img (and
        self.warp_grid2d_list when return_flow)."""
        if self.tar_lbl is None:
            raise RuntimeError("call set_test_input() before forward()")
        eng = self._get_engine(self.tar_lbl.shape[0])
        K = self.n_source
        use_prev = getattr(self, "_use_prev", None)
        if use_prev != getattr(eng, "_use_prev_applied", None):
            eng.set_source_divisors(None if use_prev is None else [1.0 if p else 255.0 for p in use_prev[:K]])
            eng._use_prev_applied = use_prev
        rec, flows = eng.forward(self.src_img_list[:K], self.src_lbl_list[:K], self.src_bbox_list[:K],
                                 self.tar_lbl, self.tar_bbox, return_flow=self.return_flow)
        self.rec_tar_img = rec
        if self.return_flow:
            self.warp_grid2d_list = flows
        if self.tar_img is not None:           # set_train_input was used: the forward's training-mode outputs
            self.warp_src_img_list, self.loss_warp, self.loss_align = eng.train_extras(self.src_img_list[:K], self.tar_img)
            if self._pose:
                self.loss_align = None         # TSNet_pose.py has no alignment loss

    # ------------------------------------------------------------------ engine management
    def _device(self):
        return next(self.parameters()).device

    def generator_state_dict(self):
        """{'<net>.<key>': tensor} over the four generator nets (checkpoint schema, train_face.py:350-355)."""
        out = {}
        for net in GEN_NETS:
            for k, v in getattr(self, net).state_dict().items():
                out[f"{net}.{k}"] = v
        return out

    def load_checkpoint(self, ckpt: dict):
        """Accepts the reference's checkpoint dict {'img_enc','lbl_enc','dec','fuse_net'[, 'netD','netDF','example']}
        (train_face.py:350-355; loader calls demo_face.py:126-129)."""
        for net in GEN_NETS:
            getattr(self, net).load_state_dict(ckpt[net])
        self._engine = None

    def _weights_version(self):
        return tuple(p._version for p in self.parameters()) + (str(self._device()),)

    def _new_engine(self, max_batch: int) -> TSNetEngine:
        """a finalized engine with this model's current weights (the caller owns it and closes it)"""
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("TSNet.forward() runs on an MI355X only: move the model with .cuda() first "
                               "(there is no CPU execution path)")
        eng = TSNetEngine(label_nc=self.label_nc, n_blocks=self.n_blocks, n_downsampling=self.n_downsampling,
                          n_source=self.n_source, ngf=self.ngf, addcoords=self.addcoords,
                          pose_composite=self._pose and getattr(self, "use_mask", False),
                          pose_mean=getattr(self, "mean", POSE_MEAN), height=self.height, width=self.width,
                          max_batch=max_batch, operands=self.operands)
        eng.load_state_dict(self.generator_state_dict())
        eng.finalize(dev)
        return eng

    def _get_engine(self, B: int) -> TSNetEngine:
        key = (self.n_source, self._weights_version())
        if self._engine is None or self._engine_key != key or B > self._engine.cfg.max_batch:
            if self._device().type != "cuda":
                raise RuntimeError("TSNet.forward() runs on an MI355X only: move the model with .cuda() first "
                                   "(there is no CPU execution path)")
            if self._engine is not None:
                self._engine.close()
            self._engine, self._engine_key = self._new_engine(max(B, self.max_batch or 0)), key
        return self._engine


class TSNetPose(TSNet):
    """Pose model (model/TSNet_pose.py:205): same generator plus the fixed-background composite
    `rec*fore + (-mean/255)*(1-fore)` (TSNet_pose.py:276-280, 416-417).  No return_flow, as in the reference."""

    _pose = True

    def __init__(self, *args, use_mask=True, mean=np.array(POSE_MEAN, dtype=np.float32), **kw):
        kw.setdefault("is_train", True)
        super().__init__(*args, **kw)
        self.use_mask = use_mask
        self.mean = tuple(float(x) for x in np.asarray(mean, dtype=np.float32))

    # set_train_input / forward are the base class's: with use_mask the engine composites warp_src_img with the fixed background
    # before the L1 and reports no alignment loss (TSNet_pose.py:343-346, 386-404); self.loss_align stays None.
