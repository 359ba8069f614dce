"""Python owner of one `tsnet_handle` (include/tsnet_abi.h): device buffers in, device buffers out.

PyTorch is plumbing here -- tensors provide device memory and the current HIP stream; every
FLOP of the forward runs in libtsnet_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib

POSE_MEAN = (101.84807705937696, 112.10832843463207, 111.65973036298041)  # reference model/TSNet_pose.py:215


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream_of(t: torch.Tensor) -> Optional[int]:
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class TSNetEngine:
    """One model replica on one device.

    lib: the bound C library.  Product code never passes it (the in-tree HIP library is used and
    its absence is an error); tests pass the CPU *emulation* build of the same sources to check
    host logic without a GPU.
    """

    def __init__(self, *, label_nc: int, n_blocks: int, n_downsampling: int = 3, n_source: int = 3, ngf: int = 64,
                 enc_blocks: int = 9, addcoords: bool = True, pose_composite: bool = False,
                 pose_mean: Sequence[float] = POSE_MEAN, height: int = 256, width: int = 256, max_batch: int = 4,
                 operands: str = "fp32", lib=None):
        self.lib = lib if lib is not None else _lib.load()
        cfg = _lib.TsnetCfg()
        cfg.label_nc, cfg.n_blocks, cfg.n_downsampling, cfg.n_source = label_nc, n_blocks, n_downsampling, n_source
        cfg.ngf, cfg.enc_blocks, cfg.addcoords, cfg.pose_composite = ngf, enc_blocks, int(addcoords), int(pose_composite)
        for i in range(3):
            cfg.pose_mean[i] = float(pose_mean[i])
        cfg.height, cfg.width, cfg.max_batch = height, width, max_batch
        if operands not in ("fp32", "bf16", "bf16s"):
            raise ValueError("operands must be 'fp32' (fp32-class arithmetic), 'bf16' (bf16 conv operands, fp32 accumulate) or 'bf16s' (+ bf16 storage "
                             "of the large activations)")
        cfg.operand_mode = {"fp32": 0, "bf16": 1, "bf16s": 2}[operands]
        self.cfg = cfg
        self.K = n_source
        self.h = height >> n_downsampling
        self.w = width >> n_downsampling
        self.C = ngf << n_downsampling
        self._h = C.c_void_p()
        rc = self.lib.tsnet_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"tsnet_create failed ({rc}): {self.lib.tsnet_last_error(None).decode()}")
        self.finalized = False
        self.device: Optional[torch.device] = None     # fixed at finalize(): every allocation, stream and launch of this handle lives there

    # ------------------------------------------------------------------ helpers
    def _on_device(self, dev=None):
        """Context that makes the engine's GPU the current HIP device for the duration of an ABI call.  The library allocates
        (hipMalloc), creates its side stream / events and launches on the CURRENT device; the caller's current device may be
        another one (model on cuda:1, torch.cuda.current_device() == 0)."""
        d = torch.device(dev) if dev is not None else self.device
        if d is not None and d.type == "cuda":
            return torch.cuda.device(d)
        return _NullCtx()

    def _same_device(self, *tensors):
        for t in tensors:
            if t is not None and self.device is not None and t.device != self.device:
                raise ValueError(f"tensor on {t.device}, engine finalized on {self.device}")

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.tsnet_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            with self._on_device():
                self.lib.tsnet_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def param_shapes(self) -> Dict[str, tuple]:
        n = self.lib.tsnet_num_params(self._h)
        out = {}
        name = C.c_char_p()
        shape = (C.c_int64 * 4)()
        rank = C.c_int()
        for i in range(n):
            self._check(self.lib.tsnet_param_info(self._h, i, C.byref(name), shape, C.byref(rank)), "tsnet_param_info")
            out[name.value.decode()] = tuple(int(shape[j]) for j in range(rank.value))
        return out

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """sd keys: '<net>.<key>' with nets img_enc / lbl_enc / fuse_net / dec (checkpoint schema of
        the reference, train_face.py:350-355).  Extra keys (netD, netDF, ...) are ignored unless strict."""
        expected = self.param_shapes()
        for k, v in sd.items():
            if k not in expected:
                if strict:
                    raise KeyError(f"unexpected parameter {k!r}")
                continue
            t = v.detach().to(torch.float32).contiguous()
            shp = (C.c_int64 * t.dim())(*t.shape)
            if t.is_cuda:      # the cast / copy above is queued on torch's stream; tsnet_load_weights does a blocking hipMemcpy on the null stream
                torch.cuda.current_stream(t.device).synchronize()
            with self._on_device(t.device if t.is_cuda else None):
                self._check(self.lib.tsnet_load_weights(self._h, k.encode(), t.data_ptr(), shp, t.dim()), f"load_weights({k})")
        return self

    def finalize(self, device: Optional[torch.device] = None):
        dev = torch.device(device) if device is not None else None
        if dev is not None and dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        stream = torch.cuda.current_stream(dev).cuda_stream if (dev is not None and dev.type == "cuda") else None
        with self._on_device(dev):
            self._check(self.lib.tsnet_finalize(self._h, stream), "tsnet_finalize")
        self.device = dev
        self.finalized = True
        return self

    def packed_weights(self, device) -> torch.Tensor:
        """Alias the engine's packed weight buffer as a flat uint8 torch tensor (for the RCCL broadcast)."""
        p = C.c_void_p()
        n = C.c_size_t()
        with self._on_device():
            self._check(self.lib.tsnet_packed_weights(self._h, C.byref(p), C.byref(n)), "tsnet_packed_weights")
        return _alias_device_bytes(p.value, n.value, device)

    # ------------------------------------------------------------------ forward
    def _ptr_array(self, ts: Sequence[torch.Tensor]):
        arr = (C.c_void_p * _lib.MAX_SOURCES)()
        for i, t in enumerate(ts):
            arr[i] = t.data_ptr()
        return arr

    @staticmethod
    def _prep(t: torch.Tensor, shape: tuple, name: str) -> torch.Tensor:
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        if t.dtype != torch.float32:
            raise TypeError(f"{name}: expected float32, got {t.dtype}")
        return t if t.is_contiguous() else t.contiguous()

    def forward(self, src_img: List[torch.Tensor], src_lbl: List[torch.Tensor], src_bbox: List[torch.Tensor],
                tar_lbl: torch.Tensor, tar_bbox: torch.Tensor, return_flow: bool = False):
        """tsnet_forward.  All tensors on one device.  Returns (rec (B,3,H,W), flows K x (B,h,w,2) | None)."""
        B = tar_lbl.shape[0]
        H, W, L, K = self.cfg.height, self.cfg.width, self.cfg.label_nc, self.K
        if len(src_img) < K or len(src_lbl) < K or len(src_bbox) < K:
            raise ValueError(f"need {K} sources")
        si = [self._prep(src_img[i], (B, 3, H, W), f"src_img[{i}]") for i in range(K)]
        sl = [self._prep(src_lbl[i], (B, L, H, W), f"src_lbl[{i}]") for i in range(K)]
        sb = [self._prep(src_bbox[i], (B, H, W), f"src_bbox[{i}]") for i in range(K)]
        tl = self._prep(tar_lbl, (B, L, H, W), "tar_lbl")
        tb = self._prep(tar_bbox, (B, H, W), "tar_bbox")
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=tl.device)
        flow = torch.empty((K, B, self.h, self.w, 2), dtype=torch.float32, device=tl.device) if return_flow else None
        self._same_device(*si, *sl, *sb, tl, tb)
        with self._on_device():
            rc = self.lib.tsnet_forward(self._h, self._ptr_array(si), self._ptr_array(sl), self._ptr_array(sb),
                                        tl.data_ptr(), tb.data_ptr(), out.data_ptr(), _ptr(flow), B, _stream_of(tl))
        self._check(rc, "tsnet_forward")
        self._keep = (si, sl, sb, tl, tb)   # keep inputs alive until the stream has consumed them
        return out, ([flow[i] for i in range(K)] if return_flow else None)

    def set_source_divisors(self, divisors: Optional[Sequence[float]] = None):
        """tsnet_set_source_divisors: per-source divisor of the source images on load -- 255 (default: the reference's `/255.0`,
        TSNet.py:267,286) or 1 for `use_prev` sources that are already in [0,1] (TSNet.py:269-276).  None resets all to 255."""
        d = list(divisors or [])
        arr = (C.c_float * max(len(d), 1))(*([float(x) for x in d] or [255.0]))
        with self._on_device():
            self._check(self.lib.tsnet_set_source_divisors(self._h, arr, len(d)), "tsnet_set_source_divisors")

    def set_sources(self, src_img, src_lbl, src_bbox):
        B = src_img[0].shape[0]
        H, W, L, K = self.cfg.height, self.cfg.width, self.cfg.label_nc, self.K
        si = [self._prep(src_img[i], (B, 3, H, W), f"src_img[{i}]") for i in range(K)]
        sl = [self._prep(src_lbl[i], (B, L, H, W), f"src_lbl[{i}]") for i in range(K)]
        sb = [self._prep(src_bbox[i], (B, H, W), f"src_bbox[{i}]") for i in range(K)]
        self._same_device(*si, *sl, *sb)
        with self._on_device():
            rc = self.lib.tsnet_set_sources(self._h, self._ptr_array(si), self._ptr_array(sl), self._ptr_array(sb), B, _stream_of(si[0]))
        self._check(rc, "tsnet_set_sources")
        self._keep_src = (si, sl, sb)

    def forward_target(self, tar_lbl, tar_bbox, return_flow: bool = False):
        B = tar_lbl.shape[0]
        H, W, L, K = self.cfg.height, self.cfg.width, self.cfg.label_nc, self.K
        tl = self._prep(tar_lbl, (B, L, H, W), "tar_lbl")
        tb = self._prep(tar_bbox, (B, H, W), "tar_bbox")
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=tl.device)
        flow = torch.empty((K, B, self.h, self.w, 2), dtype=torch.float32, device=tl.device) if return_flow else None
        self._same_device(tl, tb)
        with self._on_device():
            rc = self.lib.tsnet_forward_target(self._h, tl.data_ptr(), tb.data_ptr(), out.data_ptr(), _ptr(flow), B, _stream_of(tl))
        self._check(rc, "tsnet_forward_target")
        self._keep = (tl, tb)
        return out, ([flow[i] for i in range(K)] if return_flow else None)

    def train_extras(self, src_img: List[torch.Tensor], tar_img: torch.Tensor):
        """tsnet_train_extras: the is_train branches of the reference forward (TSNet.py:327-331, 372-390, 402-405) for the
        forward that was just run.  Returns (warp_src_img_list: K x (B,3,H,W), loss_warp, loss_align) -- the losses as 0-d
        device tensors.  With pose_composite (the pose model, TSNet_pose.py:343-346, 386-404) the warped images carry the
        fixed-background composite and loss_align is None (that model has none)."""
        B = tar_img.shape[0]
        H, W, K = self.cfg.height, self.cfg.width, self.K
        si = [self._prep(src_img[i], (B, 3, H, W), f"src_img[{i}]") for i in range(K)]
        ti = self._prep(tar_img, (B, 3, H, W), "tar_img")
        warp = torch.empty((K, B, 3, H, W), dtype=torch.float32, device=ti.device)
        losses = torch.empty(2, dtype=torch.float32, device=ti.device)
        self._same_device(*si, ti)
        with self._on_device():
            rc = self.lib.tsnet_train_extras(self._h, self._ptr_array(si), ti.data_ptr(), B, warp.data_ptr(), losses.data_ptr(), _stream_of(ti))
        self._check(rc, "tsnet_train_extras")
        self._keep_train = (si, ti)
        return [warp[i] for i in range(K)], losses[0], (None if self.cfg.pose_composite else losses[1])

    def stage(self, name: str, device, shape=None) -> torch.Tensor:
        """Copy of a stage tensor of the last forward, NHWC (see tsnet_stage_ptr); `shape` = trailing (H, W, C)
        for stages that are not at the feature resolution ("dec_up<i>")."""
        p = C.c_void_p()
        n = C.c_size_t()
        self._check(self.lib.tsnet_stage_ptr(self._h, name.encode(), C.byref(p), C.byref(n)), "tsnet_stage_ptr")
        flat = _alias_device_bytes(p.value, n.value * 4, device).view(torch.float32).clone()
        return flat.view(-1, *(shape if shape is not None else (self.h, self.w, self.C)))

    def forward_macs(self, B: int) -> float:
        return float(self.lib.tsnet_forward_macs(self._h, B))

    def timing_enable(self, on: bool):
        self._check(self.lib.tsnet_timing_enable(self._h, int(on)), "tsnet_timing_enable")

    def timing_read(self, reset: bool = True):
        ms = (C.c_double * _lib.TIMING_CLASSES)()
        cnt = (C.c_int64 * _lib.TIMING_CLASSES)()
        self._check(self.lib.tsnet_timing_read(self._h, ms, cnt, int(reset)), "tsnet_timing_read")
        return {n: (ms[i], int(cnt[i])) for i, n in enumerate(_lib.TIMING_NAMES)}


def _alias_device_bytes(ptr: int, nbytes: int, device) -> torch.Tensor:
    """View `nbytes` of memory at `ptr` as a uint8 tensor without copying.

    cuda: through __cuda_array_interface__ (zero-copy, the engine keeps ownership);
    cpu (emulation build in tests): through ctypes."""
    dev = torch.device(device)
    if dev.type == "cuda":
        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        return torch.as_tensor(h, device=dev)
    buf = (C.c_ubyte * nbytes).from_address(ptr)
    return torch.frombuffer(buf, dtype=torch.uint8)
