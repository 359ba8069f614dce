"""Synthetic weights and inputs for benchmarks and tools (no dataset or checkpoint is reachable from this image).

Weights follow the reference's init (networks.init_net / init_weights, networks.py:82,92: conv weights ~ N(0, 0.02), biases 0),
inputs quick_start1.py:12-29 (images U[0,1), labels and bounding boxes Bernoulli(0.5)), both from the integer-only counter PRNG of
wacv23_tsnet_amd.prng under the same tensor names oracle/tsnet_oracle.py uses -- the two generate identical tensors."""
from __future__ import annotations

from typing import Dict

import torch

from . import prng


def state_dict(param_shapes: Dict[str, tuple], seed: int = 0, bias_std: float = 0.0) -> Dict[str, torch.Tensor]:
    """param_shapes: TSNetEngine.param_shapes() ('<net>.<key>' -> shape, the reference's checkpoint schema)."""
    sd = {}
    for k, s in param_shapes.items():
        if k.endswith(".weight"):
            sd[k] = prng.normal(seed, k, tuple(s), 0.02)
        elif bias_std > 0:
            sd[k] = prng.normal(seed, k, tuple(s), bias_std)
        else:
            sd[k] = torch.zeros(tuple(s), dtype=torch.float32)
    return sd


def inputs(n_source: int, label_nc: int, B: int, H: int, W: int, seed: int = 1):
    """(src_img K x (B,3,H,W), src_lbl K x (B,L,H,W), src_bbox K x (B,H,W), tar_lbl (B,L,H,W), tar_bbox (B,H,W))"""
    src_img = [prng.uniform01(seed, f"src_img.{i}", (B, 3, H, W)) for i in range(n_source)]
    src_lbl = [prng.bernoulli(seed, f"src_lbl.{i}", (B, label_nc, H, W)) for i in range(n_source)]
    src_bbox = [prng.bernoulli(seed, f"src_bbox.{i}", (B, H, W)) for i in range(n_source)]
    return src_img, src_lbl, src_bbox, prng.bernoulli(seed, "tar_lbl", (B, label_nc, H, W)), prng.bernoulli(seed, "tar_bbox", (B, H, W))
