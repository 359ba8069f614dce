"""Multi-GPU execution of the forward path: independent replicas, one process per GPU.

Every (source-set, driving-frame) pair is independent end to end (InstanceNorm and softmax are
per-sample), so the batch is sharded across ranks with NO data-path
collective.  The only communication is one broadcast of the packed weight buffer at start-up
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  The reference is
single-GPU only -- there is no reference call site for this module.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from .engine import TSNetEngine


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` batch items owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def build_replica(engine: TSNetEngine, state_dict: Optional[Dict[str, torch.Tensor]], device, src: int = 0,
                  group=None) -> TSNetEngine:
    """Finalize `engine` on every rank with rank `src`'s weights.

    Rank `src` passes the checkpoint's state_dict; the others pass None, finalize from zeros (which
    only allocates and lays out the packed buffer) and receive the packed weights by ONE broadcast.
    With world_size == 1 (or torch.distributed uninitialised) this is a plain load + finalize."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    if not distributed or rank == src:
        if state_dict is None:
            raise ValueError("the source rank needs the state_dict")
        engine.load_state_dict(state_dict)
    else:
        engine.load_state_dict({k: torch.zeros(s) for k, s in engine.param_shapes().items()})
    engine.finalize(device)
    if distributed:
        buf = engine.packed_weights(device)          # zero-copy alias of the engine's buffer
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream(device).synchronize()
        dist.broadcast(buf, src=src, group=group)
        if torch.device(device).type == "cuda":
            torch.cuda.current_stream(device).synchronize()
    return engine
