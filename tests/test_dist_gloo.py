"""N>1 path on CPU: world_size-2 gloo processes, each owning a replica of the engine (CPU-emulation build),
weights replicated by ONE broadcast of the packed buffer from rank 0, batch sharded with no data-path
collective.  The gathered result must equal the oracle on the full batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, emu_path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import tsnet_oracle as O
    from wacv23_tsnet_amd import _lib
    from wacv23_tsnet_amd.dist import build_replica, shard_range
    from wacv23_tsnet_amd.engine import TSNetEngine
    lib = _lib.bind(ctypes.CDLL(emu_path))
    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=2, ngf=16, enc_blocks=1, fuse_ngf=256)
    B = 3                                                   # uneven split: rank 0 gets 2 items, rank 1 gets 1
    inp = O.synth_inputs(cfg, B, 32, 32, seed=11, mask_mode="box")
    sd = None
    if rank == 0:                                           # only rank 0 has the checkpoint
        sd = O.synth_state_dict(cfg, seed=5)
        sd = {k: (v * 4 if k.endswith("weight") else v) for k, v in sd.items()}
    eng = TSNetEngine(label_nc=2, n_blocks=0, n_source=2, ngf=16, enc_blocks=1, height=32, width=32, max_batch=2, lib=lib)
    build_replica(eng, sd, "cpu", src=0)
    lo, hi = shard_range(B, rank, world)
    sl = slice(lo, hi)
    rec, _ = eng.forward([x[sl] for x in inp[0]], [x[sl] for x in inp[1]], [x[sl] for x in inp[2]], inp[3][sl], inp[4][sl])
    torch.save((lo, hi, rec), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_match_oracle(emu_lib, tmp_path):
    from conftest import build_emu_lib
    from oracle import tsnet_oracle as O
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, build_emu_lib(), str(tmp_path)), nprocs=world, join=True)
    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=2, ngf=16, enc_blocks=1, fuse_ngf=256)
    sd = O.synth_state_dict(cfg, seed=5)
    sd = {k: (v * 4 if k.endswith("weight") else v) for k, v in sd.items()}
    inp = O.synth_inputs(cfg, 3, 32, 32, seed=11, mask_mode="box")
    ref = O.tsnet_forward(sd, cfg, *inp)["rec_tar_img"]
    got = torch.empty_like(ref)
    covered = 0
    for r in range(world):
        lo, hi, rec = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        got[lo:hi] = rec
        covered += hi - lo
    assert covered == 3
    assert (got - ref).abs().max().item() < 5e-4


def test_shard_range_partitions():
    from wacv23_tsnet_amd.dist import shard_range
    for total in (1, 4, 7, 32):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1
