"""GPU tier, N > 1: the replica path over RCCL.  Skipped on a box with one GPU (the builder's `gpurun` box); lights up wherever the driver
runs `pytest -m gpu` on a multi-GPU node.  The world-size-2 logic itself is covered on CPU (tests/test_dist_gloo.py, test_bench_ranks.py).
The file sorts last on purpose: these two tests have never met multi-GPU hardware (none is available to the build), so under `-x` they
run after every single-GPU test has been counted; the rank processes are joined with a deadline, never waited on for ever."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs")]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher: spawns its ranks, RCCL rendezvous on 127.0.0.1, one JSON line from rank 0."""
    sys.path.insert(0, ROOT)
    import bench
    line = bench.main(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-secondary"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 8
    assert line["value"] > 0 and line["cpu_baseline"] is None


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from wacv23_tsnet_amd import synth
    from wacv23_tsnet_amd.dist import build_replica, shard_range
    from wacv23_tsnet_amd.engine import TSNetEngine
    B = 3                                                        # uneven: rank 0 owns two pairs, rank 1 one
    eng = TSNetEngine(label_nc=2, n_blocks=0, n_downsampling=3, n_source=2, height=64, width=64, max_batch=B)
    sd = synth.state_dict(eng.param_shapes(), seed=7) if rank == 0 else None     # only rank 0 holds the checkpoint
    build_replica(eng, sd, dev, src=0)
    packed = eng.packed_weights(dev)
    checksum = packed.view(torch.int32).to(torch.int64).sum().item()            # the bits, not the values
    inp = synth.inputs(2, 2, B, 64, 64, seed=9)
    on = lambda x, sl: [t[sl].to(dev) for t in x] if isinstance(x, list) else x[sl].to(dev)
    lo, hi = shard_range(B, rank, world)
    rec, _ = eng.forward(*[on(x, slice(lo, hi)) for x in inp])
    out = {"lo": lo, "hi": hi, "rec": rec.cpu(), "checksum": checksum}
    if rank == 0:                                                # the single-GPU forward of the whole batch, same replica
        out["full"] = eng.forward(*[on(x, slice(0, B)) for x in inp])[0].cpu()
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_bitwise(tmp_path):
    """After ONE broadcast both ranks hold the same packed weight bits, and each rank's shard equals the single-GPU forward of the same
    pairs bit for bit (a pair's result does not depend on its batch or its GPU: DESIGN.md sections 3.1, 6)."""
    world, port = 2, _free_port()
    pc = mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=False)
    deadline = 600.0
    import time
    t0 = time.time()
    while not pc.join(timeout=5.0):
        if time.time() - t0 > deadline:
            for p in pc.processes:
                p.terminate()
            pytest.fail("the two ranks did not finish within %d s" % deadline)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt")) for k in range(world)]
    assert r[0]["checksum"] == r[1]["checksum"]
    assert (r[0]["lo"], r[0]["hi"], r[1]["lo"], r[1]["hi"]) == (0, 2, 2, 3)
    full = r[0]["full"]
    assert torch.equal(r[0]["rec"], full[0:2])
    # rank 1 runs ONE pair: the same bits as that pair inside the batch of three (round 5: no batch-dependent tile anywhere in the forward)
    assert torch.equal(r[1]["rec"], full[2:3])
