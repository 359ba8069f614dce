"""bench.py's N>1 branch, executed: world size 2 over gloo, every rank an engine on the CPU emulation build of the kernels, tiny
shapes.  bench.run_rank is the code `python bench.py --gpus N` runs on every rank; here its collectives are logged in order:
exactly ONE broadcast (the packed weight buffer, dist.build_replica), then the warm-up steps, a barrier, the timed steps with
NO collective between them, a barrier, and one MAX all-reduce of the elapsed time.  (No 8-GPU node was available to the build:
the scaling curve itself is the driver's to measure; this pins that the path it will run is the intended one.)"""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARMUP = 3, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, emu_path, out_dir):
    sys.path.insert(0, ROOT)
    import ctypes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from wacv23_tsnet_amd import _lib
    from wacv23_tsnet_amd.engine import TSNetEngine
    log = []
    for name in ("broadcast", "all_reduce", "barrier", "all_gather", "reduce_scatter", "all_to_all", "send", "recv"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o, n: (lambda *a, **k: (log.append(n), o(*a, **k))[1]))(orig, name))
    fwd = TSNetEngine.forward
    TSNetEngine.forward = lambda self, *a, **k: (log.append("step"), fwd(self, *a, **k))[1]
    lib = _lib.bind(ctypes.CDLL(emu_path))
    line = bench.run_rank(rank=rank, world=world, device="cpu", steps=STEPS, warmup=WARMUP, lib=lib, batch=1, height=32, width=32,
                          model_kw=dict(n_source=2, ngf=8, enc_blocks=1, fuse_ngf=128), cpu_baseline=False, timing_probe=False)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"log": log, "line": line}, f)
    dist.destroy_process_group()


def test_two_ranks_one_broadcast_no_collective_per_step(emu_lib, tmp_path):
    from conftest import build_emu_lib
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), build_emu_lib(), str(tmp_path)), nprocs=world, join=True)
    want = ["broadcast"] + ["step"] * WARMUP + ["barrier"] + ["step"] * STEPS + ["barrier", "all_reduce"]
    for r in range(world):
        got = json.load(open(tmp_path / f"rank{r}.json"))
        assert got["log"] == want, (r, got["log"])
        if r == 0:
            line = got["line"]
            assert line["n_gpus"] == 2 and line["steps"] == STEPS and line["warmup"] == WARMUP and line["scaling"] == "weak"
            assert line["config"]["global_batch"] == 2 and line["value"] > 0
            assert abs(line["value"] - 2 * STEPS / (line["ms_per_step"] * STEPS / 1e3)) / line["value"] < 1e-2
        else:
            assert got["line"] is None


def test_bench_main_launches_its_own_ranks(emu_lib, capsys):
    """`python bench.py --gpus 2` with no launcher around it (no WORLD_SIZE in the environment): bench.main() spawns the two ranks itself
    and prints ONE line.  Same code path as on the GPU box, with the hidden test hooks selecting gloo + the emulation build + tiny shapes."""
    sys.path.insert(0, ROOT)
    import bench
    from conftest import build_emu_lib
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        assert k not in os.environ
    line = bench.main(["--gpus", "2", "--steps", "2", "--warmup", "1", "--device", "cpu", "--lib", build_emu_lib(), "--tiny"])
    out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(out) == 1 and json.loads(out[0])["n_gpus"] == 2
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["global_batch"] == 2 and line["value"] > 0
    assert line["cpu_baseline"] is None and line["secondary_bf16_cfg2"] is None          # N > 1: neither leg runs


def test_bench_config_switch_runs_every_baseline_config_multi_rank(emu_lib, capsys):
    """`python bench.py --gpus 2 --config cfgN`: the multi-rank path runs every BASELINE.json workload, not only the headline.
    World size 2 over gloo on the emulation build, tiny shapes; each configuration keeps what defines it -- configs[3] its 25 label channels
    and decoder ResnetBlocks (TSNet_pose, demo_pose.py:120-124), configs[2] / [4] the bf16-operand mode, configs[4] its five sources."""
    sys.path.insert(0, ROOT)
    import bench
    from conftest import build_emu_lib
    want = {"cfg2": (2, "bf16"), "cfg3": (3, "f32"), "cfg4": (4, "bf16")}
    for name, (index, dt) in want.items():
        line = bench.main(["--gpus", "2", "--steps", "1", "--warmup", "1", "--device", "cpu", "--lib", build_emu_lib(), "--tiny", "--config", name])
        out = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
        assert len(out) == 1 and json.loads(out[0])["config"]["name"] == name
        assert line["n_gpus"] == 2 and line["config"]["baseline_config_index"] == index and line["config"]["global_batch"] == 2
        assert line["dtype"].startswith(dt) and "configs[%d]" % index in line["config"]["workload"] and line["value"] > 0
        assert line["scaling"] == "weak" and line["metric"] != bench.CONFIGS["cfg1"]["metric"]


def test_bench_configs_match_baseline_json():
    """The four bench workloads are BASELINE.json's configs[1..4]: label count, decoder blocks, sources, frame size, per-GPU batch, operands."""
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    c = bench.CONFIGS
    assert (c["cfg1"]["batch"], c["cfg1"]["size"], c["cfg1"]["model"]["n_source"], c["cfg1"]["operands"]) == (4, 256, 3, "fp32") and "fp32" in base[1]
    assert (c["cfg2"]["batch"], c["cfg2"]["model"]["n_blocks"], c["cfg2"]["operands"]) == (8, 4, "bf16") and "bs=8 bf16" in base[2]
    assert (c["cfg3"]["batch"] * 8, c["cfg3"]["model"]["label_nc"], c["cfg3"]["pose"], c["cfg3"]["operands"]) == (32, 25, True, "fp32") and "bs=32" in base[3]
    assert (c["cfg4"]["size"], c["cfg4"]["model"]["n_source"], c["cfg4"]["operands"]) == (512, 5, "bf16") and "512" in base[4] and "n_source=5" in base[4]
    for k, v in c.items():
        assert "BASELINE.json configs[%d]" % v["index"] in v["workload"]
