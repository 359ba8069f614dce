#!/usr/bin/env python
"""Headline benchmark: retargeted frames/s of the TS-Net forward at bs=4 per GPU, 256x256, n_source=3.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one forward (tsnet_forward through the C ABI) over one synthetic batch of B=4
(source-set, driving-frame) pairs per GPU, inputs resident in HBM, fp32.  Workload = BASELINE.json
configs[1]: TSNet(label_nc=2, n_downsampling=3, n_source=3, n_blocks=0), random-init N(0,0.02) weights.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU = 4
H = W = 256
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA
# The conv kernel evaluates every fp32 product as 6 bf16 MFMA products (3-way operand split, conv_x3.hpp), so the
# ceiling of the method in algorithmic (fp32) FLOPs is the bf16 peak / 6.
PEAK_X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
PMC_TRAFFIC_BYTES_RES_CONV = int((320.96 + 27.70) * 2**20)   # profiles/round1_pmc_summary.txt, conv_x3q<64,2,2>, per launch


def _usable_cores() -> int:
    """Cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU box exposes 256
    logical CPUs but a 16-CPU cgroup quota; oversubscribing oneDNN with 256 threads is 80x slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus N>1 must be launched through torch.distributed.run with N ranks")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from oracle import tsnet_oracle as O            # cpu_baseline leg + parity check only
    from wacv23_tsnet_amd.dist import build_replica
    from wacv23_tsnet_amd.engine import TSNetEngine

    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3)
    eng = TSNetEngine(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3, height=H, width=W, max_batch=B_PER_GPU)
    sd = O.synth_state_dict(cfg, seed=0) if rank == 0 else None
    build_replica(eng, sd, dev, src=0)

    # every rank gets its own batch of independent pairs (weak scaling: B fixed per GPU)
    inputs_cpu = O.synth_inputs(cfg, B_PER_GPU, H, W, seed=1 + rank)
    src_img, src_lbl, src_bbox, tar_lbl, tar_bbox = [[t.to(dev) for t in x] if isinstance(x, list) else x.to(dev) for x in inputs_cpu]

    def step():
        return eng.forward(src_img, src_lbl, src_bbox, tar_lbl, tar_bbox)[0]

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- roofline of the dominant kernel class (conv_igemm on the fp32 MFMA): extra forwards with
    # hipEvent brackets around every launch on the engine's stream, after the timed region.
    roofline = None
    if rank == 0:
        eng.timing_enable(True)
        nprobe = 3
        for _ in range(nprobe):
            step()
        torch.cuda.synchronize()
        tm = eng.timing_read(reset=True)
        eng.timing_enable(False)
        P, C, K = eng.h * eng.w, eng.C, eng.K
        total_macs = eng.forward_macs(B_PER_GPU)
        corr_macs = B_PER_GPU * K * P * P * (C + 2)
        conv_flops = 2.0 * (total_macs - corr_macs)
        res_ms, res_launches = tm.get("conv_res", (0.0, 0))
        conv_ms, conv_launches = tm["conv"][0] + res_ms, tm["conv"][1] + res_launches
        conv_tf = conv_flops * nprobe / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        x3 = os.environ.get("TSNET_X3", "1") != "0"
        peak = PEAK_X3_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
        class_ms = {k: round(v[0] / nprobe, 3) for k, v in tm.items() if v[1]}
        if x3 and res_launches:
            # dominant kernel: the 3x3 convolution of the encoder's residual blocks (2 per block; conv_x3p.hpp x3q tiles):
            # M = K*B*h*w output positions, Cin = Cout = C, 9 taps -- SURVEY.md section 8-d: 2*M*C*9C flop per launch
            flop_per_launch = 2.0 * (K * B_PER_GPU * P) * C * (9 * C)
            avg_ms = res_ms / res_launches
            achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12
            roofline = {"bound": "mfma",
                        "kernel": "conv_x3q_kernel<64,2,2> (3x3 ResnetBlock convolution, %d launches per forward = %.0f %% of the forward)"
                                  % (res_launches // nprobe, 100.0 * res_ms / nprobe / (dt * 1e3 / args.steps)),
                        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4),
                        # HBM-side bytes per launch of this kernel from the committed PMC passes (profiles/round1_pmc_summary.txt:
                        # FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); bench.py cannot run rocprofv3 on itself
                        "traffic": PMC_TRAFFIC_BYTES_RES_CONV,
                        "peak_basis": "2500 TF dense bf16 MFMA / 6 bf16 products per fp32 product (3-way operand split)",
                        "frac_of_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                        "algorithmic_gflop_per_launch": round(flop_per_launch / 1e9, 2),
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_forward": res_launches // nprobe,
                        "all_conv_launches": {"achieved": round(conv_tf, 2), "frac": round(conv_tf / peak, 4),
                                              "launches_per_forward": conv_launches // nprobe,
                                              "algorithmic_gflop": round(conv_flops / 1e9, 2)},
                        "class_ms_per_forward": class_ms}
        else:
            roofline = {"bound": "mfma",
                        "kernel": ("conv_x3_kernel" if x3 else "conv_dma_kernel") + " (all conv launches of one forward)",
                        "achieved": round(conv_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(conv_tf / peak, 4), "traffic": None,
                        "peak_basis": ("2500 TF dense bf16 MFMA / 6 bf16 products per fp32 product (3-way split)" if x3
                                       else "157.3 TF exact-fp32 MFMA"),
                        "frac_of_fp32_mfma_peak": round(conv_tf / PEAK_FP32_MFMA_TFLOPS, 4),
                        "algorithmic_gflop_per_launch_set": round(conv_flops / 1e9, 2),
                        "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4), "launches_per_forward": conv_launches // nprobe,
                        "class_ms_per_forward": class_ms}

    # ---- CPU baseline beside it (rank 0, N=1 only): the oracle on the host cores, same workload
    cpu_baseline, max_abs_delta = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd0 = sd
        torch.set_num_threads(_usable_cores())
        ref = O.tsnet_forward(sd0, cfg, *inputs_cpu)            # warm-up; also the parity reference
        max_abs_delta = float((out.cpu() - ref["rec_tar_img"]).abs().max())
        times, budget = [], 25.0
        while len(times) < 3 and sum(times) < budget:
            t1 = time.perf_counter()
            O.tsnet_forward(sd0, cfg, *inputs_cpu)
            times.append(time.perf_counter() - t1)
        med = sorted(times)[len(times) // 2]
        cpu_baseline = {"value": round(B_PER_GPU / med, 4), "unit": "frames/s", "cores": torch.get_num_threads(),
                        "kind": "port", "sample": f"{len(times)} timed forwards (median) of the same B=4, K=3, 256x256 workload after 1 warm-up; torch {torch.__version__} CPU oneDNN"}

    if rank == 0:
        frames = world * B_PER_GPU * args.steps
        gflop_frame = 2.0 * eng.forward_macs(1) / 1e9
        line = {
            "metric": "retargeted frames/sec at bs=4, 256x256, n_source=3; max-abs delta vs ref",
            "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (convs: 3 x bf16 operand split on the bf16 MFMA, fp32 accumulate; fp32-class accuracy)"
            if os.environ.get("TSNET_X3", "1") != "0" else "f32", "data": "synthetic",
            "config": {"workload": "TSNet(label_nc=2,n_blocks=0,n_downsampling=3,n_source=3) forward, fp32, B=4 per GPU, 256x256 (BASELINE.json configs[1])",
                       "global_batch": world * B_PER_GPU, "parallelism": f"replicas x{world} (batch-sharded, weights broadcast once)"},
            "max_abs_delta_vs_oracle": max_abs_delta,
            "algorithmic_gflop_per_frame": round(gflop_frame, 3),
            "whole_forward_frac_of_fp32_mfma_peak": round(frames / dt * gflop_frame / 1e3 / world / PEAK_FP32_MFMA_TFLOPS, 4),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
