#!/usr/bin/env python
"""Headline benchmark: retargeted frames/s of the TS-Net forward at bs=4 per GPU, 256x256, n_source=3.

    python bench.py --gpus N --steps K --warmup W [--config cfg1|cfg2|cfg3|cfg4]
    N>1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
    --gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE from the environment) or by itself: without WORLD_SIZE in the environment
    `python bench.py --gpus N` spawns its N ranks (one per GPU, rendezvous on 127.0.0.1) and prints rank 0's line.

A "step" is one forward (tsnet_forward through the C ABI) over one synthetic batch of B=4 (source-set, driving-frame) pairs
per GPU, inputs resident in HBM, fp32-class arithmetic.  Workload = BASELINE.json configs[1]:
TSNet(label_nc=2, n_downsampling=3, n_source=3, n_blocks=0), random-init N(0,0.02) weights.  Rank 0 prints ONE JSON line
(contract in the task statement) with `roofline` and `cpu_baseline` objects.

`--config` (default cfg1 = the headline, untouched) runs the other BASELINE.json configs through the SAME rank logic, so that every one
of them can be measured at N = 1, 2, 4, 8: cfg2 = configs[2] (face-checkpoint shape n_blocks=4, bs=8 per GPU, bf16 operands), cfg3 =
configs[3] (TSNet_pose shape: L=25, n_blocks=4, use_mask composite; bs=32 over 8 GPUs = 4 per GPU, weights broadcast by RCCL;
reference demo/demo_pose.py:120-124, model/TSNet_pose.py:276-280,416-417), cfg4 = configs[4] (512x512, n_source=5, bf16, one pair per
GPU).  Each line names its own workload, dtype, algorithmic GFLOP per frame, a roofline on that configuration's dominant kernel against
the peak of its arithmetic (fp16 x 2 split: 2500 / 3 TF; bf16: 2500 TF) and -- at N = 1 -- a parity figure against the oracle that
computes in the same arithmetic class (fp32 / fp64 for cfg1 and cfg3, the bf16-rounding oracle for cfg2 and cfg4).

The rank logic lives in `run_rank()` so that tests/test_bench_ranks.py can drive the very same code at world size 2 over
gloo with the CPU emulation build of the kernels (no GPU in the authoring container): one weight broadcast at start-up,
no collective inside the timed steps, one MAX all-reduce of the elapsed time.
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
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA (v_mfma_f32_32x32x16_{f16,bf16})
# every convolution evaluates an fp32 product as 3 fp16 MFMA products (2-way operand split of the scaled fp32 value, conv_common.hpp):
# the method's ceiling in algorithmic (fp32) FLOPs is the MFMA peak / 3.
PEAK_H2_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0

_F32 = "f32 (convolutions: fp16 x 2 split of the scaled fp32 operands, 3 exact MFMA products, two-level fp32 accumulate; fp32-class accuracy)"
_BF16 = "bf16 (convolution operands rounded to bf16 where they are read, one MFMA product, fp32 accumulate; transformation branch fp32-class; tsnet_cfg.operand_mode=1)"
# BASELINE.json configs[1..4] as bench workloads.  `pmc`: suffix of the committed PMC table the roofline's `traffic` is quoted from
# (profiles/round<N>_pmc_summary<suffix>.txt); `mean_gate`: the frozen end-to-end MEAN gate of the bf16 mode against the rounding oracle
# (tests/test_gpu_forward.py BF16_GATES -- the maximum is chaotic there, in the reference's own arithmetic too: DESIGN.md section 3.2)
CONFIGS = {
    "cfg1": dict(index=1, model=dict(label_nc=2, n_blocks=0, n_downsampling=3, n_source=3), pose=False, batch=4, size=256, operands="fp32", dtype=_F32, pmc="",
                 metric="retargeted frames/sec at bs=4, 256x256, n_source=3; max-abs delta vs ref",
                 workload="TSNet(label_nc=2,n_blocks=0,n_downsampling=3,n_source=3) forward, fp32, B=4 per GPU, 256x256 (BASELINE.json configs[1])"),
    "cfg2": dict(index=2, model=dict(label_nc=2, n_blocks=4, n_downsampling=3, n_source=3), pose=False, batch=8, size=256, operands="bf16", dtype=_BF16, pmc="_bf16", mean_gate=0.128,
                 metric="retargeted frames/sec at bs=8, 256x256, n_source=3, bf16 (BASELINE.json configs[2]); mean-abs delta vs the bf16-rounding ref",
                 workload="TSNet(label_nc=2,n_blocks=4,n_downsampling=3,n_source=3) forward (FaceForensics checkpoint shape, demo_face.py:30-33), bf16 operands, B=8 per GPU, 256x256 (BASELINE.json configs[2])"),
    "cfg3": dict(index=3, model=dict(label_nc=25, n_blocks=4, n_downsampling=3, n_source=3), pose=True, batch=4, size=256, operands="fp32", dtype=_F32, pmc="_cfg3",
                 metric="retargeted frames/sec at bs=4 per GPU (bs=32 over 8 GPUs), 256x256, n_source=3, TSNet_pose (BASELINE.json configs[3]); max-abs delta vs ref",
                 workload="TSNet_pose(label_nc=25,n_blocks=4,n_downsampling=3,n_source=3,use_mask) forward with the fixed-background composite (demo_pose.py:120-124, TSNet_pose.py:276-280,416-417), fp32, B=4 per GPU, 256x256 (BASELINE.json configs[3])"),
    "cfg4": dict(index=4, model=dict(label_nc=2, n_blocks=0, n_downsampling=3, n_source=5), pose=False, batch=1, size=512, operands="bf16", dtype=_BF16, pmc="_cfg4", mean_gate=0.1415,
                 metric="retargeted frames/sec at bs=1 per GPU, 512x512, n_source=5, bf16 (BASELINE.json configs[4]); mean-abs delta vs the bf16-rounding ref",
                 workload="TSNet(label_nc=2,n_blocks=0,n_downsampling=3,n_source=5) forward, bf16 operands, B=1 per GPU, 512x512 (BASELINE.json configs[4])"),
}


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


def _pmc_traffic_bytes(kernel_prefix: str, suffix: str = ""):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/round*_pmc_summary.txt, newest round first:
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/pmc_table.py).  bench.py cannot run rocprofv3 on itself.  Each table records a
    digest of the kernel sources it was measured on (tools/src_digest.py): a table from another build of the kernels is NOT quoted (None,
    with the reason)."""
    import glob
    import re
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_summary%s.txt" % suffix)),
                    key=lambda f: -int(re.search(r"round(\d+)_", os.path.basename(f)).group(1)))
    if not tables:
        return None, "no PMC table committed for this configuration (profiles/round*_pmc_summary%s.txt)" % suffix
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from src_digest import digest
        here = digest()
    except Exception as e:
        return None, "no source digest (%s)" % type(e).__name__
    why = "kernel not in the PMC table"
    for path in tables:
        lines = open(path).read().splitlines()
        rec = [l.split(":", 1)[1].strip() for l in lines if l.startswith("# source_digest:")]
        if not rec or rec[0] != here:
            if path == tables[0]:
                why = "%s was measured on another build of the kernels (digest %s, this tree %s)" % (os.path.relpath(path, ROOT), rec[0] if rec else "none", here)
            continue
        tot, n = 0.0, 0          # launch-weighted mean over the template variants of the kernel (raw input / fused InstanceNorm + ReLU)
        for line in lines:
            if line.startswith(kernel_prefix):
                f = line.split()
                tot += (float(f[-2]) + float(f[-1])) * int(f[-4])
                n += int(f[-4])
            elif n and not line.strip():
                break                # the byte table ends at the first blank line
        if n:
            return int(tot / n * 2**20), os.path.relpath(path, ROOT)
    return None, why


def run_rank(*, rank: int, world: int, device, steps: int, warmup: int, lib=None, config: str = "cfg1", batch: int | None = None,
             height: int | None = None, width: int | None = None, model_kw=None, cpu_baseline: bool = True, timing_probe: bool = True,
             secondary: bool = False) -> dict | None:
    """Everything one rank does: replica from rank 0's weights (ONE broadcast), its own batch of independent pairs, W warm-up steps,
    K timed steps between barriers, MAX over ranks.  Returns the result line on rank 0, None elsewhere.  `config` picks the BASELINE.json
    workload (CONFIGS); batch / height / width / model_kw override it (test hooks: tiny shapes on the emulation build)."""
    from wacv23_tsnet_amd import synth
    from wacv23_tsnet_amd.dist import build_replica
    from wacv23_tsnet_amd.engine import TSNetEngine

    cf = CONFIGS[config]
    batch = cf["batch"] if batch is None else batch
    height = cf["size"] if height is None else height
    width = cf["size"] if width is None else width
    bf16 = cf["operands"] != "fp32"
    dev = torch.device(device)
    cuda = dev.type == "cuda"
    kw = dict(cf["model"])
    kw.update(model_kw or {})
    eng_kw = {k: v for k, v in kw.items() if k != "fuse_ngf"}
    pose = bool(cf["pose"]) and (height, width) == (256, 256)      # the composite's fixed 64..192 column window is a 256 x 256 statement (TSNet_pose.py:279)
    eng_kw.update(pose_composite=pose, operands=cf["operands"])
    eng = TSNetEngine(height=height, width=width, max_batch=batch, lib=lib, **eng_kw)
    # synthetic weights (the reference's init) and inputs from the package's counter PRNG; nothing under oracle/ is touched before the
    # cpu_baseline leg below
    sd = synth.state_dict(eng.param_shapes(), seed=0) if rank == 0 else None
    build_replica(eng, sd, dev, src=0)

    # every rank gets its own batch of independent pairs (weak scaling: B fixed per GPU)
    inputs_cpu = synth.inputs(kw["n_source"], kw["label_nc"], batch, height, width, seed=1 + rank)
    src_img, src_lbl, src_bbox, tar_lbl, tar_bbox = [[t.to(dev) for t in x] if isinstance(x, list) else x.to(dev) for x in inputs_cpu]

    def step():
        return eng.forward(src_img, src_lbl, src_bbox, tar_lbl, tar_bbox)[0]

    def sync():
        if cuda:
            torch.cuda.synchronize()

    out = None
    for _ in range(warmup):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if cuda else None
    t0 = time.perf_counter()
    for i in range(steps):
        if ev:
            ev[i][0].record()
        out = step()
        if ev:
            ev[i][1].record()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # ---- host side of a step (what one thread pays to enqueue a forward -- the scaling question at 8 GPUs is a host question,
    # the ranks exchange nothing per step).  After the timed region, from an idle device: the wall time of `nq` step() calls BEFORE any
    # synchronisation (the HIP queue holds a few thousand packets: five forwards never fill it), beside the time until the device is done.
    host_probe = None
    if cuda:
        nq = 5
        sync()
        q0 = time.perf_counter()
        for _ in range(nq):
            step()
        q1 = time.perf_counter()
        sync()
        q2 = time.perf_counter()
        host_probe = {"host_enqueue_ms_per_step": round((q1 - q0) / nq * 1e3, 3), "device_ms_per_step_same_probe": round((q2 - q0) / nq * 1e3, 3),
                      "host_threads": 1, "steps": nq,
                      "note": "wall time of the step() calls (Python shell + ctypes + the engine's ~70 hipLaunchKernel) before any synchronisation, one host thread"}
        host_probe["host_enqueue_frac_of_step"] = round(host_probe["host_enqueue_ms_per_step"] / max(host_probe["device_ms_per_step_same_probe"], 1e-9), 3)
        # the one collective of a replica's life: the packed weight buffer.  At world size 1 there is no peer; what can be measured here is the
        # buffer's size and a device-to-device copy of it (the HBM side of a broadcast: read once, written once).
        try:
            wb = eng.packed_weights(dev)
            tmp = torch.empty_like(wb)
            tmp.copy_(wb); sync()
            c0 = time.perf_counter(); tmp.copy_(wb); sync(); c1 = time.perf_counter()
            host_probe["packed_weights_MB"] = round(wb.numel() * wb.element_size() / 1e6, 1)
            host_probe["packed_weights_d2d_copy_ms"] = round((c1 - c0) * 1e3, 3)
            del tmp
        except Exception as e:           # noqa: BLE001  (a probe must never cost the line)
            host_probe["packed_weights_error"] = "%s: %s" % (type(e).__name__, e)
    if rank != 0:
        return None
    step_ms = sorted(a.elapsed_time(b) for a, b in ev) if ev else None

    # ---- roofline of the dominant kernel: extra forwards with hipEvent brackets around every launch on the engine's stream,
    # after the timed region (per-kernel timing serialises the two lanes of the forward, so it is kept out of `value`)
    roofline = None
    if timing_probe:
        eng.timing_enable(True)
        nprobe = 3
        for _ in range(nprobe):
            step()
        sync()
        tm = eng.timing_read(reset=True)
        eng.timing_enable(False)
        P, C, K = eng.h * eng.w, eng.C, eng.K
        total_macs = eng.forward_macs(batch)
        corr_macs = batch * K * P * P * (C + 2)
        conv_flops = 2.0 * (total_macs - corr_macs)
        res_ms, res_launches = tm.get("conv_res", (0.0, 0))
        conv_ms, conv_launches = tm["conv"][0] + res_ms, tm["conv"][1] + res_launches
        conv_tf = conv_flops * nprobe / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        class_ms = {k: round(v[0] / nprobe, 3) for k, v in tm.items() if v[1]}
        if res_launches:
            # dominant kernel: the 3x3 convolution of the ResnetBlocks (2 per block; the encoder's on the K*B source images, the decoder's --
            # configs[2] / [3]: n_blocks = 4 -- on the B driving frames), Cin = Cout = C, 9 taps -- SURVEY.md section 8-d: 2*M*C*9C flop
            # per launch with M = images * h * w.  `achieved` = the algorithmic FLOP of these launches / their summed duration (hipEvents
            # on the engine's stream); the tile shape of the last one is read back from the engine.
            import ctypes
            cnt = (ctypes.c_int64 * 4)()
            eng.lib.tsnet_debug_counters(cnt, 0)
            code = int(cnt[3]) % 10000               # + 20000: the two-K-group tiles of single-frame forwards
            wino = int(cnt[3]) // 10000 == 3         # + 30000: the Winograd-along-x form (conv_w1.hpp)
            pr, bn = code // 1000, code % 1000
            side = pr == 3                           # 3128: the bf16 modes' 4 x 128 tile with its four waves side by side (conv_h2<4,128,1,4,...>)
            pr = 4 if side else pr
            n_enc, n_dec = 2 * int(kw.get("enc_blocks", 9)), 2 * int(kw["n_blocks"])
            flop_class = 2.0 * P * C * (9 * C) * (n_enc * K * batch + n_dec * batch)       # per forward
            flop_per_launch = flop_class / (n_enc + n_dec)
            avg_ms = res_ms / res_launches
            achieved = flop_class * nprobe / (res_ms * 1e-3) / 1e12
            # The ceiling in ALGORITHMIC flop of the kernel that ran: one MFMA product per bf16 product; three fp16 products per fp32 product
            # in the direct form (2500 / 3); the Winograd F(2,3)-along-x form issues them on 2/3 of the products, so ITS ceiling is
            # 2500 / (3 * 2/3) = 1250 TF -- a number the kernel cannot exceed (833 TF was not a ceiling for it).  `frac` is
            # therefore the fraction of the matrix pipe's dense peak the launches keep busy (= mfma_flops_issued_tflops / 2500); the
            # direct-form figure of rounds 3 - 5 stays beside it as frac_of_direct_form_ceiling.
            peak = PEAK_F16_MFMA_TFLOPS if bf16 else (PEAK_H2_TFLOPS * 1.5 if wino else PEAK_H2_TFLOPS)
            kname = "conv_w1_kernel" if wino else "conv_h2_kernel"
            traffic, traffic_src = _pmc_traffic_bytes("conv_w1<" if wino else "conv_h2<%d,%d," % (pr, bn), cf["pmc"]) if cuda else (None, "not a GPU run")
            roofline = {"bound": "mfma",
                        "kernel": "%s<%d rows, %d channels, ...> (3x3 ResnetBlock convolution%s%s, %d launches per forward = %.0f %% of the forward)"
                                  % (kname, pr, bn, ", Winograd F(2,3) along x: 2/3 of the direct form's MFMA products" if wino else "",
                                     (", bf16 operands: one MFMA product" + (", four waves side by side" if side else "")) if bf16 else "",
                                     res_launches // nprobe, 100.0 * res_ms / nprobe / (dt * 1e3 / steps)),
                        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        "traffic": traffic, "traffic_source": traffic_src,
                        "peak_basis": "2500 TF dense bf16 MFMA, one product per bf16 product" if bf16 else
                                      ("2500 TF dense fp16 MFMA / (3 fp16 products per fp32 product x 2/3 of the products in the Winograd F(2,3)-along-x form) = 1250 TF algorithmic" if wino else
                                       "2500 TF dense fp16 MFMA / 3 fp16 products per fp32 product (2-way split of the scaled operand, conv_common.hpp)"),
                        "frac_of_direct_form_ceiling": round(achieved / (PEAK_F16_MFMA_TFLOPS if bf16 else PEAK_H2_TFLOPS), 4),
                        # MFMA work actually issued: 3 products per fp32 product, on 2/3 of the products in the Winograd form; 1 in the bf16 mode
                        "mfma_flops_issued_tflops": round(achieved * (1.0 if bf16 else (2.0 if wino else 3.0)), 1),
                        "frac_of_fp32_mfma_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                        # the strict reading -- algorithmic FLOP / dense 16-bit MFMA peak, no credit for the 3 products each fp32 product costs
                        "frac_of_f16_mfma_peak_algorithmic": round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                        "algorithmic_gflop_per_launch": round(flop_per_launch / 1e9, 2),
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_forward": res_launches // nprobe,
                        "all_conv_launches": {"achieved": round(conv_tf, 2), "launches_per_forward": conv_launches // nprobe,
                                              "algorithmic_gflop": round(conv_flops / 1e9, 2)},
                        "class_ms_per_forward": class_ms}

    # ---- CPU baseline beside it (rank 0, N=1 only): the oracle on the host cores, same workload; and the parity figures of this pair
    cpu_base, max_abs_delta, delta64, parity16 = None, None, None, None
    if world == 1 and cpu_baseline:
        from oracle import tsnet_oracle as O        # the checker, timed on the host cores: the only use of oracle/ in this file
        okw = {k: v for k, v in kw.items() if k in ("label_nc", "n_blocks", "n_downsampling", "n_source", "ngf", "enc_blocks", "fuse_ngf")}
        cfg = O.TSNetConfig(pose=pose, **okw)
        torch.set_num_threads(_usable_cores())
        ref = O.tsnet_forward(sd, cfg, *inputs_cpu)            # warm-up; also the fp32 parity reference (the reference's own CPU arithmetic)
        o_cpu = out.cpu()
        if not bf16:
            max_abs_delta = float((o_cpu - ref["rec_tar_img"]).abs().max())
            # the same forward in fp64 (one pass, not timed): how much of that delta is the fp32 CPU forward's own rounding
            i64 = [[t.double() for t in x] if isinstance(x, list) else x.double() for x in inputs_cpu]
            r64 = O.tsnet_forward({k: v.double() for k, v in sd.items()}, cfg, *i64)["rec_tar_img"]
            delta64 = {"gpu_vs_oracle_fp64": float((o_cpu.double() - r64).abs().max()),
                       "oracle_fp32_vs_fp64": float((ref["rec_tar_img"].double() - r64).abs().max())}
            del i64, r64
        else:
            # bf16-operand mode: the oracle that rounds the same operands.  The end-to-end MAXIMUM is chaotic (softmax(100 corr) amplifies
            # rounding flips without bound, in the reference's own arithmetic too); the MEAN is the gated figure (DESIGN.md section 3.2;
            # stage-wise gates: tests/test_gpu_forward.py)
            r16 = O.tsnet_forward(sd, cfg, *inputs_cpu, round_operands="bf16")["rec_tar_img"]
            parity16 = {"mean_abs_vs_bf16_oracle": float((o_cpu - r16).abs().mean()), "max_abs_vs_bf16_oracle": float((o_cpu - r16).abs().max()),
                        "mean_abs_vs_fp32_oracle": float((o_cpu - ref["rec_tar_img"]).abs().mean()),
                        "oracle_bf16_vs_fp32_mean": float((r16 - ref["rec_tar_img"]).abs().mean()), "mean_gate": cf["mean_gate"]}
            del r16
        times, budget = [], 25.0
        while len(times) < 3 and sum(times) < budget:
            t1 = time.perf_counter()
            O.tsnet_forward(sd, cfg, *inputs_cpu)
            times.append(time.perf_counter() - t1)
        med = sorted(times)[len(times) // 2]
        cpu_base = {"value": round(batch / med, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": f"{len(times)} timed fp32 forwards (median) of the same B={batch}, K={cfg.n_source}, {height}x{width} workload after 1 warm-up; torch {torch.__version__} CPU oneDNN"}

    # ---- secondary figure, never the headline: BASELINE.json configs[2] (face-checkpoint shape n_blocks=4, bs=8) in the bf16-operand mode
    second = None
    if secondary and world == 1 and cuda and config == "cfg1":
        try:                                                       # a secondary figure must never cost the headline line
            eng.close()
            e2 = TSNetEngine(label_nc=2, n_blocks=4, n_downsampling=3, n_source=3, height=height, width=width, max_batch=8, operands="bf16", lib=lib)
            build_replica(e2, synth.state_dict(e2.param_shapes(), seed=0), dev, src=0)
            i2 = [[t.to(dev) for t in x] if isinstance(x, list) else x.to(dev) for x in synth.inputs(3, 2, 8, height, width, seed=3)]
            for _ in range(5):
                e2.forward(*i2)
            sync()
            n2 = 30
            t2 = time.perf_counter()
            for _ in range(n2):
                e2.forward(*i2)
            sync()
            d2 = time.perf_counter() - t2
            g2 = 2.0 * e2.forward_macs(1) / 1e9
            second = {"workload": "BASELINE.json configs[2]: TSNet(n_blocks=4) forward, bs=8, 256x256, n_source=3, bf16 conv operands / fp32 accumulate (tsnet_cfg.operand_mode=1)",
                      "value": round(8 * n2 / d2, 2), "unit": "frames/s", "ms_per_step": round(d2 / n2 * 1e3, 3), "steps": n2,
                      "algorithmic_gflop_per_frame": round(g2, 3),
                      "frac_of_bf16_mfma_peak": round(8 * n2 / d2 * g2 / 1e3 / PEAK_F16_MFMA_TFLOPS, 4),
                      "parity": "own tolerance, tests/test_gpu_forward.py::test_cfg2_bf16_face_checkpoint_shape_b8 (helpers.bf16_mode_report)"}
            e2.close()
            # the same with bf16 STORAGE of the large activations on top (operand_mode 2)
            e3 = TSNetEngine(label_nc=2, n_blocks=4, n_downsampling=3, n_source=3, height=height, width=width, max_batch=8, operands="bf16s", lib=lib)
            build_replica(e3, synth.state_dict(e3.param_shapes(), seed=0), dev, src=0)
            for _ in range(5):
                e3.forward(*i2)
            sync()
            t3 = time.perf_counter()
            for _ in range(n2):
                e3.forward(*i2)
            sync()
            d3 = time.perf_counter() - t3
            second["with_bf16_storage"] = {"value": round(8 * n2 / d3, 2), "unit": "frames/s", "ms_per_step": round(d3 / n2 * 1e3, 3)}
            e3.close()
        except Exception as e:                                     # noqa: BLE001
            second = {"error": "%s: %s" % (type(e).__name__, e)}
        eng = TSNetEngine(height=height, width=width, max_batch=batch, lib=lib, **eng_kw)      # forward_macs below

    frames = world * batch * steps
    gflop_frame = 2.0 * eng.forward_macs(1) / 1e9
    line = {
        "metric": cf["metric"],
        "value": round(frames / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": cf["dtype"], "data": "synthetic",
        "config": {"workload": cf["workload"], "name": config, "baseline_config_index": cf["index"],
                   "global_batch": world * batch, "parallelism": f"replicas x{world} (batch-sharded, weights broadcast once)"},
        "max_abs_delta_vs_oracle": max_abs_delta, "max_abs_delta_fp64": delta64,
        "ms_per_step_hipevent_median": round(step_ms[len(step_ms) // 2], 3) if step_ms else None,
        "ms_per_step_hipevent_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)] if step_ms else None,
        "algorithmic_gflop_per_frame": round(gflop_frame, 3),
        "whole_forward_frac_of_fp32_mfma_peak": round(frames / dt * gflop_frame / 1e3 / world / PEAK_FP32_MFMA_TFLOPS, 4),
        "roofline": roofline, "cpu_baseline": cpu_base, "secondary_bf16_cfg2": second, "host": host_probe,
    }
    if bf16:
        line["parity_bf16"] = parity16
        line["whole_forward_frac_of_bf16_mfma_peak"] = round(frames / dt * gflop_frame / 1e3 / world / PEAK_F16_MFMA_TFLOPS, 4)
    eng.close()
    # parity gates of the bench pair (north_star: <= 1e-3 max-abs against the fp32 reference; and no further from the fp64 result than the
    # fp32 CPU forward itself is, plus 1e-4): a line that fails them is still printed, and the process exits non-zero (main)
    fails = []
    if max_abs_delta is not None and not max_abs_delta <= 1e-3:
        fails.append("max_abs_delta_vs_oracle %.3e > 1e-3" % max_abs_delta)
    if delta64 is not None and not delta64["gpu_vs_oracle_fp64"] <= delta64["oracle_fp32_vs_fp64"] + 1e-4:
        fails.append("gpu_vs_oracle_fp64 %.3e > oracle_fp32_vs_fp64 %.3e + 1e-4" % (delta64["gpu_vs_oracle_fp64"], delta64["oracle_fp32_vs_fp64"]))
    if parity16 is not None and not parity16["mean_abs_vs_bf16_oracle"] <= parity16["mean_gate"]:
        fails.append("mean_abs_vs_bf16_oracle %.3e > %.4f" % (parity16["mean_abs_vs_bf16_oracle"], parity16["mean_gate"]))
    line["parity_gate"] = "ok" if not fails else "; ".join(fails)
    if max_abs_delta is None and parity16 is None:
        line["parity_gate"] = "not run (no cpu_baseline leg)"
    return line


def _rank_main(rank: int, world: int, local_rank: int, args, q=None):
    """One rank of the job (its own process).  `args.device == "cpu"` + `args.lib` is the CPU-tier test hook: gloo backend and the
    emulation build of the kernels on tiny shapes (tests/test_bench_ranks.py); the product path is device "cuda" / backend nccl (= RCCL)."""
    cuda = args.device == "cuda"
    if cuda:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (also under torch.distributed.run, before the HIP runtime starts)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the box shares one 16-CPU cgroup between the ranks: one host thread per rank is all a launch loop needs
        torch.set_num_threads(1)
        if cuda:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    lib, kw = None, {}
    if args.lib:
        import ctypes
        from wacv23_tsnet_amd import _lib
        lib = _lib.bind(ctypes.CDLL(args.lib))
    if args.tiny:       # test hook: a narrow net on 32 x 32 frames (the emulator runs ~1 GFLOP/s); the configuration keeps its label count,
        # decoder blocks, pose composite and operand mode
        tk = dict(n_source=2, ngf=8, enc_blocks=1, fuse_ngf=128)
        if CONFIGS[args.config]["model"]["n_blocks"]:
            tk["n_blocks"] = 1
        kw = dict(batch=1, height=32, width=32, model_kw=tk, timing_probe=False)
    line = run_rank(rank=rank, world=world, device=dev, steps=args.steps, warmup=args.warmup, lib=lib, config=args.config,
                    cpu_baseline=not args.no_cpu_baseline and world == 1, secondary=not args.no_secondary and world == 1, **kw)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        if q is not None:
            q.put(line)
        else:
            print(json.dumps(line), flush=True)
            if str(line.get("parity_gate", "ok")).startswith(("max_abs", "gpu_vs", "mean_abs")):
                sys.exit(3)


def _spawned(local_rank: int, world: int, port: int, args, q):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    sys.path.insert(0, ROOT)
    _rank_main(local_rank, world, local_rank, args, q)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg1", choices=sorted(CONFIGS), help="BASELINE.json workload: cfg1 = configs[1] (the headline, default), cfg2 / cfg3 / cfg4 = configs[2] / [3] / [4]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configs[2] bf16-operand figure")
    ap.add_argument("--device", default="cuda", help=argparse.SUPPRESS)      # test hooks (CPU tier): "cpu" = gloo + --lib = the emulation build
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--tiny", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    if "WORLD_SIZE" in os.environ:
        # started by a launcher (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...): one rank per process
        world = int(os.environ["WORLD_SIZE"])
        if args.gpus != world:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
        _rank_main(int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", "0")), args)
        return None
    if args.gpus <= 1:
        _rank_main(0, 1, 0, args)
        return None
    # --gpus N > 1 without a launcher: this process becomes the launcher -- N ranks of this node, one per GPU, rendezvous on 127.0.0.1
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    line, last_err = None, None
    for attempt in range(3):
        # a free port from the kernel; between close() and the children's bind another process may take it: a failed rendezvous is retried
        # on a fresh port.  Rank 0's line is drained WHILE the ranks run (a put() larger than the pipe buffer would otherwise block its
        # writer for ever behind join()).
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.SimpleQueue()
        pc = mp.start_processes(_spawned, args=(args.gpus, port, args, q), nprocs=args.gpus, join=False, start_method="spawn")
        try:
            done, t_start = False, time.time()
            while not done:
                done = pc.join(timeout=0.5)
                while not q.empty():
                    line = q.get()
                if not done and time.time() - t_start > 1800:           # a wedged rank must not hang the caller for ever
                    for p in pc.processes:
                        p.terminate()
                    raise SystemExit("bench: the ranks did not finish within 30 minutes")
            while not q.empty():
                line = q.get()
            break
        except Exception as e:           # ProcessRaisedException / ProcessExitedException of a rank
            last_err = e
            while not q.empty():         # a later rank failed after rank 0 delivered: the result is kept
                line = q.get()
            if line is not None or "EADDRINUSE" not in str(e) and "address already in use" not in str(e).lower():
                break
    if line is None:
        raise SystemExit(f"bench: the ranks failed ({last_err})")
    if last_err is not None and "EADDRINUSE" not in str(last_err) and "address already in use" not in str(last_err).lower():
        # a rank died after rank 0 had delivered its line: the figure counted that rank's frames -- not a successful N-GPU result
        line["error"] = "a rank failed after rank 0 delivered this line: %s" % str(last_err).splitlines()[0][:300]
        print(json.dumps(line), flush=True)
        sys.exit(4)
    print(json.dumps(line), flush=True)
    if str(line.get("parity_gate", "ok")).startswith(("max_abs", "gpu_vs", "mean_abs")):
        sys.exit(3)
    return line


if __name__ == "__main__":
    main()
