"""Diagnostic (GPU box): host CPU topology and oracle throughput vs torch thread count."""
import os, sys, time, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))   # repo root
import torch
from oracle import tsnet_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
print(subprocess.run("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|^CPU\\(s\\)'", shell=True, capture_output=True, text=True).stdout)
cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=3)
sd = O.synth_state_dict(cfg, seed=0)
inp = O.synth_inputs(cfg, 4, 256, 256, seed=1)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    O.tsnet_forward(sd, cfg, *inp)
    t = time.time(); O.tsnet_forward(sd, cfg, *inp); dt = time.time() - t
    print(json.dumps(dict(threads=th, sec=dt, fps=4 / dt)), flush=True)
