"""GPU box: where a conv_w1 workgroup's time goes (tools build).  One stamped launch (abl 30: raw input, 31: IN + ReLU input; statistics
on) of the ResnetBlock layer; every wave records s_memrealtime (100 MHz) at: 0 entry, 1 before / 2 after the prologue barrier (V(0), V(1)
complete), then per tile of the workgroup's chunk the end of its K loop and of its epilogue.  Prints the median / p10 / p90 of each phase over the workgroups, per dispatch round, and the
per-CU occupancy picture (how many workgroups a CU ran, when each started relative to the launch's first stamp).
Stamps (16 slots per wave): 0 entry, 1 before / 2 after the prologue barrier, 3 + 3 j end of tile j's K loop (producers: of its item stream),
5 + 3 j end of tile j's epilogue, 14 end of the workgroup, 15 XCC_ID << 32 | HW_ID.
usage: w1_timeline.py [N images = 12] [norm = 0|1] [Cin = 512] [Cout = 512] [H = 32] [chunk = 0 (the launcher's choice) | 1 | 2 | 3]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load_tools()
torch.zeros(1, device="cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
norm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Cin = int(sys.argv[3]) if len(sys.argv) > 3 else 512
Cout = int(sys.argv[4]) if len(sys.argv) > 4 else 512
H = int(sys.argv[5]) if len(sys.argv) > 5 else 32
W = H
chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 0
abl = 31 if norm else 30
variant = 32768 | (abl << 16) | chunk
ms = C.c_float()
rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, 3, 1, 1, 1, (1 if norm else 0) | 2, variant, 4, C.byref(ms), None)
if rc != 0:
    raise SystemExit("bench_conv: " + lib.tsnet_op_last_error().decode())
tiles = N * (H // 4) * (W // 32) * (Cout // 64)
WV, SL = 14, 16
buf = np.zeros(1024 * WV * SL, dtype=np.uint64)
lib.tsnet_w1_prof_read.argtypes = [C.c_void_p, C.c_int]
rc = lib.tsnet_w1_prof_read(buf.ctypes.data_as(C.c_void_p), 1024)
if rc != 0:
    raise SystemExit("prof_read %d" % rc)
t = buf.reshape(1024, WV, SL).astype(np.int64)
# how many tiles a workgroup ran: the last non-zero per-tile stamp of workgroup 0 (stamps are overwritten by every launch: same everywhere)
c = max(1, sum(1 for j in range(3) if t[0, 0, 5 + 3 * j] > 0 and t[0, 0, 5 + 3 * j] >= t[0, 0, 0]))
nwg = min(tiles // c, 1024)
t = t[:nwg]
t0 = t[:, :, 0].min()
us = lambda x: x / 100.0            # 100 MHz ticks -> us
print(f"conv_w1 N={N} {H}x{W} {Cin}->{Cout} norm={norm} stats=1 chunk={c}: {ms.value*1e3:.1f} us per launch (4 launches timed), {tiles} tiles in {tiles // c} workgroups ({nwg} stamped)")
cons, prod = t[:, :8, :], t[:, 8:, :]
start = t[:, :, 0].min(axis=1)
end = t[:, :, 14].max(axis=1)
dur = us(end - start)
print(f"workgroup duration: median {np.median(dur):.2f} us  p10 {np.percentile(dur,10):.2f}  p90 {np.percentile(dur,90):.2f}; launch span (first stamp -> last stamp) {us(end.max()-t0):.1f} us")
def ph(name, a):
    a = us(a.reshape(-1).astype(np.float64))
    print(f"   {name:58s} median {np.median(a):7.2f}  p10 {np.percentile(a,10):7.2f}  p90 {np.percentile(a,90):7.2f} us")
print("consumers (waves 0..7):")
ph("entry -> prologue barrier reached (table, first weights)", cons[:, :, 1] - cons[:, :, 0])
ph("waiting at the prologue barrier (V(0), V(1) in production)", cons[:, :, 2] - cons[:, :, 1])
prev = cons[:, :, 2]
for j in range(c):
    ph(f"tile {j}: K loop", cons[:, :, 3 + 3 * j] - prev)
    ph(f"tile {j}: epilogue (exchange, transform, statistics, stores)", cons[:, :, 5 + 3 * j] - cons[:, :, 3 + 3 * j])
    prev = cons[:, :, 5 + 3 * j]
print("producers (waves 8..13):")
ph("entry -> V(0), V(1) written", prod[:, :, 1] - prod[:, :, 0])
ph("at the prologue barrier", prod[:, :, 2] - prod[:, :, 1])
prev = prod[:, :, 2]
for j in range(c):
    ph(f"tile {j}: item stream", prod[:, :, 3 + 3 * j] - prev)
    ph(f"tile {j}: epilogue barriers", prod[:, :, 5 + 3 * j] - prod[:, :, 3 + 3 * j])
    prev = prod[:, :, 5 + 3 * j]
print("time spent waiting at the K loop's barriers, per wave over its whole chunk (who waits for whom):")
ph("consumers", cons[:, :, 12])
ph("producers", prod[:, :, 12])
simd = [int(np.median((t[:, w, 15].astype(np.int64) >> 4) & 3)) if False else None for w in range(WV)]
print("   per wave (median over the workgroups; waves 0..7 consumers = (position, channel half), 8..13 producers):")
print("   " + "  ".join(f"w{w}:{np.median(us(t[:, w, 12].astype(np.float64))):5.1f}" for w in range(WV)))
hw = (t[:, 0, 15] & 0xFFFFFFFF).astype(np.int64)
xcc = (t[:, 0, 15] >> 32).astype(np.int64) & 0xF
key = xcc * 10000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 0xF)
uniq, cnt = np.unique(key, return_counts=True)
print(f"placement: {len(uniq)} distinct (xcc, se, sh, cu) ran {nwg} workgroups; workgroups per CU: min {cnt.min()} max {cnt.max()}")
per_round = {}
for k in uniq:
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(start[idx])]
    for r, i in enumerate(idx):
        per_round.setdefault(r, []).append((us(start[i] - t0), us(end[i] - t0), dur[i]))
        if r > 0:
            per_round.setdefault(("gap", r), []).append(us(start[i] - end[idx[r - 1]]))
for r in sorted(k for k in per_round if isinstance(k, int)):
    a = np.array(per_round[r])
    g = per_round.get(("gap", r))
    print(f"   round {r}: {len(a):4d} workgroups  start median {np.median(a[:,0]):7.2f} us  end median {np.median(a[:,1]):7.2f}  duration median {np.median(a[:,2]):6.2f}" +
          (f"  gap after the CU's previous workgroup: median {np.median(g):5.2f} p90 {np.percentile(g,90):5.2f}" if g else ""))
