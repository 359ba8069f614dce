"""GPU box: socket power and shader clock WHILE one convolution variant runs back to back (tools build).

The ResnetBlock convolution issues 3 fp16 MFMA products per fp32 product; under that load the chip does not hold its 2.4 GHz.  This probe
keeps one variant of tsnet_bench_conv running for a few seconds in a worker thread and samples the driver's own sensors beside it
(hwmon power1_average / power1_input, `amd-smi metric` as a fallback; sclk from pp_dpm_sclk / amd-smi), so the clock the PMC table derives
(GRBM_GUI_ACTIVE / 8 / duration) can be set against the power the board reports.
usage: power_probe.py [seconds per variant]"""
import ctypes as C, glob, json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

lib = _lib.load_tools()
torch.zeros(1, device="cuda")
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def code(tile=0, abl=0, opt=0):
    return tile | (abl << 16) | ((opt & 15) << 24)


def sysfs_sample():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for f in ("power1_average", "power1_input"):
            p = os.path.join(h, f)
            if os.path.exists(p):
                try:
                    out["power_w"] = int(open(p).read()) / 1e6
                except Exception:
                    pass
        p = os.path.join(h, "freq1_input")
        if os.path.exists(p):
            try:
                out["sclk_mhz"] = int(open(p).read()) / 1e6
            except Exception:
                pass
        if out:
            break
    return out


def smi_sample():
    try:
        r = subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=5)
        j = json.loads(r.stdout)
        g = j[0] if isinstance(j, list) else j
        if "gpu_data" in g:
            g = g["gpu_data"][0]
        out = {}
        pw = g.get("power", {})
        for k in ("socket_power", "current_socket_power", "average_socket_power"):
            v = pw.get(k)
            if isinstance(v, dict) and isinstance(v.get("value"), (int, float)):
                out["power_w"] = float(v["value"]); break
        ck = g.get("clock", {})
        gf = [v.get("clk", {}).get("value") for k, v in ck.items() if k.startswith("gfx") and isinstance(v, dict)]
        gf = [x for x in gf if isinstance(x, (int, float))]
        if gf:
            out["sclk_mhz"] = sum(gf) / len(gf); out["sclk_max_mhz"] = max(gf)
        return out
    except Exception as e:
        return {"err": str(e)[:80]}


def run_variant(name, shape, v, nrm):
    N, H, W, Cin, Cout, k, s, p, refl = shape
    stop = [False]; times = []

    def work():
        ms = C.c_float()
        while not stop[0]:
            rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, k, s, p, refl, nrm, v, 200, C.byref(ms), None)
            if rc != 0:
                print("ERR", lib.tsnet_op_last_error().decode()); break
            times.append(ms.value)

    th = threading.Thread(target=work); th.start()
    samples = []
    t0 = time.time()
    time.sleep(0.5)
    while time.time() - t0 < SECS:
        sm = sysfs_sample()
        if "power_w" not in sm or "sclk_mhz" not in sm:
            sm.update({k: v for k, v in smi_sample().items() if k not in sm})
        samples.append(sm)
        time.sleep(0.1)
    stop[0] = True; th.join()
    pw = [s["power_w"] for s in samples if "power_w" in s]
    ck = [s["sclk_mhz"] for s in samples if "sclk_mhz" in s]
    us = sorted(times)[len(times) // 2] * 1e3 if times else float("nan")
    flops = 2.0 * N * H * W * Cout * Cin * k * k
    print(f"{name:34s} {us:7.1f} us {flops/us/1e6:6.1f} TF   power {sum(pw)/max(len(pw),1):7.1f} W (max {max(pw) if pw else 0:.0f}, n={len(pw)})   "
          f"sclk {sum(ck)/max(len(ck),1):6.0f} MHz (n={len(ck)})", flush=True)


print("idle:", sysfs_sample(), smi_sample())
RES = (12, 32, 32, 512, 512, 3, 1, 1, 1)
for nm, v, nrm in (("res 4x64 raw", code(64), 0), ("res 4x64 IN+ReLU (half zeros)", code(64), 1), ("res 4x64 no staging (abl1)", code(64, abl=1), 0),
                   ("res 4x64 weights once (abl2)", code(64, abl=2), 0), ("res 4x64 MFMA+fold only (abl7)", code(64, abl=7), 0),
                   ("res 4x64 no fold (abl8)", code(64, abl=8), 0), ("res 4x128 raw", code(128), 0)):
    run_variant(nm, RES, v, nrm)
run_variant("fuse_c2 4x128 raw", (12, 32, 32, 1024, 1024, 3, 1, 1, 1), code(128), 0)
time.sleep(1.0)
print("idle after:", sysfs_sample(), smi_sample())
