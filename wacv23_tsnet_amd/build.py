"""Build the HIP extension in-tree: wacv23_tsnet_amd/lib/libtsnet_hip.so (gfx950 only).

    python -m wacv23_tsnet_amd.build [--force] [--tools]

hipcc cross-compiles without a GPU.  The translation units (host logic + small kernels, patch convolutions, general convolution) are
compiled in parallel and linked into one shared library.  The .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
UNITS = ("engine.cpp", "conv_h2_launch.cpp", "conv_h2r_launch.cpp", "conv_g64_launch.cpp", "conv_w1_launch.cpp", "flow_p_launch.cpp")
OUT = os.path.join(HERE, "lib", "libtsnet_hip.so")
# the same sources with -DTSNET_TOOLS: the product kernels PLUS the ablation / experiment instantiations ("computes garbage" variants)
# that tools/x3_ablate.py and tools/h2_variants.py time.  Never loaded by the package.
OUT_TOOLS = os.path.join(HERE, "lib", "libtsnet_tools.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
         "-fhip-fp32-correctly-rounded-divide-sqrt",   # IEEE / and sqrt: the /255, F.normalize and softmax divisions
         "-ffp-contract=off",                           # FMAs only where the source asks for them (fmaf)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]            # MFMA C/D in VGPRs: no AGPR<->VGPR copies around the K loop
# per-unit extras.  The patch convolutions fold accumulator registers with plain fp32 adds; the SLP vectoriser pairs them into
# v_pk_add_f32, which beside MFMAs costs more than the two v_add_f32 it replaces (MI355X_MICROARCH.md, "price of one filler")
UNIT_FLAGS = {"conv_h2_launch.cpp": ["-fno-slp-vectorize"], "conv_g64_launch.cpp": ["-fno-slp-vectorize"], "conv_w1_launch.cpp": ["-fno-slp-vectorize"], "flow_p_launch.cpp": ["-fno-slp-vectorize"]}


def _deps():
    return [os.path.join(CSRC, u) for u in UNITS] + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
           [os.path.join(os.path.dirname(HERE), "include", "tsnet_abi.h"), os.path.abspath(__file__)]      # the flags live in this file


def _up_to_date(out: str) -> bool:
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in _deps())


def up_to_date() -> bool:
    return _up_to_date(OUT)


def _build(out: str, extra, verbose: bool) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build the HIP extension")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj_dir = os.path.join(HERE, "lib", "obj_tools" if extra else "obj")
    os.makedirs(obj_dir, exist_ok=True)

    def one(unit):
        obj = os.path.join(obj_dir, unit.replace(".cpp", ".o"))
        cmd = [hipcc] + FLAGS + UNIT_FLAGS.get(unit, []) + extra + ["-c", os.path.join(CSRC, unit), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(one, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_tools(force: bool = False, verbose: bool = True) -> str:
    if not force and _up_to_date(OUT_TOOLS):
        return OUT_TOOLS
    return _build(OUT_TOOLS, ["-DTSNET_TOOLS"], verbose)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    return _build(OUT, [], verbose)


if __name__ == "__main__":
    if "--tools" in sys.argv:
        print(build_tools(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv))
