"""Build the HIP extension in-tree: wacv23_tsnet_amd/lib/libtsnet_hip.so (gfx950 only).

    python -m wacv23_tsnet_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "engine.cpp")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("conv_igemm.hpp", "conv_dma.hpp", "conv_x3.hpp", "conv_x3p.hpp", "conv_x3r.hpp", "conv_h2.hpp", "split3.hpp", "head_conv.hpp", "flow_warp.hpp", "norm_elementwise.hpp", "postproc.hpp", "raster.hpp", "train_extras.hpp")] + \
       [os.path.join(os.path.dirname(HERE), "include", "tsnet_abi.h")]
OUT = os.path.join(HERE, "lib", "libtsnet_hip.so")
# the same sources with -DTSNET_TOOLS: the product kernels PLUS the superseded convolution generations and the ablation
# instantiations ("computes garbage" variants) that tools/x3_ablate.py and tools/conv_sweep.py time.  Never loaded by the package.
OUT_TOOLS = os.path.join(HERE, "lib", "libtsnet_tools.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
         "-fhip-fp32-correctly-rounded-divide-sqrt",   # IEEE / and sqrt: the /255, F.normalize and softmax divisions
         "-ffp-contract=off",                           # FMAs only where the source asks for them (fmaf)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]            # MFMA C/D in VGPRs: no AGPR<->VGPR copies around the K loop


def up_to_date() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)


def build_tools(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT_TOOLS) and all(os.path.getmtime(OUT_TOOLS) >= os.path.getmtime(d) for d in DEPS):
        return OUT_TOOLS
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-DTSNET_TOOLS", SRC, "-o", OUT_TOOLS]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT_TOOLS


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build the HIP extension")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc] + FLAGS + [SRC, "-o", OUT]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--tools" in sys.argv:
        print(build_tools(force="--force" in sys.argv))
    else:
        build(force="--force" in sys.argv)
        print(OUT)
