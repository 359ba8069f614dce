"""pytest configuration: `gpu` marker, repo root on sys.path, and the CPU-emulation build fixture.

`emu_lib` compiles the UNMODIFIED engine sources (wacv23_tsnet_amd/csrc) against tests/emu's HIP
emulation header with a host compiler.  It exists so that host logic, launch geometry, indexing and
MFMA fragment layouts are checked against the oracle in the CPU-only test tier.  It is test
infrastructure: nothing under wacv23_tsnet_amd/ ever loads it.
"""
import ctypes
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def build_emu_lib() -> str:
    csrc = os.path.join(ROOT, "wacv23_tsnet_amd", "csrc")
    src = [os.path.join(csrc, u) for u in ("engine.cpp", "conv_h2_launch.cpp", "conv_h2r_launch.cpp", "conv_g64_launch.cpp", "conv_w1_launch.cpp", "flow_p_launch.cpp")] + [os.path.join(ROOT, "tests", "emu", "emu_runtime.cpp")]
    deps = src + sorted(glob.glob(os.path.join(ROOT, "wacv23_tsnet_amd", "csrc", "*.hpp")))
    deps += [os.path.join(ROOT, "include", "tsnet_abi.h"), os.path.join(ROOT, "tests", "emu", "include", "hip", "hip_runtime.h")]
    out_dir = os.path.join(ROOT, "tests", "emu", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libtsnet_emu.so")
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cxx = None
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", "amdclang++", "clang++"):
        if os.path.isabs(cand) and os.path.exists(cand):
            cxx = cand
            break
        if not os.path.isabs(cand) and subprocess.call(["which", cand], stdout=subprocess.DEVNULL) == 0:
            cxx = cand
            break
    if cxx is None:
        pytest.skip("no clang++ available for the emulation build")
    cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-psabi", "-mf16c",
           "-I" + os.path.join(ROOT, "tests", "emu", "include")] + src + ["-o", out]
    subprocess.check_call(cmd)
    return out


@pytest.fixture(scope="session")
def emu_lib():
    from wacv23_tsnet_amd import _lib
    return _lib.bind(ctypes.CDLL(build_emu_lib()))
