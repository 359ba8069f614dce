"""The C-ABI library loads and exports every symbol include/tsnet_abi.h declares (no compute calls:
this tier has no GPU).  The HIP library is cross-compiled by hipcc if it is not built yet."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "tsnet_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(tsnet_[a-z0-9_]+)\s*\(", src))


def test_header_matches_binding_list():
    from wacv23_tsnet_amd import _lib
    assert header_symbols() == set(_lib.ABI_SYMBOLS)


def test_hip_library_exports_every_symbol():
    from wacv23_tsnet_amd import build
    try:
        path = build.build(verbose=False)
    except RuntimeError as e:
        pytest.skip(str(e))
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (tsnet_[a-z0-9_]+)", out))
    assert header_symbols() <= exported
    needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
    assert "libamdhip64.so" in needed            # it is the HIP build, not a host stub
    kernels = subprocess.run(["strings", path], capture_output=True, text=True).stdout
    assert "conv_h2_kernel" in kernels and "conv_h2r_kernel" in kernels and "flow_kernel" in kernels
    # ONE schedule in the product library: no superseded convolution generation, no ablation / experiment instantiation
    for legacy in ("conv_igemm", "conv_dma", "conv_x3"):
        assert legacy not in kernels, legacy


def test_product_sources_do_not_read_the_environment():
    """Behaviour of the product library must not depend on environment variables: no getenv anywhere under csrc/."""
    import glob
    for f in glob.glob(os.path.join(ROOT, "wacv23_tsnet_amd", "csrc", "*")):
        assert "getenv" not in open(f).read(), f


def test_library_loads_and_reports_version():
    import torch  # noqa: F401  (maps torch's libamdhip64 first)
    from wacv23_tsnet_amd import _lib, build
    try:
        build.build(verbose=False)
    except RuntimeError as e:
        pytest.skip(str(e))
    lib = _lib.load()
    assert lib.tsnet_abi_version() == 5
    for sym in _lib.ABI_SYMBOLS:
        assert hasattr(lib, sym)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from wacv23_tsnet_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_cached", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_model_refuses_cpu_execution():
    """There is no PyTorch/CPU execution path behind the reference-surface object."""
    import torch
    from wacv23_tsnet_amd.model import TSNet, Encoder
    m = TSNet(is_train=False, label_nc=2, n_blocks=0, n_downsampling=3, n_source=1, height=64, width=64)
    x = torch.zeros(1, 3, 64, 64)
    m.set_test_input([x], [torch.zeros(1, 2, 64, 64)], [torch.zeros(1, 64, 64)], torch.zeros(1, 2, 64, 64), torch.zeros(1, 64, 64))
    with pytest.raises(RuntimeError, match="MI355X"):
        m.forward()
    with pytest.raises(RuntimeError, match="parameter container"):
        Encoder(5)(x)
    with pytest.raises(NotImplementedError):
        TSNet(is_train=True, n_downsampling=3)
