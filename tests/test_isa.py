"""CPU tier: the hot patch-convolution kernels must not spill inside their main loop.  They sit exactly at their register limits (168 VGPRs =
three workgroups per CU, 256 = two): a source change that looks neutral can make the allocator spill there -- this round an `& 3` on the wave
index cost 11 scratch operations per two slabs and 3 % of the headline before the ISA was looked at.  tools/isa_check.py compiles the
translation unit for gfx950 with -save-temps (hipcc cross-compiles without a GPU) and counts scratch operations in each kernel's loop."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_IC = []


def _isa_check():
    """tools/isa_check.py, loaded once; the four translation units the tests look at are compiled side by side on first use"""
    if not _IC:
        spec = importlib.util.spec_from_file_location("isa_check", os.path.join(ROOT, "tools", "isa_check.py"))
        ic = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ic)
        ic.compile_all(["conv_h2_launch.cpp", "conv_h2r_launch.cpp", "conv_g64_launch.cpp", "conv_w1_launch.cpp", "flow_p_launch.cpp", "engine.cpp"])
        _IC.append(ic)
    return _IC[0]


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_hot_kernels_do_not_spill_in_their_loops():
    ic = _isa_check()
    rows = {r["name"]: r for r in ic.analyse(ic.compile_asm(), "conv_h2_kernel")}
    for name, limit in ic.HOT:
        hit = [r for n, r in rows.items() if name in n]
        assert hit, f"{name} is not in the product library"
        r = hit[0]
        assert r["scratch"] <= limit, f"{name}: {r['scratch']} scratch operations in the main loop (limit {limit}), {r['vgpr']} VGPRs"
        assert r["vgpr"] <= (256 if ("4, 128, 2, 2" in name or ", 24>" in name) else 168), (name, r["vgpr"])
    # the encoder front end and the general kernel's layers of the headline forward (DESIGN.md section 1's "no spill in a hot loop" now
    # covers them; the one tolerated case is written down in tools/isa_check.py)
    for unit, pat, hot in (("conv_h2_launch.cpp", "conv_h2d_kernel", ic.HOT_FRONT[:2]), ("conv_h2_launch.cpp", "conv_h2s_kernel", ic.HOT_FRONT[2:]), ("conv_h2_launch.cpp", "conv_h2d_kernel", ic.HOT_FRONT_DEEP),
                           ("conv_h2r_launch.cpp", "conv_h2r_kernel", ic.HOT_GENERAL), ("conv_g64_launch.cpp", "conv_g64_kernel", ic.HOT_G64),
                           ("conv_g64_launch.cpp", "conv_h2s32_kernel", ic.HOT_S32)):
        rows = {r["name"]: r for r in ic.analyse(ic.compile_asm(unit=unit), pat)}
        for name, limit in hot:
            hit = [r for n, r in rows.items() if name in n]
            assert hit, f"{name} is not in the product library"
            assert hit[0]["scratch"] <= limit and hit[0]["vgpr"] <= 256, (name, hit[0]["scratch"], hit[0]["vgpr"])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_winograd_kernel_fits_fourteen_waves():
    """conv_w1 runs fourteen waves per workgroup (eight MFMA waves + six transform waves): two SIMDs hold four of them, so every instantiation
    must stay within 128 VGPRs.  No spill inside a period loop (the consumers' K loop, the producers' item loop); the producers' prologue and
    tile-boundary code may hold a few values in scratch (<= 128 bytes: a handful of instructions per TILE, since round 5's chunks carry two
    tiles' offsets), the forward's own instantiations (reflection padding or raw input) at most 64."""
    ic = _isa_check()
    asm = ic.compile_asm(unit="conv_w1_launch.cpp")
    res = ic.kernel_resources(asm, "conv_w1_kernel")
    assert len(res) >= 4, res          # the chunk kernel exists for the two-plane (fp16 x 2) stages only
    for name, (vgpr, scratch) in res.items():
        assert vgpr <= 128 and scratch <= (128 if ", 1>" in name else 64), (name, vgpr, scratch)
    # the one-tile path (conv_w1_one.hpp: launches of a single round) is round 4's kernel: 128 VGPRs, no scratch at all in the forward's forms
    one = ic.kernel_resources(asm, "conv_w1_one_kernel")
    assert len(one) >= 8, one
    for name, (vgpr, scratch) in one.items():
        assert vgpr <= 128 and (scratch == 0 or "<1," in name), (name, vgpr, scratch)
    inner = ic.scratch_in_inner_loops(asm, "conv_w1_kernel")
    assert len(inner) == len(res) and all(v == 0 for v in inner.values()), inner


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_large_map_flow_kernel_has_no_packed_fp32_arithmetic():
    """flow_kernel_p built WITH the SLP vectorizer (96 v_pk_fma_f32, 114 v_pk_mul_f32, 79 v_pk_add_f32 in its softmax epilogue, beside the
    other wave's MFMAs) returned different and wrong flows from run to run on the MI355X (csrc/flow_persist.hpp, profiles/round4_flow_cfg4.txt);
    the scalar build is exact.  Its translation unit is compiled with -fno-slp-vectorize: this pins the flag, two waves per SIMD (<= 256
    VGPRs) and no scratch."""
    ic = _isa_check()
    asm = ic.compile_asm(unit="flow_p_launch.cpp")
    res = ic.kernel_resources(asm, "flow_kernel_p")
    assert res, "flow_kernel_p is not in the product library"
    for name, (vgpr, scratch) in res.items():
        assert vgpr <= 256 and scratch == 0, (name, vgpr, scratch)
    text = open(asm).read()
    start = text.index("flow_kernel_p")
    assert "v_mfma_f32_32x32x16_f16" in text[start:]
    assert "v_pk_fma_f32" not in text and "v_pk_mul_f32" not in text and "v_pk_add_f32" not in text
    # round 5: flow_kernel<NT> (csrc/flow_sweep.hpp) is compiled in the same unit -- it carried 189 v_pk_mul_f32 + 85 v_pk_add_f32 beside its
    # MFMAs while it was built in engine.cpp -- and is covered by the same pin
    res2 = ic.kernel_resources(asm, "flow_kernel<")
    assert len(res2) == 2, res2
    for name, (vgpr, scratch) in res2.items():
        assert vgpr <= 256 and scratch == 0, (name, vgpr, scratch)


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_no_packed_fp32_arithmetic_beside_mfma_anywhere():
    """The rule of DESIGN.md section 4.2 (packed fp32 arithmetic in the epilogue of a kernel that runs MFMAs returned wrong flows on the MI355X),
    held library-wide: no unit that contains an MFMA contains v_pk_fma_f32 or v_pk_mul_f32; the flow and Winograd units contain no packed fp32
    arithmetic at all; the direct patch kernels, conv_h2r and conv_g64 keep the v_pk_add_f32 of their accumulator folds (exact adds whose
    results the cross-kernel bit-equality tests pin on the GPU every round).  engine.cpp -- the one unit built WITH the SLP vectoriser (the RGB
    head's packed FMAs are its point) -- contains no MFMA (nothing stopped one from being added there)."""
    import re
    ic = _isa_check()
    for unit in ("conv_h2_launch.cpp", "conv_h2r_launch.cpp", "conv_g64_launch.cpp", "conv_w1_launch.cpp", "flow_p_launch.cpp", "engine.cpp"):
        asm = open(ic.compile_asm(unit=unit)).read()
        has_mfma = "v_mfma" in asm
        fma_mul = len(re.findall(r"v_pk_(?:fma|mul)_f32", asm))
        adds = len(re.findall(r"v_pk_add_f32", asm))
        if unit == "engine.cpp":
            assert not has_mfma, "an MFMA kernel in engine.cpp: move it to a unit built with -fno-slp-vectorize"
            continue
        assert has_mfma and fma_mul == 0, (unit, fma_mul)
        if unit not in ("conv_h2_launch.cpp", "conv_h2r_launch.cpp", "conv_g64_launch.cpp"):
            assert adds == 0, (unit, adds)
