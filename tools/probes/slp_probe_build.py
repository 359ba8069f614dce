"""CPU (cross-compile): builds of the library whose flow translation unit is compiled WITH the SLP vectorizer (the build that returned wrong
flows from flow_kernel_p in round 4: csrc/flow_persist.hpp), plain and with probe states inserted, for tools/probes/slp_probe_run.py:
    lib/libtsnet_probe_slp0.so   SLP on                                  (the failing build)
    lib/libtsnet_probe_slp1.so   SLP on + 64 idle states between the MFMA sweep and the epilogue's first read of the accumulators
    lib/libtsnet_probe_slp2.so   SLP on + s_waitcnt vmcnt(0) lgkmcnt(0) + 32 idle states before the packed arithmetic
Only flow_p_launch.cpp is recompiled; the other objects are the tools build's.  Prints the packed-fp32 instruction counts of each build."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wacv23_tsnet_amd import build as B

B.build_tools(verbose=False)
objd = os.path.join(B.HERE, "lib", "obj_tools")
for k in (0, 1, 2):
    obj = os.path.join(objd, f"flow_p_probe{k}.o")
    cmd = ["/opt/rocm/bin/hipcc"] + B.FLAGS + ["-DTSNET_TOOLS", f"-DTSNET_FLOWP_PROBE={k}", "-save-temps=obj", "-c", os.path.join(B.CSRC, "flow_p_launch.cpp"), "-o", obj]
    subprocess.check_call(cmd, cwd=objd)
    asm = open(os.path.join(objd, "flow_p_launch-hip-amdgcn-amd-amdhsa-gfx950.s")).read()          # (-save-temps names the temporaries after the source)
    body = asm[asm.index("flow_kernel_pILi0"):]
    body = body[:body.index(".Lfunc_end")]
    print(f"probe {k}: " + "  ".join(f"{op} {len(re.findall(op, body))}" for op in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mfma", "s_nop")))
    objs = [os.path.join(objd, u.replace(".cpp", ".o")) for u in B.UNITS if u != "flow_p_launch.cpp"] + [obj]
    out = os.path.join(B.HERE, "lib", f"libtsnet_probe_slp{k}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    print("  ", out)
