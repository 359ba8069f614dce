"""Compile engine.cpp with -save-temps and print a compact op sequence of the blocks of one kernel that
contain MFMA / LDS-DMA instructions.  usage: asm_loop.py <mangled-name-regex> [max_chars]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/asm"; os.makedirs(out, exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-fhip-fp32-correctly-rounded-divide-sqrt",
                "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form", "-save-temps", "-c", os.path.join(ROOT, "wacv23_tsnet_amd/csrc/engine.cpp"), "-o", "e.o"],
               cwd=out, stderr=subprocess.DEVNULL)
s = open(out + "/engine-hip-amdgcn-amd-amdhsa-gfx950.s").read()
pat = sys.argv[1]; maxc = int(sys.argv[2]) if len(sys.argv) > 2 else 900
name = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if re.search(pat, n)][0]
start = s.index(name + ':'); end = s.index('.Lfunc_end', start)
blk = []; cur = ['entry']
for l in s[start:end].split('\n'):
    if re.match(r'^\.LBB', l): blk.append(cur); cur = [l]
    else: cur.append(l)
blk.append(cur)
print(name)
for b in blk:
    if not any('mfma' in x or 'load_lds' in x for x in b): continue
    seq = []
    for l in b[1:]:
        t = l.strip().split()
        if not t or t[0].startswith((';', '.')): continue
        op = t[0]
        if 'mfma' in op: seq.append('M')
        elif op.startswith('ds_read'): seq.append('r')
        elif op.startswith('ds_write'): seq.append('W')
        elif 'load_lds' in op: seq.append('G')
        elif op.startswith('global_load'): seq.append('g')
        elif op.startswith('s_waitcnt'): seq.append('w(' + ''.join(t[1:]) + ')')
        elif op.startswith('s_barrier'): seq.append('BAR')
        elif op.startswith(('s_cbranch', 's_branch')): seq.append('br')
        elif op.startswith('v_'): seq.append('v')
        elif op.startswith('s_'): seq.append('s')
        else: seq.append(op)
    o = ' '.join(seq)
    for ch in 'vsM':
        o = re.sub(r'(?:%s ){3,}' % ch, lambda m: '%s*%d ' % (ch, len(m.group(0)) // 2), o)
    print(b[0][:12], o[:maxc]); print()
m = re.search(r'\.name:\s+' + re.escape(name) + r'.*?\.vgpr_count:\s+(\d+)', s, re.S)
i = s.index('.name:           ' + name) if ('.name:           ' + name) in s else -1
if i > 0:
    seg = s[i - 1500:i + 800]
    print({k: re.search(k + r':\s+(\d+)', seg).group(1) for k in ('.vgpr_count', '.sgpr_count', '.group_segment_fixed_size', '.vgpr_spill_count') if re.search(k + r':\s+(\d+)', seg)})
