#!/usr/bin/env python3
"""Static instruction counts of the phased kernel's phase bodies (markers are injected into a temp copy)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "ray-tracing_amd/csrc/rt_kernels.hip")).read()
def mark(s, needle, name, after=True):
    assert needle in s, needle
    return s.replace(needle, (needle + f'\n asm volatile("; MARK {name}");') if after else (f' asm volatile("; MARK {name}");\n' + needle), 1)
src = mark(src, "        if (pick == PH_NODE) {\n            if constexpr (kPrioNode != 0) __builtin_amdgcn_s_setprio(kPrioNode);\n            do {", "node_begin")
src = mark(src, "            } while (pick == PH_NODE);", "node_end")
src = mark(src, "        } else if (pick == PH_LEAF) {\n            if constexpr (kPrioLeaf != 0) __builtin_amdgcn_s_setprio(kPrioLeaf);\n            do {", "leaf_begin")
src = mark(src, "            } while (pick == PH_LEAF);", "leaf_end")
src = mark(src, "        } else if (pick == PH_SHADE) {", "shade_begin")
src = mark(src, "            // pixel boundary: write out finished", "shade_end", after=False)
EXTRA = os.environ.get("ISA_EXTRA", "")   # extra compiler flags to try
os.makedirs("/tmp/isa", exist_ok=True)
open("/tmp/isa/marked.hip", "w").write(src)
subprocess.check_call(f"cp {ROOT}/ray-tracing_amd/csrc/rt_device.h {ROOT}/ray-tracing_amd/csrc/rt_math.h {ROOT}/ray-tracing_amd/csrc/rt_pack.h /tmp/isa/ && /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize --cuda-device-only -DRT_TU_PHASED -mllvm -structurizecfg-skip-uniform-regions=1 {EXTRA} -S /tmp/isa/marked.hip -o /tmp/isa/marked.s 2>/dev/null", shell=True)
s = open("/tmp/isa/marked.s").read()
FMT = os.environ.get('ISA_FMT', '2')   # node format instance: 0 = 48-byte binary, 1 = 32-byte binary, 2 = 8-ary
SPH = os.environ.get('ISA_SPH', '0')   # EXT of the instance: 0 triangles, 1 + spheres, 2 + textures
m = re.search(r'(_ZN12_GLOBAL__N_116rt_render_phasedILb0ELb0ELi' + SPH + r'ELb1ELi' + FMT + r'ELb0EEEv12RenderParams:.*?\.end_amdhsa_kernel)', s, re.S)
body = m.group(1).split('\n')
marks = [(i, l.strip().split()[-1]) for i, l in enumerate(body) if 'MARK' in l]
def count(a, b):
    ls = [l.strip() for l in body[a:b] if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
    v = sum(1 for l in ls if l.startswith('v_')); mv = sum(1 for l in ls if l.startswith('v_mov'))
    return dict(total=len(ls), valu=v, vmov=mv, salu=sum(1 for l in ls if l.startswith('s_')), mem=sum(1 for l in ls if l.startswith(('global_', 'ds_'))))
print(marks)
for k in range(len(marks) - 1):
    print(marks[k][1], '->', marks[k + 1][1], count(marks[k][0], marks[k + 1][0]))
print('whole kernel', count(0, len(body)))
if len(sys.argv) > 1:
    a = [i for i, n in marks if n == sys.argv[1]][0]; b = [i for i, n in marks if n == sys.argv[2]][0]
    for l in body[a:b + 1]:
        if l.strip() and not l.strip().startswith(';') and 'implicit-def' not in l: print(l)
