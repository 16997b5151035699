"""
-mllvm -structurizecfg-skip-uniform-regions (ray-tracing_amd/Makefile) keeps wave-uniform branches out of the structurizer; it was
caught miscompiling a nested uniform branch whose arms hold DIVERGENT LOOPS (kernel variants 0 / 1).  The
phased kernel gets the flag on the strength of having no divergent loop at all.  This test holds the build to that: the
translation unit compiled with the flag must contain no divergent-loop back edge in any kernel (ISA pattern of a divergent loop:
`s_andn2_b64 exec, exec, <mask>` directly followed by a backward `s_cbranch_execnz`), and the Makefile must give the flag to that
unit only.  The detector is checked on the other unit, where such loops exist.  CPU only (device-only compile to assembly).
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "ray-tracing_amd")
BASE = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
        "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "--cuda-device-only", "-S"]


def divergent_loop_back_edges(asm_path):
    """kernel name -> number of divergent-loop back edges in its ISA"""
    text = open(asm_path).read()
    out = {}
    for name, body in re.findall(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        labels, n, instr, cleared_at = set(), 0, 0, -9
        for line in body.split("\n"):
            line = line.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", line)
            if m:
                labels.add(m.group(1))
                continue
            if not line or line[0] in ";.":
                continue
            instr += 1
            if line.startswith("s_andn2_b64 exec, exec,"):
                cleared_at = instr
            m = re.match(r"s_cbranch_execnz\s+(\.LBB\d+_\d+)", line)
            if m and m.group(1) in labels and instr - cleared_at == 1:      # target already seen: a backward branch
                n += 1
        out[name] = n
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_uniform_branch_flag_only_where_no_divergent_loop_exists(tmp_path):
    mk = open(os.path.join(PKG, "Makefile")).read()
    rules = re.findall(r"^(build/[\w/]+\.o):[^\n]*\n\t([^\n]+)", mk, re.M)
    flagged = [t for t, cmd in rules if "UNIFORM_BRANCHES" in cmd]
    assert sorted(flagged) == ["build/guard/rt_kernels.o", "build/rt_kernels.o"], flagged
    for t, cmd in rules:
        if t in flagged:
            assert "-DRT_TU_PHASED" in cmd and "-DRT_TU_SIMPLE" not in cmd, (t, cmd)
    flag = re.search(r"^UNIFORM_BRANCHES \?= (.*)$", mk, re.M).group(1).split()
    src = os.path.join(PKG, "csrc", "rt_kernels.hip")
    phased, simple = str(tmp_path / "phased.s"), str(tmp_path / "simple.s")
    subprocess.run(BASE + flag + ["-DRT_TU_PHASED", src, "-o", phased], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(BASE + ["-DRT_TU_SIMPLE", src, "-o", simple], check=True, stderr=subprocess.DEVNULL)
    p, s = divergent_loop_back_edges(phased), divergent_loop_back_edges(simple)
    assert sum("rt_render_phased" in k for k in p) == 96 and not any("rt_render_persistent" in k or "rt_render_tiles" in k for k in p)
    assert {k: v for k, v in p.items() if v} == {}, "a divergent loop in the unit built with -structurizecfg-skip-uniform-regions"
    # the detector sees the per-lane traversal loops of variants 0 / 1
    assert all(v >= 1 for k, v in s.items() if "rt_render_persistent" in k or "rt_render_tiles" in k) and len(s) == 16
