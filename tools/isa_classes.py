#!/usr/bin/env python3
"""Static VALU instruction mix of the phased kernel's phase bodies by ISSUE-RATE CLASS (profiles/r3_calibration.txt, CALIBRATE_MORE run):
fast = 2.4-2.6 nominal clocks per wave64 instruction (v_add/sub/mul_f32, v_fmac_f32, v_add/sub_u32, v_and/or/xor_b32, v_mov_b32),
fma = v_fma_f32 3.2, quarter = v_rcp / v_sqrt / v_rsq 8.2, slow = everything else 4.1-4.5.
usage: ISA_FMT=0|3 ISA_SPH=0|1 python tools/isa_classes.py      (runs tools/isa_bodies.py for the listing)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32",
        "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32")
CLK = {"fast": 2.5, "fma": 3.2, "slow": 4.25, "quarter": 8.2}
def klass(op):
    op = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if op in FAST: return "fast"
    if op == "v_fma_f32": return "fma"
    if op in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_iflag_f32"): return "quarter"
    return "slow"
for a, b in (("node_begin", "node_end"), ("leaf_begin", "shade_begin"), ("shade_begin", "shade_end")):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_bodies.py"), a, b], capture_output=True, text=True).stdout
    n = {"fast": 0, "fma": 0, "slow": 0, "quarter": 0}
    for line in out.splitlines():
        t = line.strip().split()
        if t and t[0].startswith("v_"): n[klass(t[0])] += 1
    tot = sum(n.values())
    clk = sum(n[k] * CLK[k] for k in n)
    print(f"{a} .. {b}: {tot} VALU: " + ", ".join(f"{k} {n[k]} ({100.0 * n[k] / max(tot, 1):.0f} %)" for k in n) + f"; {clk / max(tot, 1):.2f} clocks per instruction by class rates")
