"""
Randomised parity (tests/checkers/fuzz_parity.py) as part of the GPU suite: random triangle soups, spheres, transforms through
the device refit, runs of frames, rt_group shards, all three kernel variants, the counting and the materials-in-memory kernel
instances — against the oracle's brute force, bit for bit, on every node-format setting.

Why it is in the suite and not only a tool: rt_kernels.hip's phased kernel is built with -mllvm
-structurizecfg-skip-uniform-regions (ray-tracing_amd/Makefile), a flag that was found to MISCOMPILE kernel variant 0 (this
checker found it; those kernels are now built without the flag).  Every build of the library is re-checked here.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("node_format", [0, 1, 2, 3, 4])
def test_fuzz_against_brute_force(node_format):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checkers", "fuzz_parity.py"), "--iters", "400", "--max-tris", "1200",
                        "--seed", str(500 + node_format), "--node-format", str(node_format)],
                       capture_output=True, text=True, timeout=280, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-5:])
    assert r.returncode == 0, f"{tail}\n{r.stderr[-2000:]}"
    assert "all identical" in r.stdout.splitlines()[-1], tail


def test_flag_and_no_flag_builds_render_the_same_bits_at_size():
    """tools/ab_flag_bits.py --quick: C2, C3 (binary and 8-ary nodes) and C4 (8-ary and binary nodes, spheres) at full size from
    librt_amd.so and from librt_amd_noflag.so (no uniform-branch flag anywhere): same accumulation buffers, same ray counts."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_flag_bits.py"), "--quick"], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, f"{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    assert r.stdout.count("identical") == 5 and "DIFFERENT" not in r.stdout, r.stdout
