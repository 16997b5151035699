"""
The guarded build of the render kernel (make guard -> librt_amd_guard.so, -DRT_GUARD): every traversal reference is
range-checked against the node / triangle arrays before use, stack pointers are bounded by the LDS the launch gave them,
and a ray's traversal steps are capped.  A failed check ends the ray and counts in RtStats.guard_trips.  Round 1 reported
a GPU fault in an experimental kernel variant at the 61x37 window that was never explained; this test
runs that window and the C4-lattice case through the guarded build of the SHIPPING kernel, on each node format, and
requires zero trips and the oracle's bits.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_guarded_build_never_trips():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checkers", "guard_check.py")], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    cases = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(cases) >= 4
    for c in cases:
        assert c["guard_trips"] == 0 and c["identical"] and c["rays_equal"], c
