"""
The north_star's radiance tolerance: stated in tests/golden/tolerance.json, earned by
tests/golden/make_tolerance.py, enforced here.

CPU: the alternative-arithmetic oracle (libm transcendentals, division, no explicit fmaf, -ffp-contract=fast) against the
contract oracle stays inside the frozen bounds, every pixel outside the per-pixel abs / rel tolerance is one whose paths
diverged, and every differing primary hit id is a documented tie case.
GPU: the HIP path against the ALTERNATIVE oracle stays inside the same bounds — i.e. the product agrees with an
implementation that shares none of its arithmetic choices as closely as two legal implementations agree with each other.
"""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, config, scene_path

sys.path.insert(0, os.path.join(ROOT, "tests"))
import tolerance  # noqa: E402
from golden import make_tolerance  # noqa: E402


def test_alt_oracle_is_another_arithmetic(oracle_mod):
    alt, rtn = oracle_mod.lib(True), oracle_mod.lib(False)
    assert alt.oracle_variant() == 1 and rtn.oracle_variant() == 0
    import ctypes as C
    differ = 0
    for k in range(2000):
        x = np.float32(0.001 * k * 3.7)
        s0, c0, s1, c1 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        rtn.oracle_sincos(C.c_float(x), C.byref(s0), C.byref(c0))
        alt.oracle_sincos(C.c_float(x), C.byref(s1), C.byref(c1))
        assert abs(s0.value - s1.value) <= 5e-7 and abs(c0.value - c1.value) <= 5e-7      # both far inside Vulkan's 2^-11
        differ += (s0.value != s1.value) or (c0.value != c1.value)
    assert differ > 20          # ... and not the same function


def test_tolerance_file_is_complete():
    tol = tolerance.load()
    for name in make_tolerance.CASES:
        rec = tol[name]
        assert set(("abs", "rel", "rmse", "outlier_fraction", "measured")) <= set(rec)
        assert 0 < rec["abs"] <= 1e-4 and 0 < rec["rel"] <= 1e-3 and 0 < rec["rmse"] <= 0.02 and 0 < rec["outlier_fraction"] <= 0.1
        # frozen = measured x margin (a re-run of the generator that moved the measurement must be committed, not drift)
        assert rec == {**make_tolerance.freeze(rec["measured"]), "measured": rec["measured"]}


@pytest.mark.parametrize("name", list(make_tolerance.CASES))
def test_alt_oracle_vs_contract_oracle_inside_frozen_tolerance(name):
    tol = tolerance.load()[name]
    m, (a, b, diverged) = make_tolerance.measure(name, verbose=False)      # asserts: differing primary ids are all tie cases
    frac, rmse = tolerance.check(a, b, tol, name)
    outlier, _, inl_max = tolerance.compare(a, b, tol["abs"], tol["rel"])
    assert not (outlier & ~diverged).any(), "a pixel that followed the same paths in both is outside the per-pixel tolerance"
    assert m["diverged_pixel_fraction"] <= tol["outlier_fraction"]
    assert m["primary_hit_id_differences"] == m["primary_hit_id_differences_that_are_tie_cases"]
    # the measurement is deterministic: what was frozen is what is measured now
    assert abs(m["rmse"] - tol["measured"]["rmse"]) <= 1e-9 + 1e-6 * tol["measured"]["rmse"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(make_tolerance.CASES))
def test_hip_vs_alt_oracle_inside_tolerance(rt, oracle_mod, obj_loader, name):
    """HIP accumulation buffer against the alternative-arithmetic oracle: inside {abs, rel, rmse, outlier fraction}."""
    tol = tolerance.load()[name]
    cname, w, h, spp, frames = make_tolerance.CASES[name]
    cfg = config(cname)
    scene = obj_loader.load(os.path.join(ROOT, cfg["scene"]))
    alt = oracle_mod.OracleScene(scene, alt=True)
    u = oracle_mod.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=spp)
    want, hits_alt, _ = make_tolerance.render(alt, u, frames)
    with rt.Context(0) as ctx:
        ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ru = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=spp)
        ru["frame"], ru["accumulation_frame"] = 0, 1
        ctx.dispatch_frames(ru, frames)
        got = ctx.accum_download()
    frac, rmse = tolerance.check(got, want, tol, f"HIP vs alt oracle, {name}")
    assert frac > 0 or rmse > 0 or name == "never", "HIP equals the ALTERNATIVE arithmetic bit for bit: the yardstick is not independent"
