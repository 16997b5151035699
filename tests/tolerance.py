"""
The north_star's radiance tolerance, as code: "accumulated radiance within a stated per-pixel absolute/relative tolerance
plus an image-RMSE bound".

Two renders of one config by two LEGAL implementations of the reference's arithmetic (WGSL leaves sin / cos / tan / pow /
normalize accuracy, FMA contraction and the whole ray query to the driver: shaders/ray.wgsl:6,55-58, main.wgsl:79-83,107)
agree like this:
  * a pixel whose paths took the same turns in both differs by rounding only:  |a - b| <= abs + rel * max(|a|, |b|);
  * a pixel in which some compare flipped (a hit id near an edge, `rand() < p`, Schlick) follows other paths from there
    on and is an OUTLIER: unbounded per pixel (the image is in [0, 1]), bounded in NUMBER (outlier_fraction) and in
    weight (rmse over the whole image, outliers included).
tests/golden/tolerance.json holds {abs, rel, rmse, outlier_fraction} per config, frozen from the measured difference
between the contract oracle (RTN-1) and the alternative-arithmetic oracle (oracle/rt_oracle.c, -DORACLE_ALT_MATH) by
tests/golden/make_tolerance.py.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    return json.load(open(os.path.join(HERE, "golden", "tolerance.json")))


def compare(a, b, abs_tol, rel_tol):
    """a, b: [H, W, >=3] f32 accumulation buffers.  Returns (outlier mask [H, W], rmse, max abs diff of the non-outliers)."""
    a = np.asarray(a, np.float64)[..., :3]
    b = np.asarray(b, np.float64)[..., :3]
    d = np.abs(a - b)
    d = np.where(np.isnan(a) & np.isnan(b), 0.0, d)
    lim = abs_tol + rel_tol * np.maximum(np.abs(a), np.abs(b))
    out = ~(d <= lim)                       # NaN on one side only: an outlier
    outlier = out.any(axis=-1)
    rmse = float(np.sqrt(np.nanmean(np.where(np.isnan(d), 1.0, d) ** 2)))
    inl = d[~outlier]
    return outlier, rmse, (float(inl.max()) if inl.size else 0.0)


def check(a, b, tol, what=""):
    """Assert that two renders agree within a frozen tolerance record {abs, rel, rmse, outlier_fraction}."""
    outlier, rmse, _ = compare(a, b, tol["abs"], tol["rel"])
    frac = float(outlier.mean())
    assert frac <= tol["outlier_fraction"], f"{what}: {frac:.5f} of the pixels outside abs {tol['abs']:g} / rel {tol['rel']:g}, bound {tol['outlier_fraction']:g}"
    assert rmse <= tol["rmse"], f"{what}: image RMSE {rmse:.3g}, bound {tol['rmse']:g}"
    return frac, rmse
