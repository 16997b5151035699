"""
Measures how far a different but legal arithmetic moves the image and the hit ids, and freezes the radiance tolerance
(tests/golden/tolerance.json; see tests/tolerance.py for what the four numbers mean).

Renders each case twice — the contract oracle (RTN-1, liboracle.so) and the alternative-arithmetic oracle
(liboracle_alt.so: libm sinf / cosf / tanf / powf, normalize by division, Moller-Trumbore without explicit fmaf,
-ffp-contract=fast) — on the same scene, camera, seed and spp, and records: per-pixel abs / rel difference quantiles,
image RMSE, the fraction of pixels whose paths diverged (hash of the hit ids of all their rays), and the pixels whose
PRIMARY hit id differs, each of which must be a documented tie case (tests/golden/tie_cases.py: within EPS_B of an
edge or EPS_T of another triangle in depth).

    python tests/golden/make_tolerance.py            # rewrites tests/golden/tolerance.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import obj_loader, oracle  # noqa: E402
import tolerance  # noqa: E402
from golden import tie_cases  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
# name: (config, width, height, samples per frame, frames) — C1 at BASELINE's full size; C2 / C3 at the golden fixtures' window
# with the configs' own samples x frames split (64 and 256 spp)
CASES = {
    "c1_256": ("C1", 256, 256, 16, 1),
    "c2_96x54": ("C2", 96, 54, 16, 4),
    "c3_96x54": ("C3", 96, 54, 16, 16),
    "c3_192x108": ("C3", 192, 108, 16, 16),
}
ABS_FLOOR, REL_FLOOR = 2e-6, 2e-5     # never freeze a per-pixel tolerance below a few f32 ulps of an image in [0, 1]
MARGIN = 4.0                          # frozen bound = measured x MARGIN


def render(sc, u, frames, mode=1):
    W, H = int(u["window"][0][0]), int(u["window"][0][1])
    accum = np.zeros((H, W, 4), np.float32)
    hashes = np.zeros((frames, H, W), np.uint32)
    hits0 = None
    u = np.array(u, copy=True)
    for f in range(frames):
        u["frame"], u["accumulation_frame"] = f + 1, f
        _, hits, _ = sc.render(u, accum, mode, want_hits=(f == 0), path_hash=hashes[f])
        if f == 0:
            hits0 = hits
    return accum, hits0, hashes


def measure(name, verbose=True):
    cname, w, h, spp, frames = CASES[name]
    cfg = json.load(open(os.path.join(ROOT, "configs", "baseline_configs.json")))[cname]
    scene = obj_loader.load(os.path.join(ROOT, cfg["scene"]))
    rtn, alt = oracle.OracleScene(scene), oracle.OracleScene(scene, alt=True)
    u = oracle.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=spp)
    t0 = time.time()
    a, hits_a, hash_a = render(rtn, u, frames)
    b, hits_b, hash_b = render(alt, u, frames)
    diverged = (hash_a != hash_b).any(axis=0)
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3])
    rel = d / np.maximum(np.maximum(np.abs(a[..., :3]), np.abs(b[..., :3])), 1e-6)
    same = ~diverged
    q = lambda x, p: float(np.quantile(x, p)) if x.size else 0.0
    m = {
        "window": [w, h], "samples": spp, "frames": frames, "max_bounces": cfg["max_bounces"],
        "rmse": float(np.sqrt((d ** 2).mean())),
        "abs_diff_quantiles": {p: q(d.max(axis=-1), pp) for p, pp in (("p50", .5), ("p90", .9), ("p99", .99), ("p99.9", .999), ("max", 1.0))},
        "rel_diff_quantiles": {p: q(rel.max(axis=-1), pp) for p, pp in (("p50", .5), ("p90", .9), ("p99", .99), ("p99.9", .999), ("max", 1.0))},
        "diverged_pixel_fraction": float(diverged.mean()),
        "same_path_max_abs": float(d[same].max()) if same.any() else 0.0,
        "same_path_max_rel": float(rel[same].max()) if same.any() else 0.0,
    }
    # primary hit ids (sample 0 of frame 1): every difference must be a documented tie case
    differ = (hits_a != hits_b).any(axis=-1)
    brute = oracle.OracleScene(scene, build_bvh=False)
    u1 = np.array(u, copy=True)
    u1["frame"], u1["accumulation_frame"] = 1, 0
    ties, not_ties = 0, []
    for y, x in np.argwhere(differ):
        o, dd = oracle.primary_ray(u1, int(x), int(y))
        hit, tuv, ids = brute.trace(o, dd, cull=False, mode=0)
        ok = False
        if hit:
            t, bu, bv = float(tuv[0]), float(tuv[1]), float(tuv[2])
            t2 = brute.second_nearest(o, dd, int(ids[0]), int(ids[1]))
            ok = bool(np.isfinite(t2) and abs(t2 - t) <= tie_cases.EPS_T * abs(t)) or min(bu, bv, 1.0 - bu - bv) < tie_cases.EPS_B
        else:
            # the contract's ray misses and the other arithmetic's ray hits: a silhouette; a tie iff the OTHER side's hit is within EPS_B of an edge
            o2, d2 = o, dd
            hit2, tuv2, _ = oracle.OracleScene(scene, build_bvh=False, alt=True).trace(o2, d2, cull=False, mode=0)
            ok = (not hit2) or min(float(tuv2[1]), float(tuv2[2]), 1.0 - float(tuv2[1]) - float(tuv2[2])) < tie_cases.EPS_B
        ties += ok
        if not ok:
            not_ties.append((int(x), int(y)))
    m["primary_hit_id_differences"] = int(differ.sum())
    m["primary_hit_id_differences_that_are_tie_cases"] = int(ties)
    m["primary_pixels"] = int(differ.size)
    if verbose:
        print(name, json.dumps(m), f"({time.time() - t0:.1f} s)", flush=True)
    assert not not_ties, f"{name}: primary hit ids differ outside the tie set at {not_ties}: widen EPS in tie_cases.py and re-freeze"
    return m, (a, b, diverged)


def freeze(m):
    return {"abs": max(ABS_FLOOR, MARGIN * m["same_path_max_abs"]), "rel": max(REL_FLOOR, MARGIN * m["same_path_max_rel"]),
            "rmse": MARGIN * m["rmse"], "outlier_fraction": min(1.0, MARGIN * m["diverged_pixel_fraction"] + 1e-4)}


def main():
    out = {"_comment": "Frozen by tests/golden/make_tolerance.py: contract oracle (RTN-1) vs alternative-arithmetic oracle (-DORACLE_ALT_MATH), BVH mode, "
                       f"bounds = measured x {MARGIN:g} (abs >= {ABS_FLOOR:g}, rel >= {REL_FLOOR:g}); see tests/tolerance.py for the meaning of the four numbers."}
    for name in CASES:
        m, _ = measure(name)
        rec = freeze(m)
        rec["measured"] = m
        out[name] = rec
    json.dump(out, open(os.path.join(HERE, "tolerance.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "measured"} for k, v in out.items() if k != "_comment"}, indent=1))


if __name__ == "__main__":
    main()
