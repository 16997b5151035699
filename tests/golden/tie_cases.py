"""
Tie-case classifier for primary hit ids (SURVEY §8 A4 / north_star: "primary-ray hit triangle IDs must match
bit-exactly, excluding documented FP tie cases, which are counted").

A primary pixel (sample 0 of frame 1, the pixel the AOV records) is a TIE CASE when an intersector with different
rounding than the contract's could legitimately commit another triangle:
  * depth tie:  the brute-force nearest and second-nearest triangles differ by |t2 - t1| <= EPS_T * t1, or
  * edge tie:   the hit lies within EPS_B of a triangle edge or vertex (u, v or 1-u-v < EPS_B), where a neighbouring
                triangle (or a miss, on a silhouette) is an equally valid answer.
HIP vs oracle needs no exclusion (same intersector: zero differing ids are allowed); these are the pixels that WOULD be
excluded when comparing against a real driver's ray query (shaders/main.wgsl:79-83).

    python tests/golden/tie_cases.py        # rewrites tests/golden/tie_counts.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

EPS_T = 1e-5     # relative depth gap
EPS_B = 1e-4     # barycentric distance to an edge
CASES = {"c1_64": "C1", "c2_96x54": "C2", "c3_96x54": "C3"}


def classify(name):
    from oracle import obj_loader, oracle
    cfgs = json.load(open(os.path.join(ROOT, "configs", "baseline_configs.json")))
    cfg = cfgs[CASES[name]]
    z = np.load(os.path.join(HERE, name + ".npz"))
    w, h, spp, _frames = (int(x) for x in z["meta"])
    sc = oracle.OracleScene(obj_loader.load(os.path.join(ROOT, cfg["scene"])), build_bvh=False)
    u = oracle.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=spp)
    out = {"pixels": w * h, "hits": 0, "misses": 0, "depth_ties": 0, "edge_ties": 0, "tie_cases": 0, "eps_t": EPS_T, "eps_b": EPS_B}
    for y in range(h):
        for x in range(w):
            o, d = oracle.primary_ray(u, x, y)
            hit, tuv, ids = sc.trace(o, d, cull=False, mode=0)
            want = z["hits"][y, x]
            if not hit:
                assert want[0] == 0xFFFFFFFF and want[1] == 0xFFFFFFFF
                out["misses"] += 1
                continue
            assert (int(ids[0]), int(ids[1])) == (int(want[0]), int(want[1]))     # the classifier's ray IS the fixture's ray
            out["hits"] += 1
            t, bu, bv = float(tuv[0]), float(tuv[1]), float(tuv[2])
            t2 = sc.second_nearest(o, d, int(ids[0]), int(ids[1]))
            depth = bool(np.isfinite(t2) and abs(t2 - t) <= EPS_T * abs(t))
            edge = min(bu, bv, 1.0 - bu - bv) < EPS_B
            out["depth_ties"] += depth
            out["edge_ties"] += edge
            out["tie_cases"] += depth or edge
    return out


if __name__ == "__main__":
    res = {name: classify(name) for name in CASES}
    json.dump(res, open(os.path.join(HERE, "tie_counts.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
