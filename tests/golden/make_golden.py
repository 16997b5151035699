"""
Generates the golden fixtures under tests/golden/ from the CPU oracle
(brute-force intersection mode, the ground truth).  The reference itself cannot
run here (Rust + Vulkan ray query), so these vectors pin the HIP
path and the oracle's faster BVH mode to the oracle's brute-force mode, and pin
all of them against regressions; they are NOT captures of the running reference.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import obj_loader, oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    # name: (config, width, height, samples, frames)
    "c1_64": ("C1", 64, 64, 4, 2),
    "c2_96x54": ("C2", 96, 54, 2, 1),
    "c3_96x54": ("C3", 96, 54, 2, 1),
}


def main():
    cfgs = json.load(open(os.path.join(ROOT, "configs", "baseline_configs.json")))
    for name, (cname, w, h, spp, frames) in CASES.items():
        cfg = cfgs[cname]
        sc = oracle.OracleScene(obj_loader.load(os.path.join(ROOT, cfg["scene"])))
        u = oracle.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=spp)
        accum = np.zeros((h, w, 4), np.float32)
        rays = 0
        hits0 = None
        for f in range(frames):
            u["frame"] = f + 1
            u["accumulation_frame"] = f
            _, hits, cnt = sc.render(u, accum, mode=0, want_hits=True)
            rays += cnt.rays
            if f == 0:
                hits0 = hits
        np.savez_compressed(os.path.join(HERE, name + ".npz"), accum=accum, hits=hits0, rays=np.uint64(rays),
                            meta=np.array([w, h, spp, frames], np.uint32))
        print(name, "rays", rays, "mean", accum[..., :3].mean())


if __name__ == "__main__":
    main()
