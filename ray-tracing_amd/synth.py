"""
synth.py — deterministic synthetic scenes for BASELINE configs C4 / C5
(SURVEY.md §8d "Synthetic inputs"; the reference has no such scenes).

  C4: ~10 M triangles = copies of the reference's teapot mesh (object 0 of
      scenes/teapot.obj) baked onto a jittered 3-D lattice — baked, not
      instanced, because the reference builds a single BLAS with identity
      transforms (src/scene.rs:53-64) — plus 100 k analytic spheres, all
      Lambertian albedo 0.8, one emissive quad above the cloud, camera in an
      empty cell at the centre so bounce rays are incoherent.
  C5: the same generator at ~200 M triangles.

Everything derives from `seed` through numpy's PCG64 stream.  Output arrays use
the WGSL layouts of include/rt_abi.h (numpy dtypes of this package).
"""
from __future__ import annotations

import math

import numpy as np

from . import MODEL_DTYPE, PRIMITIVE_DTYPE, SPHERE_DTYPE, VERTEX_DTYPE, HostScene


class SynthScene:
    def __init__(self, vertices, indices, primitives, models, spheres, camera):
        self.vertices, self.indices, self.primitives, self.models, self.spheres = vertices, indices, primitives, models, spheres
        self.camera = camera      # dict(position, yaw, pitch)
        self.names = ["teapots", "light"]


def lattice_scene(teapot_obj: str, n_instances: int, n_spheres: int, seed: int = 1, spacing: float = 2.6) -> SynthScene:
    base = HostScene(teapot_obj)
    p0 = base.primitives[0]
    nv, ni = int(p0["vertex_count"]), int(p0["index_count"])
    bv = base.vertices[int(p0["first_vertex"]):int(p0["first_vertex"]) + nv]
    bi = base.indices[int(p0["first_index"]):int(p0["first_index"]) + ni].astype(np.uint32)
    centre = (bv["position"].min(axis=0) + bv["position"].max(axis=0)) * np.float32(0.5)

    rng = np.random.Generator(np.random.PCG64(seed))
    side = max(2, math.ceil((n_instances + 1) ** (1.0 / 3.0)))
    if side % 2 == 0:
        side += 1                                  # odd, so a centre cell exists
    cells = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), axis=-1).reshape(-1, 3)
    mid = side // 2
    keep = ~((cells == mid).all(axis=1))           # the camera's cell stays empty
    cells = cells[keep]
    cells = cells[rng.permutation(len(cells))][:n_instances]
    n_instances = len(cells)
    offs = ((cells - mid) * spacing + rng.uniform(-0.45, 0.45, (n_instances, 3))).astype(np.float32)
    scale = rng.uniform(0.8, 1.25, n_instances).astype(np.float32)

    n_quad_v = 4
    vertices = np.zeros(n_instances * nv + n_quad_v, VERTEX_DTYPE)
    pos = (bv["position"] - centre)[None, :, :] * scale[:, None, None] + offs[:, None, :]
    vertices["position"][:n_instances * nv] = pos.reshape(-1, 3).astype(np.float32)
    vertices["normal"][:n_instances * nv] = np.tile(bv["normal"], (n_instances, 1))
    vertices["uv"][:n_instances * nv] = np.tile(bv["uv"], (n_instances, 1))
    idx = (bi[None, :] + (np.arange(n_instances, dtype=np.uint32) * np.uint32(nv))[:, None]).reshape(-1)

    half = np.float32(side * spacing * 0.5)
    top = np.float32(half + 2.0)
    q = np.array([[-half, top, -half], [half, top, -half], [half, top, half], [-half, top, half]], np.float32)
    vertices["position"][n_instances * nv:] = q
    vertices["normal"][n_instances * nv:] = (0.0, -1.0, 0.0)
    indices = np.concatenate([idx, np.array([0, 1, 2, 0, 2, 3], np.uint32)])

    primitives = np.zeros(2, PRIMITIVE_DTYPE)
    primitives[0] = (0, n_instances * nv, 0, n_instances * ni, 0)
    primitives[1] = (n_instances * nv, n_quad_v, n_instances * ni, 6, 1)
    models = np.zeros(2, MODEL_DTYPE)
    models["diffuse_color"][0] = (0.8, 0.8, 0.8)
    models["diffuse_color"][1] = (0.8, 0.8, 0.8)
    models["emission_color"][1] = (0.57735026, 0.57735026, 0.57735026)
    models["emission_strength"][1] = 8.660254
    models["vertex_start"] = primitives["first_vertex"]
    models["index_start"] = primitives["first_index"]

    spheres = np.zeros(n_spheres, SPHERE_DTYPE)
    if n_spheres:
        c = rng.uniform(-half, half, (n_spheres, 3)).astype(np.float32)
        r = rng.uniform(0.05, 0.2, n_spheres).astype(np.float32)
        near_cam = np.linalg.norm(c, axis=1) < 0.6      # keep the camera outside every sphere
        c[near_cam] += np.float32(1.0)
        spheres["center"], spheres["radius"], spheres["model"] = c, r, 0
    camera = {"position": [0.0, 0.0, 0.0], "yaw": 0.35, "pitch": 0.15}
    return SynthScene(vertices, indices, primitives, models, spheres, camera)


def load_config_scene(cfg, root):
    """cfg["scene"] = "synth:lattice:<instances>:<spheres>:<seed>" (configs/baseline_configs.json)."""
    import os
    _, kind, inst, sph, seed = cfg["scene"].split(":")
    assert kind == "lattice"
    return lattice_scene(os.path.join(root, "scenes", "teapot.obj"), int(inst), int(sph), int(seed))
