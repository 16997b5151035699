#!/usr/bin/env python3
"""Randomised parity run: random triangle soups (sizes, extents, offsets from the origin), random geometry
transforms applied through rt_models_update (device refit), random cameras and materials; HIP path against the
oracle's BRUTE-FORCE mode, bit for bit.  usage: python tests/checkers/fuzz_parity.py [--iters 40] [--seed 1]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
from oracle import obj_loader, oracle

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-tris", type=int, default=400, help="per geometry")
ap.add_argument("--node-format", type=int, default=0, help="RT_OPT_NODE_FORMAT: 0 by size, 1 48-byte, 2 32-byte binary nodes, 3 80-byte, 4 64-byte 8-ary nodes")
args = ap.parse_args()
rt = graft.load_package()
rng = np.random.default_rng(args.seed)
bad = 0
for it in range(args.iters):
    n_geo = int(rng.integers(1, 4))
    verts, idx, prims, mods = [], [], [], []
    centre = (rng.random(3) - 0.5) * 10.0 ** rng.integers(0, 5)
    spread = 10.0 ** rng.uniform(-1, 2)
    for g in range(n_geo):
        n = int(rng.integers(1, args.max_tris))
        size = spread * 10.0 ** rng.uniform(-2.5, -0.3)
        c = centre + (rng.random((n, 1, 3)) - 0.5) * 2 * spread
        p = (c + (rng.random((n, 3, 3)) - 0.5) * 2 * size).astype(np.float32)
        v = np.zeros(3 * n, obj_loader.VERTEX_DTYPE)
        v["position"] = p.reshape(-1, 3)
        nr = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        nr /= np.maximum(np.linalg.norm(nr, axis=1, keepdims=True), 1e-20)
        v["normal"] = np.repeat(nr.astype(np.float32), 3, axis=0)
        prims.append((sum(len(x) for x in verts), 3 * n, sum(len(x) for x in idx), 3 * n, g))
        m = np.zeros(1, obj_loader.MODEL_DTYPE)
        m["tag"] = 1 if rng.random() < 0.25 else 0
        m["diffuse_color"] = rng.random(3)
        m["specular_color"] = rng.random(3)
        m["specular_probability"] = rng.random()
        m["roughness"] = rng.random()
        m["emission_color"] = rng.random(3)
        m["emission_strength"] = rng.random() * (rng.random() < 0.3)
        m["refractive_index"] = 1.0 + rng.random()
        m["vertex_start"] = prims[-1][0]
        m["index_start"] = prims[-1][2]
        mods.append(m)
        verts.append(v)
        idx.append(np.arange(3 * n, dtype=np.uint32))
    V = np.zeros(sum(len(x) for x in verts), obj_loader.VERTEX_DTYPE)
    o = 0
    for v in verts:
        V[o:o + len(v)] = v; o += len(v)
    I = np.concatenate(idx)
    P = np.zeros(n_geo, obj_loader.PRIMITIVE_DTYPE)
    for g, t in enumerate(prims): P[g] = t
    M = np.zeros(n_geo, obj_loader.MODEL_DTYPE)
    for g, m in enumerate(mods): M[g] = m[0]
    T = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (n_geo, 1))
    if rng.random() < 0.7:
        T += ((rng.random((n_geo, 12)) - 0.5) * 0.4).astype(np.float32)
        T[:, 3::4] += ((rng.random((n_geo, 3)) - 0.5) * spread).astype(np.float32)
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    cam = centre + (rng.random(3) - 0.5) * 3 * spread
    kw = dict(max_bounces=int(rng.integers(0, 6)), samples=int(rng.integers(1, 4)), fov=float(rng.uniform(0.3, 2.5)),
              flags=int(rng.integers(0, 2)), frame=int(rng.integers(1, 2**32)))
    d = centre - cam                                     # look at the soup (ray.wgsl:52-58), with some jitter
    d = d / max(np.linalg.norm(d), 1e-20)
    yaw = float(np.arctan2(d[2], d[0]) + rng.uniform(-0.3, 0.3))
    pitch = float(np.clip(np.arcsin(d[1]) + rng.uniform(-0.3, 0.3), -1.5, 1.5))
    scene = obj_loader.Scene(V, I, P, M)
    want, want_hits, cnt = oracle.OracleScene(scene, transforms=T, build_bvh=False).render(
        oracle.make_uniform(w, h, cam, yaw, pitch, **kw), mode=0, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, args.node_format)
        ctx.scene_upload(V, I, P, M)
        ctx.models_update(M, transforms=T)               # device refit of the tree built for the untransformed scene
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ctx.dispatch(rt.make_uniform(w, h, cam, yaw, pitch, **kw))
        got, hits, st = ctx.accum_download(), ctx.aov_download(), ctx.stats()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    ok = st.rays == cnt.rays and np.array_equal(hits, want_hits) and bool(same.all())
    hit_frac = float((want_hits[..., 0] != 0xFFFFFFFF).mean())
    print(f"iter {it}: {len(I) // 3} tris in {n_geo} geometries, {w}x{h}, {kw}, primary hit fraction {hit_frac:.2f}, rays {cnt.rays}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("fuzz:", "all identical" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
