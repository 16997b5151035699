#!/usr/bin/env python3
"""Experiment (timing only, NOT a parity case): what would splitting the scene's giant triangles buy?  Re-tessellates every
triangle whose box covers more than 1 % of the scene's into N x N pieces IN THE INPUT and times the same frames on both scenes.
usage: python tests/checkers/exp_ground_split.py [config] [N] [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
from oracle import obj_loader

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rt = graft.load_package()
cfg = graft.load_config(name)
s = obj_loader.load(os.path.join(ROOT, cfg["scene"]))


def cat(parts):
    out = np.zeros(sum(len(x) for x in parts), obj_loader.VERTEX_DTYPE)
    o = 0
    for x in parts:
        out[o:o + len(x)] = x
        o += len(x)
    return out


def split(scene, n):
    V, I, P, M = scene.vertices, scene.indices, scene.primitives, scene.models
    pos = V["position"]
    ext = pos.max(0) - pos.min(0)
    area = lambda e: e[..., 0] * e[..., 1] + e[..., 1] * e[..., 2] + e[..., 2] * e[..., 0]
    nv, ni, prims, models = [], [], [], []
    for g in range(len(P)):
        fv, cv, fi, ci = int(P[g]["first_vertex"]), int(P[g]["vertex_count"]), int(P[g]["first_index"]), int(P[g]["index_count"])
        assert int(M[g]["vertex_start"]) == fv and int(M[g]["index_start"]) == fi
        v = V[fv:fv + cv]
        idx = I[fi:fi + ci].reshape(-1, 3)
        tp = v["position"][idx]
        big = area(tp.max(1) - tp.min(1)) > 0.01 * area(ext)
        out_v, out_i = [v], [idx[~big]]
        base = cv
        for t in idx[big]:
            a, b, c = (v[int(k)] for k in t)
            grid = {}
            verts = []
            for i in range(n + 1):
                for j in range(n + 1 - i):
                    wa, wb, wc = (n - i - j) / n, i / n, j / n
                    x = np.zeros(1, obj_loader.VERTEX_DTYPE)
                    for f in ("position", "normal", "uv"):
                        x[f] = (wa * a[f].astype(np.float64) + wb * b[f].astype(np.float64) + wc * c[f].astype(np.float64)).astype(np.float32)
                    grid[(i, j)] = base + len(verts)
                    verts.append(x)
            tris = []
            for i in range(n):
                for j in range(n - i):
                    tris.append((grid[(i, j)], grid[(i + 1, j)], grid[(i, j + 1)]))
                    if i + j < n - 1:
                        tris.append((grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]))
            out_v.append(cat(verts))
            out_i.append(np.array(tris, np.uint32))
            base += len(verts)
        gv = cat(out_v)
        gi = np.concatenate(out_i).reshape(-1).astype(np.uint32)
        prims.append((sum(len(x) for x in nv), len(gv), sum(len(x) for x in ni), len(gi), int(P[g]["transformation_offset"])))
        m = M[g].copy()
        m["vertex_start"], m["index_start"] = prims[-1][0], prims[-1][2]
        models.append(m)
        nv.append(gv)
        ni.append(gi)
    P2 = np.zeros(len(P), obj_loader.PRIMITIVE_DTYPE)
    for g, t in enumerate(prims):
        P2[g] = t
    M2 = np.zeros(len(M), obj_loader.MODEL_DTYPE)
    for g, m in enumerate(models):
        M2[g] = m
    return obj_loader.Scene(cat(nv), np.concatenate(ni), P2, M2)


def run(scene, label):
    W, H = cfg["width"], cfg["height"]
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 0)
        ctx.scene_upload(scene.vertices, scene.indices, scene.primitives, scene.models)
        ctx.accum_resize(W, H)
        best = 1e9
        for rep in range(3):
            ctx.accum_resize(W, H)
            ctx.stats_reset()
            u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
            u["frame"] = 0
            u["accumulation_frame"] = 1
            ctx.dispatch_frames(u, steps)
            ctx.sync()
            best = min(best, ctx.last_dispatch_ms() / steps)
        st = ctx.stats()
    with rt.Context(0) as ctx:                                  # per-ray work, one small counted frame
        ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 1)
        ctx.scene_upload(scene.vertices, scene.indices, scene.primitives, scene.models)
        ctx.accum_resize(W // 4, H // 4)
        ctx.dispatch(rt.make_uniform(W // 4, H // 4, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=np.float32(W) / np.float32(H),
                                     max_bounces=cfg["max_bounces"], samples=cfg["samples"]))
        c = ctx.stats()
    print(f"{label}: {len(scene.indices) // 3} triangles, {st.bvh_nodes} nodes, {best:.3f} ms/step, {st.rays / steps / 1e6:.1f} M rays/step, "
          f"{c.node_visits / c.rays:.2f} node visits and {c.triangle_tests / c.rays:.2f} triangle tests per ray", flush=True)


run(s, "as loaded")
run(split(s, N), f"giant triangles in {N}x{N} pieces")
