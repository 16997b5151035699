#!/usr/bin/env python3
"""Randomised SESSIONS: one context (or rt_group) lives through a random sequence of calls — scene uploads and re-uploads, sphere
and texture uploads, rt_models_update, accumulation-buffer resizes, camera moves, option changes (kernel variant, counting, AOV,
frame pipeline, refit mode, node format), rt_dispatch / rt_dispatch_frames / rt_dispatch_frames_static — while the oracle's
brute force keeps a shadow accumulation buffer through the same calls (main.wgsl:27-32: the running mean across dispatches).
After every dispatch the buffers must hold the same bits and the ray counts must agree.  What fuzz_parity.py cannot see —
state carried from one call to the next — is what this one is for.
usage: python tests/checkers/fuzz_session.py [--sessions 40] [--seed 1] [--node-format 0]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
from oracle import obj_loader, oracle

ap = argparse.ArgumentParser()
ap.add_argument("--sessions", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--node-format", type=int, default=0)
ap.add_argument("--verbose", action="store_true")
args = ap.parse_args()
rt = graft.load_package()
rng = np.random.default_rng(args.seed)
IDENT = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


def draw_material(m):
    m["tag"] = 1 if rng.random() < 0.25 else 0
    m["diffuse_color"] = rng.random(3)
    m["specular_color"] = rng.random(3)
    m["specular_probability"] = rng.random()
    m["roughness"] = rng.random()
    m["emission_color"] = rng.random(3)
    m["emission_strength"] = rng.random() * (rng.random() < 0.3)
    m["refractive_index"] = 1.0 + rng.random()


def draw_scene():
    n_geo = int(rng.integers(1, 4)) if rng.random() < 0.9 else int(rng.integers(17, 20))
    centre = (rng.random(3) - 0.5) * 10.0 ** rng.integers(0, 4)
    spread = 10.0 ** rng.uniform(-1, 2)
    verts, idx = [], []
    P = np.zeros(n_geo, obj_loader.PRIMITIVE_DTYPE)
    M = np.zeros(n_geo, obj_loader.MODEL_DTYPE)
    for g in range(n_geo):
        n = int(rng.integers(1, 500 if n_geo < 4 else 60))
        size = spread * 10.0 ** rng.uniform(-2.0, 0.0)
        c = centre + (rng.random((n, 1, 3)) - 0.5) * 2 * spread
        p = (c + (rng.random((n, 3, 3)) - 0.5) * 2 * size).astype(np.float32)
        v = np.zeros(3 * n, obj_loader.VERTEX_DTYPE)
        v["position"] = p.reshape(-1, 3)
        v["normal"] = (rng.random((3 * n, 3)) - 0.5).astype(np.float32)
        v["uv"] = (rng.random((3 * n, 2)) * 1.4 - 0.2).astype(np.float32)
        P[g] = (sum(len(x) for x in verts), 3 * n, sum(len(x) for x in idx), 3 * n, g)
        draw_material(M[g:g + 1])
        M[g]["vertex_start"], M[g]["index_start"] = P[g]["first_vertex"], P[g]["first_index"]
        verts.append(v)
        idx.append(np.arange(3 * n, dtype=np.uint32))
    V = np.zeros(sum(len(x) for x in verts), obj_loader.VERTEX_DTYPE)
    o = 0
    for v in verts:
        V[o:o + len(v)] = v
        o += len(v)
    S = None
    if rng.random() < 0.35:
        ns = int(rng.integers(1, 60))
        S = np.zeros(ns, obj_loader.SPHERE_DTYPE)
        S["center"] = (centre + (rng.random((ns, 3)) - 0.5) * 2 * spread).astype(np.float32)
        S["radius"] = (spread * 10.0 ** rng.uniform(-2.0, -0.5, ns)).astype(np.float32)
        S["model"] = rng.integers(0, n_geo, ns)
    textures = []
    if rng.random() < 0.25:
        textures = [rng.integers(0, 256, (int(rng.integers(1, 8)), int(rng.integers(1, 8)), 4), dtype=np.uint8) for _ in range(int(rng.integers(1, 3)))]
        M["diffuse_texture"] = rng.integers(0, len(textures) + 2, n_geo)
        M["normal_texture"] = rng.integers(0, len(textures) + 2, n_geo)
    return dict(V=V, I=np.concatenate(idx), P=P, M=M, S=S, textures=textures, centre=centre, spread=spread, n_geo=n_geo)


def draw_transforms(sc):
    T = np.tile(IDENT, (sc["n_geo"], 1))
    T += ((rng.random((sc["n_geo"], 12)) - 0.5) * 0.4).astype(np.float32)
    T[:, 3::4] += ((rng.random((sc["n_geo"], 3)) - 0.5) * sc["spread"]).astype(np.float32)
    return T


def draw_camera(sc):
    cam = sc["centre"] + (rng.random(3) - 0.5) * 3 * sc["spread"]
    d = sc["centre"] - cam
    d = d / max(np.linalg.norm(d), 1e-20)
    return dict(position=cam, yaw=float(np.arctan2(d[2], d[0]) + rng.uniform(-0.3, 0.3)), pitch=float(np.clip(np.arcsin(d[1]) + rng.uniform(-0.3, 0.3), -1.5, 1.5)),
                fov=float(rng.uniform(0.3, 2.5)), max_bounces=int(rng.integers(0, 5)), samples=int(rng.integers(1, 3)), flags=int(rng.integers(0, 2)),
                exposure=float(rng.choice([1.0, rng.uniform(0.2, 3.0)])), environment=float(rng.choice([1.0, rng.uniform(0.0, 2.0)])))


def uniform(mod, w, h, cam, frame, acc):
    return mod.make_uniform(w, h, cam["position"], cam["yaw"], cam["pitch"], fov=cam["fov"], frame=frame, accumulation_frame=acc, flags=cam["flags"],
                            exposure=cam["exposure"], environment=cam["environment"], max_bounces=cam["max_bounces"], samples=cam["samples"])


bad = 0
n_dispatch = 0
for session in range(args.sessions):
    shards = int(rng.integers(2, 4)) if rng.random() < 0.25 else 1
    log = []
    ok = True
    with (rt.Group([0] * shards) if shards > 1 else rt.Context(0)) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, args.node_format)
        opt = {"kernel": 2, "aov": 0, "count": 0, "pipeline": 1}
        sc = cam = osc = accum = None
        M = T = None
        w = h = 0
        fresh = True
        frame_ctr, acc_ctr = 0, 1                        # the caller-side uniform of rt_dispatch_frames (cold start, src/main.rs:76-77)
        rays_expected = 0
        pending = ["upload", "resize"]
        for step in range(int(rng.integers(6, 18))):
            op = pending.pop(0) if pending else str(rng.choice(["dispatch", "dispatch", "frames", "frames", "frames", "update", "update", "upload", "resize", "option", "camera"]))
            if op == "upload":
                sc = draw_scene()
                M, T = sc["M"].copy(), None
                ctx.scene_upload(sc["V"], sc["I"], sc["P"], M)
                if sc["S"] is not None:
                    ctx.spheres_upload(sc["S"])
                if sc["textures"]:
                    ctx.textures_upload(sc["textures"])
                cam = draw_camera(sc)
                osc = None
                log.append(f"upload({len(sc['I']) // 3} tris, {sc['n_geo']} geometries{', spheres' if sc['S'] is not None else ''}{', textures' if sc['textures'] else ''})")
            elif op == "update":
                M = M.copy()
                for g in range(sc["n_geo"]):
                    if rng.random() < 0.5:
                        draw_material(M[g:g + 1])
                moved = rng.random() < 0.8                # transforms = NULL keeps the transforms in force (rt_abi.h rt_models_update)
                if moved:
                    T = draw_transforms(sc)
                ctx.models_update(M, transforms=T if moved else None)
                osc = None
                log.append("update(T)" if moved else "update(materials)")
            elif op == "resize":
                w, h = int(rng.integers(1, 50)), int(rng.integers(1, 40))
                ctx.accum_resize(w, h)
                accum = np.zeros((h, w, 4), np.float32)
                fresh = True
                log.append(f"resize({w}x{h})")
            elif op == "camera":
                cam = draw_camera(sc)
                log.append("camera")
            elif op == "option":
                k = str(rng.choice(["kernel", "aov", "count", "pipeline", "format", "refit", "stage", "reserve"]))
                if k == "kernel":
                    opt["kernel"] = int(rng.integers(0, 3)); ctx.set_option(rt.RT_OPT_KERNEL, opt["kernel"])
                elif k == "aov":
                    opt["aov"] ^= 1; ctx.set_option(rt.RT_OPT_AOV, opt["aov"])
                elif k == "count":
                    opt["count"] ^= 1; ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, opt["count"])
                elif k == "pipeline":
                    opt["pipeline"] ^= 1; ctx.set_option(rt.RT_OPT_FRAME_PIPELINE, opt["pipeline"])
                elif k == "stage":                      # developer knob 109: staging budget in KiB -> runs of frames split into several launches
                    ctx.set_option(109, int(rng.choice([1, 4, 16, 64, 4 << 20])))
                elif k == "reserve":
                    if shards == 1 and w:
                        ctx.frames_reserve(int(rng.integers(1, 6)))
                elif k == "format":
                    ctx.set_option(rt.RT_OPT_NODE_FORMAT, int(rng.integers(0, 5)))      # takes effect at the next upload / rebuild
                else:
                    ctx.set_option(rt.RT_OPT_REFIT, int(rng.integers(0, 3)))
                log.append(f"option({k} -> {opt})")
            else:
                if osc is None:
                    scene = obj_loader.Scene(sc["V"], sc["I"], sc["P"], M, textures=sc["textures"])
                    osc = oracle.OracleScene(scene, transforms=T, spheres=sc["S"], build_bvh=False)
                if op == "dispatch":
                    frame, acc = int(rng.integers(1, 2**32)), int(rng.integers(0, 6))
                    if opt["kernel"] != 1 and rng.random() < 0.4:     # the frame in pieces (rt_dispatch_bands; a group: every shard its own bands of the range), in random order
                        nb = -(-((h + 7) // 8) // shards)
                        cut = sorted(set([0, nb] + [int(c) for c in rng.integers(0, nb + 1, int(rng.integers(0, 4)))]))
                        pieces = list(zip(cut[:-1], cut[1:]))
                        for k in rng.permutation(len(pieces)):
                            ctx.dispatch_bands(uniform(rt, w, h, cam, frame, acc), pieces[k][0], pieces[k][1] if pieces[k][1] < nb or rng.random() < 0.5 else 0xFFFFFFFF)
                        log.append(f"pieces {pieces}")
                    else:
                        ctx.dispatch(uniform(rt, w, h, cam, frame, acc))
                    _, _, cnt = osc.render(uniform(oracle, w, h, cam, frame, acc), accum, mode=0)
                    rays_expected += int(cnt.rays)
                    fresh = False
                    log.append(f"dispatch(frame {frame}, accumulation_frame {acc})")
                else:
                    n = int(rng.integers(1, 6))
                    static = shards == 1 and not opt["aov"] and rng.random() < 0.4
                    u = uniform(rt, w, h, cam, frame_ctr, acc_ctr)
                    (ctx.dispatch_frames_static if static else ctx.dispatch_frames)(u, n)
                    for _ in range(n):                   # the bookkeeping of src/app.rs:78-96, as rt_dispatch_frames does it
                        acc_ctr += 1
                        if fresh:
                            acc_ctr, fresh = 0, False
                        frame_ctr += 1
                        _, _, cnt = osc.render(uniform(oracle, w, h, cam, frame_ctr, acc_ctr), accum, mode=0)
                        rays_expected += int(cnt.rays)
                    if int(u["frame"][0]) != frame_ctr or int(u["accumulation_frame"][0]) != acc_ctr:
                        ok = False
                        log.append(f"UNIFORM device ({int(u['frame'][0])}, {int(u['accumulation_frame'][0])}) expected ({frame_ctr}, {acc_ctr})")
                    log.append(f"frames({n}{', static' if static else ''})")
                n_dispatch += 1
                got = ctx.accum_download()
                same = (got.view(np.uint32) == accum.view(np.uint32)) | (np.isnan(got) & np.isnan(accum))
                rays = int(ctx.stats().rays)
                if not same.all() or rays != rays_expected:
                    ok = False
                    log.append(f"MISMATCH: {int((~same.all(axis=-1)).sum())} of {w * h} pixels differ, rays {rays} expected {rays_expected}")
                    break
                if rng.random() < 0.2:
                    rgb = ctx.capture_rgb8().reshape(-1, 3)
                    if not np.array_equal(rgb, oracle.capture_rgb8(accum)):
                        ok = False
                        log.append("MISMATCH in capture_rgb8")
                        break
    if not ok or args.verbose:
        print(f"session {session} ({'group of ' + str(shards) if shards > 1 else 'context'}): " + "; ".join(log), flush=True)
    bad += 0 if ok else 1
print(f"sessions: {args.sessions}, {n_dispatch} dispatch calls checked;", "all identical" if bad == 0 else f"{bad} sessions with mismatches")
sys.exit(1 if bad else 0)
