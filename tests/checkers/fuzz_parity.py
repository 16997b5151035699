#!/usr/bin/env python3
"""Randomised parity run: random triangle soups (sizes, extents, offsets from the origin), random geometry
transforms applied through rt_models_update (device refit), random cameras and materials, and per case at random: spheres,
a run of frames in one launch (rt_dispatch_frames / rt_dispatch_frames_static), an rt_group of 2-4 shards, the other kernel
variants; HIP path against the oracle's BRUTE-FORCE mode, bit for bit.  usage: python tests/checkers/fuzz_parity.py [--iters 40] [--seed 1]"""
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
ap.add_argument("--big", action="store_true", help="bigger scenes and windows (use --max-tris 50000); the checker is then the oracle's own BVH mode, not brute force")
ap.add_argument("--only", type=int, default=-1, help="run just this iteration (the random stream is still advanced through the others), with details")
ap.add_argument("--force-aov", type=int, default=-1)
ap.add_argument("--force-kernel", type=int, default=-1)
args = ap.parse_args()
rt = graft.load_package()
rng = np.random.default_rng(args.seed)
bad = 0
total_rays = total_hits = 0
for it in range(args.iters):
    n_geo = int(rng.integers(1, 4)) if rng.random() < 0.9 else int(rng.integers(17, 21))   # > 16 models: the kernel instance that reads materials from memory
    verts, idx, prims, mods = [], [], [], []
    notes = set()
    centre = (rng.random(3) - 0.5) * 10.0 ** rng.integers(0, 5)
    spread = 10.0 ** rng.uniform(-1, 2)
    for g in range(n_geo):
        n = int(rng.integers(1, args.max_tris if n_geo < 4 else max(2, args.max_tris // 8)))
        size = spread * 10.0 ** rng.uniform(-2.2, 0.0)
        c = centre + (rng.random((n, 1, 3)) - 0.5) * 2 * spread
        p = (c + (rng.random((n, 3, 3)) - 0.5) * 2 * size).astype(np.float32)
        kind = rng.random()
        mesh_idx = None
        if kind < 0.2 and n >= 8:                        # a height field with SHARED vertices: rays through shared edges and corners
            gw = int(rng.integers(2, max(3, int(np.sqrt(n / 2)) + 2))); gh = max(2, n // (2 * (gw - 1)) + 1)
            gx, gz = np.meshgrid(np.linspace(-1, 1, gw), np.linspace(-1, 1, gh), indexing="xy")
            gy = (rng.random((gh, gw)) - 0.5) * rng.uniform(0, 0.6)
            grid = (centre + np.stack([gx, gy, gz], -1).reshape(-1, 3) * spread).astype(np.float32)
            q = np.arange((gh - 1) * (gw - 1)); r0 = q // (gw - 1) * gw + q % (gw - 1)
            mesh_idx = np.stack([r0, r0 + 1, r0 + gw, r0 + 1, r0 + gw + 1, r0 + gw], -1).reshape(-1).astype(np.uint32)
            notes.add('mesh')
        elif kind < 0.3:                                 # one giant quad under / through everything (C3's ground plane)
            e = spread * rng.uniform(5, 200)
            y = float(centre[1] - spread * rng.uniform(0, 1))
            quad = np.array([[-e, y, -e], [e, y, -e], [e, y, e], [-e, y, -e], [e, y, e], [-e, y, e]], np.float64) + [centre[0], 0, centre[2]]
            p = np.concatenate([p, quad.reshape(2, 3, 3).astype(np.float32)]); n += 2
            notes.add('giant quad')
        if n > 4 and rng.random() < 0.15:                # exact duplicates: equal t, the tie goes to the lower index-buffer position
            k = rng.integers(0, n, max(1, n // 10)); p[k] = p[rng.integers(0, n, len(k))]; notes.add('duplicates')
        if rng.random() < 0.1:                           # degenerate triangles (two equal corners): never hit
            k = rng.integers(0, n, max(1, n // 20)); p[k, 1] = p[k, 0]; notes.add('degenerate')
        if rng.random() < 0.02:                          # a non-finite corner: that triangle never hits, the rest of the scene does
            p[int(rng.integers(0, n)), int(rng.integers(0, 3)), int(rng.integers(0, 3))] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]; notes.add('non-finite')
        if mesh_idx is not None:
            v = np.zeros(len(grid), obj_loader.VERTEX_DTYPE)
            v["position"] = grid
            v["normal"] = (rng.random((len(grid), 3)) - 0.5).astype(np.float32) + (0, 1, 0)      # un-normalised, interpolated (main.wgsl:95)
            gi = mesh_idx
        else:
            v = np.zeros(3 * n, obj_loader.VERTEX_DTYPE)
            v["position"] = p.reshape(-1, 3)
            with np.errstate(all="ignore"):
                nr = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
                nr /= np.maximum(np.linalg.norm(nr, axis=1, keepdims=True), 1e-20)
            v["normal"] = np.repeat(np.nan_to_num(nr).astype(np.float32), 3, axis=0)
            gi = np.arange(3 * n, dtype=np.uint32)
        prims.append((sum(len(x) for x in verts), len(v), sum(len(x) for x in idx), len(gi), g))
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
        idx.append(gi)
    V = np.zeros(sum(len(x) for x in verts), obj_loader.VERTEX_DTYPE)
    o = 0
    for v in verts:
        V[o:o + len(v)] = v; o += len(v)
    I = np.concatenate(idx)
    P = np.zeros(n_geo, obj_loader.PRIMITIVE_DTYPE)
    for g, t in enumerate(prims): P[g] = t
    M = np.zeros(n_geo, obj_loader.MODEL_DTYPE)
    for g, m in enumerate(mods): M[g] = m[0]
    if n_geo > 1 and rng.random() < 0.3:                 # src/scene.rs:141-147: a geometry names its transform by offset
        P["transformation_offset"] = rng.permutation(n_geo) if rng.random() < 0.5 else rng.integers(0, n_geo, n_geo)
        notes.add('offsets')
    T = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (n_geo, 1))
    if rng.random() < 0.7:
        T += ((rng.random((n_geo, 12)) - 0.5) * 0.4).astype(np.float32)
        T[:, 3::4] += ((rng.random((n_geo, 3)) - 0.5) * spread).astype(np.float32)
    T0 = None
    if rng.random() < 0.3:                               # an earlier update first: two refits in a row
        T0 = T + ((rng.random((n_geo, 12)) - 0.5) * 0.2).astype(np.float32)
    if rng.random() < 0.1:                               # a long way: out of the 8-ary formats' corner grid / scale range -> rebuild on the host
        T[:, 3::4] += ((rng.random((n_geo, 3)) - 0.5) * spread * 60).astype(np.float32); notes.add('far move')
    refit = int(rng.choice([2, 2, 2, 1, 0]))             # RT_OPT_REFIT: device refit, host refit, rebuild
    cw = rng.random() < 0.2                              # RT_OPT_FRONT_FACE_CW
    w, h = (int(rng.integers(1, 70)), int(rng.integers(1, 50))) if not args.big else (int(rng.integers(64, 320)), int(rng.integers(48, 200)))
    cam = centre + (rng.random(3) - 0.5) * 3 * spread
    kw = dict(max_bounces=int(rng.integers(0, 6)), samples=int(rng.integers(1, 4)) if rng.random() < 0.97 else 0, fov=float(rng.uniform(0.3, 2.5)),
              flags=int(rng.integers(0, 2)), frame=int(rng.integers(1, 2**32)), exposure=float(rng.choice([1.0, rng.uniform(0.1, 4.0)])),
              environment=float(rng.choice([1.0, rng.uniform(0.0, 3.0)])))
    d = centre - cam                                     # look at the soup (ray.wgsl:52-58), with some jitter
    d = d / max(np.linalg.norm(d), 1e-20)
    yaw = float(np.arctan2(d[2], d[0]) + rng.uniform(-0.3, 0.3))
    pitch = float(np.clip(np.arcsin(d[1]) + rng.uniform(-0.3, 0.3), -1.5, 1.5))
    if rng.random() < 0.12:                              # a camera ALONG an axis with a narrow field of view: rays (nearly) parallel to one or two axes —
        yaw = float(int(rng.integers(0, 4)) * np.pi / 2)  # the case in which the box test's per-axis pads matter (slab_pad), and at pitch +-pi/2 a degenerate camera basis
        pitch = 0.0 if rng.random() < 0.7 else float(rng.choice([-1.0, 1.0]) * np.pi / 2)
        kw["fov"] = float(10.0 ** rng.uniform(-7, -0.5))
        fwd = np.array([np.cos(yaw) * np.cos(pitch), np.sin(pitch), np.sin(yaw) * np.cos(pitch)])
        cam = centre - fwd * spread * rng.uniform(1.5, 4.0) + (rng.random(3) - 0.5) * spread * 0.5
        notes.add('axis camera')
    textures = []
    if rng.random() < 0.25:                              # textures: index 0 = none, k = texture k - 1, beyond = black
        textures = [rng.integers(0, 256, (int(rng.integers(1, 10)), int(rng.integers(1, 10)), 4), dtype=np.uint8) for _ in range(int(rng.integers(1, 4)))]
        M["diffuse_texture"] = rng.integers(0, len(textures) + 2, n_geo)
        M["normal_texture"] = rng.integers(0, len(textures) + 2, n_geo)
        V["uv"] = (rng.random((len(V), 2)) * 1.4 - 0.2).astype(np.float32)          # some outside [0, 1]: clamp to edge
    scene = obj_loader.Scene(V, I, P, M, textures=textures)
    S = None
    if rng.random() < 0.35:                              # spheres (build-defined primitive)
        ns = int(rng.integers(1, 120))
        S = np.zeros(ns, obj_loader.SPHERE_DTYPE)
        S["center"] = (centre + (rng.random((ns, 3)) - 0.5) * 2 * spread).astype(np.float32)
        S["radius"] = (spread * 10.0 ** rng.uniform(-2.0, -0.5, ns)).astype(np.float32)
        S["model"] = rng.integers(0, n_geo, ns)
    n_frames = int(rng.integers(2, 4)) if rng.random() < 0.3 else 1
    shards = int(rng.integers(2, 5)) if rng.random() < 0.25 else 1
    kernel = int(rng.choice([2, 2, 2, 0, 1]))
    counting = rng.random() < 0.15                       # the counting instances of the kernels
    static = n_frames > 1 and shards == 1 and kernel == 2 and rng.random() < 0.5
    if args.only >= 0 and it != args.only:
        continue
    if args.force_kernel >= 0:
        kernel = args.force_kernel
    osc = oracle.OracleScene(scene, transforms=T, spheres=S, build_bvh=args.big)
    omode = 1 if args.big else 0
    oracle.lib().oracle_scene_set_front_face_cw(osc.handle, int(cw))
    sampler = int(rng.integers(0, 4)) if textures else 0      # RT_OPT_SAMPLER: nearest / linear x clamp / repeat
    osc.set_sampler(sampler)
    if n_frames == 1:
        want, want_hits, cnt = osc.render(oracle.make_uniform(w, h, cam, yaw, pitch, **kw), mode=omode, want_hits=True)
    else:                                                # the cold-start frame sequence of src/app.rs:78-96
        kw.pop("frame")
        want, cnt = osc.render_frames(oracle.make_uniform(w, h, cam, yaw, pitch, **kw), n_frames, mode=omode)
        want_hits = None
    with (rt.Group([0] * shards) if shards > 1 else rt.Context(0)) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, args.node_format)
        ctx.set_option(rt.RT_OPT_KERNEL, kernel)
        ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, int(counting))
        ctx.set_option(rt.RT_OPT_REFIT, refit)
        ctx.set_option(rt.RT_OPT_FRONT_FACE_CW, int(cw))
        ctx.set_option(rt.RT_OPT_SAMPLER, sampler)
        ctx.scene_upload(V, I, P, M)
        if S is not None:
            ctx.spheres_upload(S)
        if textures:
            ctx.textures_upload(textures)
        if T0 is not None:
            ctx.models_update(M, transforms=T0)
        ctx.models_update(M, transforms=T)               # device refit of the tree built for the untransformed scene
        aov = n_frames == 1 and shards == 1
        if args.force_aov >= 0 and shards == 1:
            aov = bool(args.force_aov)
        if aov:
            ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        u = rt.make_uniform(w, h, cam, yaw, pitch, **kw)
        if n_frames == 1:
            ctx.dispatch(u)
        else:
            u["frame"] = 0                               # cold start, src/main.rs:76-77
            u["accumulation_frame"] = 1
            (ctx.dispatch_frames_static if static else ctx.dispatch_frames)(u, n_frames)
        got, st = ctx.accum_download(), ctx.stats()
        hits = ctx.aov_download() if aov else None
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if counting and cnt.rays and (st.node_visits == 0 or st.wave_steps == 0):
        print('  counting instance counted nothing'); bad += 1
    ok = st.rays == cnt.rays and (hits is None or np.array_equal(hits, want_hits)) and bool(same.all())
    if args.only >= 0:
        diff = ~same.all(axis=-1)
        ys, xs = np.nonzero(diff)
        print(f"  rays device {st.rays} oracle {cnt.rays}; {int(diff.sum())} of {diff.size} pixels differ; first: {[(int(x), int(y)) for x, y in zip(xs[:8], ys[:8])]}")
        for x, y in list(zip(xs[:4], ys[:4])):
            print(f"   ({x},{y}): device {got[y, x]} oracle {want[y, x]}")
    if args.only >= 0 and args.big:                      # who is right?  ask brute force (slow)
        if n_frames == 1:
            bf, bf_hits, bcnt = osc.render(oracle.make_uniform(w, h, cam, yaw, pitch, **kw), mode=0, want_hits=True)
        else:
            bf, bcnt = osc.render_frames(oracle.make_uniform(w, h, cam, yaw, pitch, **kw), n_frames, mode=0)
        eq = lambda a, b: bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
        print(f"  brute force: rays {bcnt.rays}; device == brute force: {eq(got, bf)}; oracle BVH == brute force: {eq(want, bf)}")
    total_rays += int(cnt.rays)
    total_hits += int(cnt.hits)
    what = f"{len(I) // 3} tris in {n_geo} geometries" + (f" + {len(S)} spheres" if S is not None else "") + (f" + {len(textures)} textures" if textures else "") + \
           (f", {n_frames} frames" + (" (static)" if static else "") if n_frames > 1 else "") + (f", group of {shards}" if shards > 1 else "") + f", kernel {kernel}, refit {refit}" + (" counting" if counting else "") + (" cw" if cw else "") + (" 2 updates" if T0 is not None else "") + "".join(" " + x for x in sorted(notes))
    print(f"iter {it}: {what}, {w}x{h}, {kw}, rays {cnt.rays}, hits {cnt.hits}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print(f"fuzz: {args.iters} cases, {total_rays} rays, {total_hits} of them hit something;", "all identical" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
