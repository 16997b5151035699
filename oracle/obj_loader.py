"""
obj_loader.py — Python restatement of the reference's scene loading.

TEST INFRASTRUCTURE ONLY (see oracle/rt_oracle.c header).  Follows
/root/reference/src/scene.rs:95-207 (Scene::load) and
/root/reference/src/misc.rs:49-75 (Pm / Pr / Ke), plus the documented
behaviour of the third-party parser the reference calls there: tobj 4.0.3
(Cargo.lock:3057-3058) with `triangulate: true, single_index: true`
(src/scene.rs:100-107).  tobj is not vendored in /root/reference, so its
published algorithm is restated:

  * a model is the run of faces under one `o`/`g` name; a `usemtl` that changes
    the material after faces were read also closes the current model;
  * polygons are fan-triangulated in file order: (v0, v[i], v[i+1]);
  * single_index: vertices are de-duplicated per model on the (v, vt, vn)
    index triple, numbered in first-use order; texcoords are emitted only for
    vertices that carry a vt index;
  * indices are model-local; negative OBJ indices count back from the end;
  * MTL: Kd -> diffuse, Ks -> specular, map_Kd -> diffuse_texture,
    map_Bump/map_bump/bump -> normal_texture, everything unrecognised (Ke, Pm,
    Pr, ...) lands in `unknown_param` as the raw string after the key.

Parity unpinned against the running reference (no Rust toolchain here); pinned
by triangle/vertex counts recorded in SURVEY.md §8 and by the C++ loader of the
product having to agree bit-for-bit (tests/test_scene_loader.py).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

# WGSL/encase layouts, restated from shaders/types.wgsl (not from include/rt_abi.h)
VERTEX_DTYPE = np.dtype(
    {"names": ["position", "normal", "uv"],
     "formats": [(np.float32, 3), (np.float32, 3), (np.float32, 2)],
     "offsets": [0, 16, 32], "itemsize": 48})
METAL_FIELDS = {
    "names": ["tag", "diffuse_color", "specular_color", "specular_probability", "roughness",
              "emission_color", "emission_strength", "diffuse_texture", "normal_texture",
              "refractive_index", "vertex_start", "index_start"],
    "formats": [np.uint32, (np.float32, 3), (np.float32, 3), np.float32, np.float32,
                (np.float32, 3), np.float32, np.uint32, np.uint32,
                np.float32, np.uint32, np.uint32],
    # Material.tag @0; metal @16: diffuse @16, specular @32, sp @44, rough @48,
    # emission_color @64, strength @76, textures @80/@84; dielectric @96;
    # vertex_start @112, index_start @116; size 128
    "offsets": [0, 16, 32, 44, 48, 64, 76, 80, 84, 96, 112, 116],
    "itemsize": 128}
MODEL_DTYPE = np.dtype(METAL_FIELDS)
PRIMITIVE_DTYPE = np.dtype(
    {"names": ["first_vertex", "vertex_count", "first_index", "index_count", "transformation_offset"],
     "formats": [np.uint32, np.uint32, np.uint32, np.uint32, np.uint64],
     "offsets": [0, 4, 8, 12, 16], "itemsize": 24})
SPHERE_DTYPE = np.dtype(
    {"names": ["center", "radius", "model"],
     "formats": [(np.float32, 3), np.float32, np.uint32],
     "offsets": [0, 12, 16], "itemsize": 32})


@dataclass
class Scene:
    """≙ struct Scene, src/scene.rs:20-27 ."""
    vertices: np.ndarray
    indices: np.ndarray
    primitives: np.ndarray
    models: np.ndarray
    names: list = field(default_factory=list)
    textures: list = field(default_factory=list)   # [h, w, 4] uint8 RGBA, file row order (src/scene.rs:23)


def _f32(tok: str) -> np.float32:
    # Rust `parse::<f32>()` rounds the decimal string once to binary32; going
    # through binary64 first can differ only for strings within 2^-53 relative
    # of a binary32 midpoint, which short decimal OBJ literals are not.
    return np.float32(float(tok))


def _load_mtl(path: str) -> tuple[list[dict], dict[str, int]]:
    mats: list[dict] = []
    by_name: dict[str, int] = {}
    cur = None
    with open(path, "r") as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            parts = line.split(None, 1)
            key = parts[0]
            rest = parts[1].strip() if len(parts) > 1 else ""
            if key == "newmtl":
                cur = {"name": rest, "diffuse": None, "specular": None,
                       "diffuse_texture": None, "normal_texture": None, "unknown": {}}
                by_name[rest] = len(mats)
                mats.append(cur)
            elif cur is None:
                continue
            elif key == "Kd":
                cur["diffuse"] = [_f32(t) for t in rest.split()[:3]]
            elif key == "Ks":
                cur["specular"] = [_f32(t) for t in rest.split()[:3]]
            elif key == "map_Kd":
                cur["diffuse_texture"] = rest
            elif key in ("map_Bump", "map_bump", "bump"):
                cur["normal_texture"] = rest
            elif key in ("Ka", "Ns", "Ni", "d", "illum", "map_Ka", "map_Ks", "map_Ns", "map_d"):
                pass  # known tobj fields the reference never reads
            else:
                cur["unknown"][key] = rest
    return mats, by_name


def strip_flags(path: str) -> str:
    """src/scene.rs:210-223: drop leading `-flag value` pairs of an MTL map statement."""
    i = 0
    while path[i:].startswith("-"):
        for _ in range(2):
            while not path[i:].startswith(" "):
                i += 1
            i += 1
    return path[i:]


def _load_image(path: str) -> np.ndarray:
    """image::load(...).into_rgba8() (src/scene.rs:163-164), via PIL here.  8-bit-or-less PNGs only: PIL reduces 16-bit samples by
    truncation, the image crate by rounding ((c + 128) / 257, what host/image_decode.cpp does; tests/test_textures.py checks that rule)."""
    from PIL import Image
    return np.array(Image.open(path).convert("RGBA"), np.uint8)


def _unknown_f32(mat: dict, key: str) -> np.float32:
    """src/misc.rs:53-59: missing or unparsable → 0."""
    s = mat["unknown"].get(key)
    if s is None:
        return np.float32(0)
    try:
        return _f32(s)
    except ValueError:
        return np.float32(0)


def _unknown_vec3(mat: dict, key: str) -> np.ndarray:
    """src/misc.rs:62-75: missing → zeros."""
    s = mat["unknown"].get(key)
    if s is None:
        return np.zeros(3, np.float32)
    return np.array([_f32(t) for t in s.split()], np.float32)[:3]


def load(path: str) -> Scene:
    """≙ Scene::load, src/scene.rs:95-207."""
    pos: list[tuple] = []
    nrm: list[tuple] = []
    tex: list[tuple] = []
    mats: list[dict] = []
    mat_by_name: dict[str, int] = {}

    models_out = []  # (name, material_id, verts[(p,t,n)], indices)

    cur_name = "unnamed_object"
    cur_mat = None
    cur_faces: list[list[tuple]] = []

    def flush():
        nonlocal cur_faces
        if not cur_faces:
            return
        index_map: dict[tuple, int] = {}
        verts: list[tuple] = []
        idx: list[int] = []

        def add(v):
            i = index_map.get(v)
            if i is None:
                i = len(verts)
                index_map[v] = i
                verts.append(v)
            idx.append(i)

        for f in cur_faces:
            if len(f) < 3:
                raise ValueError("points/lines are not supported (none in the reference scenes)")
            a = f[0]
            b = f[1]
            for c in f[2:]:
                add(a); add(b); add(c)
                b = c
        models_out.append((cur_name, cur_mat, verts, idx))
        cur_faces = []

    def fix(i: int, n: int) -> int:
        return i - 1 if i > 0 else n + i

    with open(path, "r") as fh:
        for line in fh:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            parts = line.split()
            key = parts[0]
            if key == "v":
                pos.append(tuple(_f32(t) for t in parts[1:4]))
            elif key == "vn":
                nrm.append(tuple(_f32(t) for t in parts[1:4]))
            elif key == "vt":
                tex.append(tuple(_f32(t) for t in parts[1:3]))
            elif key == "f":
                face = []
                for tok in parts[1:]:
                    sub = tok.split("/")
                    vi = fix(int(sub[0]), len(pos))
                    ti = fix(int(sub[1]), len(tex)) if len(sub) > 1 and sub[1] else -1
                    ni = fix(int(sub[2]), len(nrm)) if len(sub) > 2 and sub[2] else -1
                    face.append((vi, ti, ni))
                cur_faces.append(face)
            elif key in ("o", "g"):
                flush()
                cur_name = line[len(key):].strip() or "unnamed_object"
            elif key == "usemtl":
                name = line[len(key):].strip()
                new_mat = mat_by_name.get(name)
                if new_mat != cur_mat and cur_faces:
                    flush()
                cur_mat = new_mat
            elif key == "mtllib":
                m, byn = _load_mtl(os.path.join(os.path.dirname(path), line[len(key):].strip()))
                base = len(mats)
                mats.extend(m)
                for k, v in byn.items():
                    mat_by_name[k] = base + v
    flush()

    all_verts = []
    all_idx = []
    prims = []
    models = []
    names = []
    textures = []

    def load_texture(rel):                                                # src/scene.rs:157-175
        if rel is None:
            return 0
        textures.append(_load_image(os.path.join(os.path.dirname(path), strip_flags(rel))))
        return len(textures)
    for name, mat_id, verts, idx in models_out:
        first_vertex = len(all_verts)
        first_index = len(all_idx)
        has_tex = all(t >= 0 for (_, t, _) in verts) and len(verts) > 0
        for (p, t, n) in verts:
            if n < 0:
                raise ValueError("vertex without a normal: the reference's zip() would drop it (src/scene.rs:121-124)")
            uv = tex[t] if (t >= 0 and has_tex) else (np.float32(0), np.float32(0))  # src/scene.rs:127-130
            all_verts.append((pos[p], nrm[n], uv))
        all_idx.extend(idx)
        prims.append((first_vertex, len(verts), first_index, len(idx), len(prims)))  # src/scene.rs:141-147
        if mat_id is None:
            raise ValueError(f"object {name!r} has no material (material_id.unwrap(), src/scene.rs:149)")
        mat = mats[mat_id]
        diffuse = np.array(mat["diffuse"] or [0, 0, 0], np.float32)       # :151
        specular = np.array(mat["specular"] or [0, 0, 0], np.float32)     # :152
        sp = _unknown_f32(mat, "Pm")                                      # :153
        rough = _unknown_f32(mat, "Pr")                                   # :154
        ke = _unknown_vec3(mat, "Ke")                                     # :155
        # nalgebra: norm = sqrt(x*x + y*y + z*z) in f32; try_normalize(0.0)
        n2 = np.float32(np.float32(ke[0] * ke[0] + ke[1] * ke[1]) + ke[2] * ke[2])
        norm = np.sqrt(n2, dtype=np.float32)
        em_color = (ke / norm).astype(np.float32) if norm > 0 else np.zeros(3, np.float32)  # :191
        diffuse_texture = load_texture(mat["diffuse_texture"])            # :177
        normal_texture = load_texture(mat["normal_texture"])              # :178
        models.append((0, diffuse, specular, sp, rough, em_color, norm, diffuse_texture, normal_texture,
                       np.float32(0),  # DielectricMaterial::default() → 0.0 (src/types.rs:107)
                       first_vertex, first_index))
        names.append(name)

    v = np.zeros(len(all_verts), VERTEX_DTYPE)
    for i, (p, n, uv) in enumerate(all_verts):
        v["position"][i] = p
        v["normal"][i] = n
        v["uv"][i] = uv
    pr = np.zeros(len(prims), PRIMITIVE_DTYPE)
    for i, t in enumerate(prims):
        pr[i] = t
    md = np.zeros(len(models), MODEL_DTYPE)
    for i, t in enumerate(models):
        for name_, val in zip(MODEL_DTYPE.names, t):
            md[name_][i] = val
    return Scene(v, np.array(all_idx, np.uint32), pr, md, names, textures)
