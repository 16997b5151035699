"""
oracle.py — ctypes front end of oracle/rt_oracle.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import obj_loader

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_ALT_LIB_PATH = os.path.join(_HERE, "liboracle_alt.so")     # the same source under -DORACLE_ALT_MATH (rt_oracle.c header)
_libs = {}

UNIFORM_DTYPE = np.dtype(
    {"names": ["window", "position", "pitch", "yaw", "fov", "aspect", "frame", "accumulation_frame",
               "flags", "exposure", "enviroment", "max_bounces", "samples"],
     "formats": [(np.uint32, 2), (np.float32, 3), np.float32, np.float32, np.float32, np.float32,
                 np.uint32, np.uint32, np.uint32, np.float32, np.float32, np.uint32, np.uint32],
     "offsets": [0, 16, 28, 32, 36, 40, 48, 52, 56, 60, 64, 68, 72], "itemsize": 80})


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_visits", C.c_uint64), ("triangle_tests", C.c_uint64),
                ("hits", C.c_uint64), ("pixel_frames", C.c_uint64)]


def build(force: bool = False) -> str:
    """Compile rt_oracle.c → liboracle.so (gcc; RTN-1 flags)."""
    src = os.path.join(_HERE, "rt_oracle.c")
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src) for p in (_LIB_PATH, _ALT_LIB_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def lib(alt: bool = False):
    """liboracle.so (the RTN-1 contract: the checker of bits) or, alt=True, liboracle_alt.so (the tolerance yardstick)."""
    if alt not in _libs:
        path = _ALT_LIB_PATH if alt else _LIB_PATH
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_int]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_scene_set_front_face_cw.argtypes = [C.c_void_p, C.c_int]
        L.oracle_scene_set_sampler.argtypes = [C.c_void_p, C.c_int]
        L.oracle_sample_rgb.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.oracle_scene_set_textures.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.oracle_scene_triangles.restype = C.c_uint64
        L.oracle_scene_triangles.argtypes = [C.c_void_p]
        L.oracle_threads.restype = C.c_int
        L.oracle_scene_nodes.restype = C.c_uint64
        L.oracle_scene_nodes.argtypes = [C.c_void_p]
        L.oracle_render_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_int, C.c_void_p]
        L.oracle_render_rect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_int, C.c_void_p]
        L.oracle_trace.restype = C.c_int
        L.oracle_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_second_nearest.restype = C.c_float
        L.oracle_second_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]
        L.oracle_primary_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_rand.restype = C.c_float
        L.oracle_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.oracle_seed.restype = C.c_uint32
        L.oracle_seed.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_capture_rgb8.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_render_rect_hash.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_int, C.c_void_p]
        L.oracle_variant.restype = C.c_int
        assert L.oracle_variant() == (1 if alt else 0)
        _libs[alt] = L
    return _libs[alt]


def make_uniform(width, height, position, yaw, pitch=0.0, fov=np.pi / 2, aspect=None, frame=1,
                 accumulation_frame=0, flags=0, exposure=1.0, environment=1.0, max_bounces=10, samples=5):
    """Uniform with the reference's defaults (src/main.rs:73-84); aspect = W/H (src/camera.rs:56)."""
    u = np.zeros(1, UNIFORM_DTYPE)
    u["window"][0] = (width, height)
    u["position"][0] = position
    u["pitch"] = np.float32(pitch)
    u["yaw"] = np.float32(yaw)
    u["fov"] = np.float32(fov)
    u["aspect"] = np.float32(width) / np.float32(height) if aspect is None else np.float32(aspect)
    u["frame"] = frame
    u["accumulation_frame"] = accumulation_frame
    u["flags"] = flags
    u["exposure"] = exposure
    u["enviroment"] = environment
    u["max_bounces"] = max_bounces
    u["samples"] = samples
    return u


class OracleScene:
    """Holds numpy scene arrays alive and the C-side triangle list / BVH."""

    def __init__(self, scene: obj_loader.Scene, transforms=None, spheres=None, build_bvh=True, alt=False):
        self.scene = scene
        self.alt = alt
        self.vertices = np.ascontiguousarray(scene.vertices)
        self.indices = np.ascontiguousarray(scene.indices, dtype=np.uint32)
        self.primitives = np.ascontiguousarray(scene.primitives)
        self.models = np.ascontiguousarray(scene.models)
        self.transforms = None if transforms is None else np.ascontiguousarray(transforms, dtype=np.float32)
        self.spheres = None if spheres is None else np.ascontiguousarray(spheres)
        L = lib(alt)
        self.handle = L.oracle_scene_create(
            self.vertices.ctypes.data, len(self.vertices), self.indices.ctypes.data, len(self.indices),
            self.primitives.ctypes.data, len(self.primitives),
            None if self.transforms is None else self.transforms.ctypes.data,
            self.models.ctypes.data, len(self.models),
            None if self.spheres is None else self.spheres.ctypes.data,
            0 if self.spheres is None else len(self.spheres), 1 if build_bvh else 0)
        # textures: list of [h, w, 4] uint8 images in FILE row order; flipped here as src/scene.rs:77 does at upload
        tex = getattr(scene, "textures", None) or []
        self._tex = [np.ascontiguousarray(t[::-1], np.uint8) for t in tex]
        if self._tex:
            self._tex_ptrs = (C.c_void_p * len(self._tex))(*[t.ctypes.data for t in self._tex])
            self._tex_w = np.array([t.shape[1] for t in self._tex], np.uint32)
            self._tex_h = np.array([t.shape[0] for t in self._tex], np.uint32)
            L.oracle_scene_set_textures(self.handle, self._tex_ptrs, self._tex_w.ctypes.data, self._tex_h.ctypes.data, len(self._tex))

    def sample_rgb(self, texture: int, u: float, v: float):
        out = np.zeros(3, np.float32)
        lib(self.alt).oracle_sample_rgb(self.handle, texture, C.c_float(u), C.c_float(v), out.ctypes.data)
        return out

    def set_sampler(self, sampler: int):
        """RT_OPT_SAMPLER of the product: bit 0 linear filter, bit 1 repeat addressing (0 = nearest, clamp to edge)."""
        lib(self.alt).oracle_scene_set_sampler(self.handle, int(sampler))

    def __del__(self):
        if getattr(self, "handle", None):
            lib(self.alt).oracle_scene_destroy(self.handle)
            self.handle = None

    @property
    def n_triangles(self):
        return lib(self.alt).oracle_scene_triangles(self.handle)

    @property
    def n_nodes(self):
        return lib(self.alt).oracle_scene_nodes(self.handle)

    def render(self, uniform, accum=None, mode=1, rows=None, want_hits=False, cols=None, path_hash=None):
        """One dispatch (shaders/main.wgsl:14-33) into `accum` ([H, W, 4] f32, updated in place); rows / cols restrict it to a
        rectangle of pixels (the others keep their values).  path_hash: [H, W] uint32 array that receives a hash of the hit ids
        of all rays of each pixel-frame (which pixels took the same paths in two renders)."""
        u = np.ascontiguousarray(uniform)
        W, H = int(u["window"][0][0]), int(u["window"][0][1])
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        assert accum.shape == (H, W, 4) and accum.dtype == np.float32 and accum.flags.c_contiguous
        hits = np.full((H, W, 2), 0xFFFFFFFF, np.uint32) if want_hits else None
        cnt = Counters()
        y0, y1 = rows if rows is not None else (0, H)
        x0, x1 = cols if cols is not None else (0, W)
        if path_hash is not None:
            assert path_hash.shape == (H, W) and path_hash.dtype == np.uint32 and path_hash.flags.c_contiguous
        lib(self.alt).oracle_render_rect_hash(self.handle, u.ctypes.data, accum.ctypes.data, None if hits is None else hits.ctypes.data,
                                              None if path_hash is None else path_hash.ctypes.data, x0, x1, y0, y1, mode, C.byref(cnt))
        return accum, hits, cnt

    def render_frames(self, uniform, n_frames, mode=1):
        """Cold-start frame sequence of src/app.rs:78-96: frame = 1.., accumulation_frame = 0.."""
        u = np.array(uniform, copy=True)
        W, H = int(u["window"][0][0]), int(u["window"][0][1])
        accum = np.zeros((H, W, 4), np.float32)
        total = Counters()
        for f in range(n_frames):
            u["frame"] = f + 1
            u["accumulation_frame"] = f
            _, _, c = self.render(u, accum, mode)
            for name, _t in Counters._fields_:
                setattr(total, name, getattr(total, name) + getattr(c, name))
        return accum, total

    def trace(self, origin, direction, cull=False, mode=0):
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        out = np.zeros(3, np.float32)
        ids = np.zeros(3, np.uint32)
        hit = lib(self.alt).oracle_trace(self.handle, o.ctypes.data, d.ctypes.data, int(cull), mode,
                                 out.ctypes.data, ids.ctypes.data)
        return (bool(hit), out, ids)

    def second_nearest(self, origin, direction, geometry, primitive, cull=False):
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        return float(lib(self.alt).oracle_second_nearest(self.handle, o.ctypes.data, d.ctypes.data, int(cull),
                                                 int(geometry), int(primitive)))


def primary_ray(uniform, x, y):
    u = np.ascontiguousarray(uniform)
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    lib().oracle_primary_ray(u.ctypes.data, x, y, o.ctypes.data, d.ctypes.data)
    return o, d


def rand_sequence(pixel_idx, frame, n):
    """seed + n LCG states + floats (shaders/random.wgsl:3-7, shaders/main.wgsl:23)."""
    L = lib()
    s = C.c_uint32(L.oracle_seed(pixel_idx, frame))
    seed0 = s.value
    states, floats = [], []
    for _ in range(n):
        f = L.oracle_rand(C.byref(s))
        states.append(s.value)
        floats.append(np.float32(f))
    return seed0, states, floats


def sincos(x):
    s = C.c_float()
    c = C.c_float()
    lib().oracle_sincos(C.c_float(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def capture_rgb8(accum):
    a = np.ascontiguousarray(accum, np.float32).reshape(-1, 4)
    out = np.zeros((a.shape[0], 3), np.uint8)
    lib().oracle_capture_rgb8(a.ctypes.data, a.shape[0], out.ctypes.data)
    return out
