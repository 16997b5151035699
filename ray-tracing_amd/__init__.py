"""
ray-tracing_amd — ctypes binding of librt_amd.so, the MI355X path-tracing backend.

The product is the C-ABI shared library (include/rt_abi.h); this module only
loads it and gives tests, bench.py and smoke() a thin object wrapper.  There is
no fallback: if the library (or a GPU) is missing the calls raise.

The directory name has a hyphen, so import it through `load_package()` in
__graft_entry__.py (or importlib with an explicit spec), as `ray_tracing_amd`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: developer override to A/B another build of the same library (tools/)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

# ---------------------------------------------------------------- layouts
# numpy views of the structs in include/rt_abi.h (offsets asserted there)
UNIFORM_DTYPE = np.dtype(
    {"names": ["window", "position", "pitch", "yaw", "fov", "aspect", "frame", "accumulation_frame",
               "flags", "exposure", "environment", "max_bounces", "samples"],
     "formats": [(np.uint32, 2), (np.float32, 3), np.float32, np.float32, np.float32, np.float32,
                 np.uint32, np.uint32, np.uint32, np.float32, np.float32, np.uint32, np.uint32],
     "offsets": [0, 16, 28, 32, 36, 40, 48, 52, 56, 60, 64, 68, 72], "itemsize": 80})
VERTEX_DTYPE = np.dtype(
    {"names": ["position", "normal", "uv"],
     "formats": [(np.float32, 3), (np.float32, 3), (np.float32, 2)],
     "offsets": [0, 16, 32], "itemsize": 48})
MODEL_DTYPE = np.dtype(
    {"names": ["tag", "diffuse_color", "specular_color", "specular_probability", "roughness",
               "emission_color", "emission_strength", "diffuse_texture", "normal_texture",
               "refractive_index", "vertex_start", "index_start"],
     "formats": [np.uint32, (np.float32, 3), (np.float32, 3), np.float32, np.float32,
                 (np.float32, 3), np.float32, np.uint32, np.uint32, np.float32, np.uint32, np.uint32],
     "offsets": [0, 16, 32, 44, 48, 64, 76, 80, 84, 96, 112, 116], "itemsize": 128})
PRIMITIVE_DTYPE = np.dtype(
    {"names": ["first_vertex", "vertex_count", "first_index", "index_count", "transformation_offset"],
     "formats": [np.uint32, np.uint32, np.uint32, np.uint32, np.uint64],
     "offsets": [0, 4, 8, 12, 16], "itemsize": 24})

SPHERE_DTYPE = np.dtype(
    {"names": ["center", "radius", "model"],
     "formats": [(np.float32, 3), np.float32, np.uint32],
     "offsets": [0, 12, 16], "itemsize": 32})

# Test hook of THIS binding, not of the library (tests/test_gpu_wide_nodes.py repeats the parity suite on the 8-ary formats in a
# child process): RT_AMD_NODE_FORMAT=0..4 makes every Context / Group created here start with that RT_OPT_NODE_FORMAT.
_TEST_NODE_FORMAT = int(os.environ["RT_AMD_NODE_FORMAT"]) if os.environ.get("RT_AMD_NODE_FORMAT", "") in ("0", "1", "2", "3", "4") else None

RT_OPT_COUNT_TRAVERSAL = 1
RT_OPT_AOV = 2
RT_OPT_KERNEL = 3
RT_OPT_FRONT_FACE_CW = 4
RT_OPT_REFIT = 5
RT_OPT_NODE_STRIDE = 6
RT_OPT_FRAME_PIPELINE = 7
RT_OPT_NODE_FORMAT = 8
RT_OPT_SAMPLER = 9

EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_set_option",
    "rt_scene_upload", "rt_spheres_upload", "rt_textures_upload", "rt_models_update",
    "rt_set_shard", "rt_accum_local_rows", "rt_accum_resize", "rt_accum_bind",
    "rt_dispatch", "rt_dispatch_bands", "rt_group_dispatch_bands", "rt_dispatch_frames", "rt_dispatch_frames_static", "rt_frames_reserve", "rt_sync", "rt_capture_rgb8",
    "rt_accum_download", "rt_aov_download", "rt_stats", "rt_stats_reset", "rt_last_dispatch_ms",
    "rt_debug_alloc_count", "rt_group_peer_access", "rt_group_accum_gather",
    "rt_group_create", "rt_group_destroy", "rt_group_size", "rt_group_context", "rt_group_set_option", "rt_group_scene_upload",
    "rt_group_spheres_upload", "rt_group_textures_upload", "rt_group_models_update", "rt_group_accum_resize", "rt_group_dispatch",
    "rt_group_dispatch_frames", "rt_group_sync", "rt_group_accum_download", "rt_group_capture_rgb8", "rt_group_stats", "rt_group_stats_reset",
    "rt_host_scene_load", "rt_host_scene_free", "rt_host_scene_vertex_count", "rt_host_scene_index_count",
    "rt_host_scene_model_count", "rt_host_scene_vertices", "rt_host_scene_indices", "rt_host_scene_primitives",
    "rt_host_scene_models", "rt_host_scene_model_name", "rt_host_scene_texture_count", "rt_host_scene_texture",
    "rt_host_scene_finish",
    "rt_host_uniform_default", "rt_host_capture_rgb8", "rt_host_write_png", "rt_host_bvh_info",
    "rt_host_app_init", "rt_host_app_invalidate", "rt_host_app_set_params", "rt_host_app_begin_frame", "rt_app_render",
]


class RtTexture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


APP_DTYPE = np.dtype(
    {"names": ["uniform", "inner_size", "last_window", "last_invalidation", "accumulate", "screen_fraction"],
     "formats": [UNIFORM_DTYPE, (np.uint32, 2), (np.uint32, 2), np.float64, np.uint8, np.uint8],
     "offsets": [0, 80, 88, 96, 104, 105], "itemsize": 112})


class App:
    """≙ struct App + App::render (src/app.rs:20-103), headless; time is passed in."""

    def __init__(self, inner_width, inner_height, now=0.0):
        self.state = np.zeros(1, APP_DTYPE)
        lib().rt_host_app_init(self.state.ctypes.data, inner_width, inner_height, C.c_double(now))

    @property
    def uniform(self):
        return self.state["uniform"]

    def invalidate(self, now):
        lib().rt_host_app_invalidate(self.state.ctypes.data, C.c_double(now))

    def set_params(self, edited_uniform, now):
        u = np.ascontiguousarray(edited_uniform)
        return bool(lib().rt_host_app_set_params(self.state.ctypes.data, u.ctypes.data, C.c_double(now)))

    def begin_frame(self, now):
        return bool(lib().rt_host_app_begin_frame(self.state.ctypes.data, C.c_double(now)))

    def render(self, ctx, now):
        _check(lib().rt_app_render(ctx.handle, self.state.ctypes.data, C.c_double(now)))
        ctx.width, ctx.height = int(self.uniform["window"][0][0]), int(self.uniform["window"][0][1])


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("pixel_frames", C.c_uint64), ("node_visits", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("wave_node_iters", C.c_uint64), ("wave_leaf_iters", C.c_uint64),
                ("wave_steps", C.c_uint64), ("dispatches", C.c_uint64), ("bvh_nodes", C.c_uint64),
                ("bvh_bytes", C.c_uint64), ("bvh_build_seconds", C.c_double), ("bvh_refits", C.c_uint64),
                ("bvh_rebuilds", C.c_uint64), ("launches", C.c_uint64),
                ("stack_entries", C.c_uint32), ("resident_blocks", C.c_uint32), ("guard_trips", C.c_uint64),
                ("node_format", C.c_uint32), ("pad0", C.c_uint32)]


class RtBvhInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint64), ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64),
                ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32), ("sah_cost", C.c_double),
                ("build_seconds", C.c_double)]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


_lib = None


def lib():
    """Load librt_amd.so.  Raises if it has not been built (no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C ray-tracing_amd)")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        L.rt_last_error.restype = C.c_char_p
        L.rt_create.argtypes = [i32, C.POINTER(vp)]
        L.rt_destroy.argtypes = [vp]
        L.rt_set_stream.argtypes = [vp, vp]
        L.rt_set_option.argtypes = [vp, i32, C.c_int64]
        L.rt_scene_upload.argtypes = [vp, vp, u64, vp, u64, vp, u32, vp, vp, u32]
        L.rt_spheres_upload.argtypes = [vp, vp, u64]
        L.rt_textures_upload.argtypes = [vp, vp, u32]
        L.rt_host_scene_texture_count.restype = u32
        L.rt_host_scene_texture_count.argtypes = [vp]
        L.rt_host_scene_texture.argtypes = [vp, u32, C.POINTER(RtTexture)]
        L.rt_models_update.argtypes = [vp, vp, u32, vp]
        L.rt_set_shard.argtypes = [vp, u32, u32]
        L.rt_accum_local_rows.restype = u64
        L.rt_accum_local_rows.argtypes = [vp, u32, u32]
        L.rt_accum_resize.argtypes = [vp, u32, u32]
        L.rt_accum_bind.argtypes = [vp, vp, u64, u32, u32]
        L.rt_dispatch.argtypes = [vp, vp]
        L.rt_dispatch_bands.argtypes = [vp, vp, u32, u32]
        L.rt_group_dispatch_bands.argtypes = [vp, vp, u32, u32]
        L.rt_dispatch_frames.argtypes = [vp, vp, u32]
        L.rt_dispatch_frames_static.argtypes = [vp, vp, u32]
        L.rt_frames_reserve.argtypes = [vp, u32]
        L.rt_capture_rgb8.argtypes = [vp, vp]
        L.rt_sync.argtypes = [vp]
        L.rt_accum_download.argtypes = [vp, vp]
        L.rt_aov_download.argtypes = [vp, vp]
        L.rt_stats.argtypes = [vp, C.POINTER(RtStats)]
        L.rt_stats_reset.argtypes = [vp]
        L.rt_last_dispatch_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.rt_group_create.argtypes = [C.POINTER(C.c_int), i32, C.POINTER(vp)]
        L.rt_group_destroy.argtypes = [vp]
        L.rt_group_size.argtypes = [vp]
        L.rt_group_context.argtypes = [vp, i32, C.POINTER(vp)]
        L.rt_group_peer_access.argtypes = [vp, C.POINTER(C.c_int), i32]
        L.rt_group_accum_gather.argtypes = [vp, C.POINTER(vp)]
        L.rt_debug_alloc_count.restype = u64
        L.rt_debug_alloc_count.argtypes = []
        L.rt_group_set_option.argtypes = [vp, i32, C.c_int64]
        L.rt_group_scene_upload.argtypes = [vp, vp, u64, vp, u64, vp, u32, vp, vp, u32]
        L.rt_group_spheres_upload.argtypes = [vp, vp, u64]
        L.rt_group_textures_upload.argtypes = [vp, vp, u32]
        L.rt_group_models_update.argtypes = [vp, vp, u32, vp]
        L.rt_group_accum_resize.argtypes = [vp, u32, u32]
        L.rt_group_dispatch.argtypes = [vp, vp]
        L.rt_group_dispatch_frames.argtypes = [vp, vp, u32]
        L.rt_group_sync.argtypes = [vp]
        L.rt_group_accum_download.argtypes = [vp, vp]
        L.rt_group_capture_rgb8.argtypes = [vp, vp]
        L.rt_group_stats.argtypes = [vp, C.POINTER(RtStats)]
        L.rt_group_stats_reset.argtypes = [vp]
        L.rt_host_scene_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.rt_host_scene_free.argtypes = [vp]
        L.rt_host_scene_free.restype = None
        for name in ("rt_host_scene_vertex_count", "rt_host_scene_index_count"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [vp]
        L.rt_host_scene_model_count.restype = u32
        L.rt_host_scene_model_count.argtypes = [vp]
        for name in ("rt_host_scene_vertices", "rt_host_scene_indices", "rt_host_scene_primitives", "rt_host_scene_models"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp]
        L.rt_host_scene_model_name.restype = C.c_char_p
        L.rt_host_scene_model_name.argtypes = [vp, u32]
        L.rt_host_scene_finish.argtypes = [vp, vp]
        L.rt_host_uniform_default.argtypes = [vp]
        L.rt_host_uniform_default.restype = None
        L.rt_host_capture_rgb8.argtypes = [vp, u64, vp]
        L.rt_host_capture_rgb8.restype = None
        L.rt_host_write_png.argtypes = [C.c_char_p, vp, u32, u32]
        L.rt_host_bvh_info.argtypes = [vp, u64, vp, u64, vp, u32, vp, C.POINTER(RtBvhInfo)]
        L.rt_host_app_init.argtypes = [vp, u32, u32, C.c_double]
        L.rt_host_app_init.restype = None
        L.rt_host_app_invalidate.argtypes = [vp, C.c_double]
        L.rt_host_app_invalidate.restype = None
        L.rt_host_app_set_params.argtypes = [vp, vp, C.c_double]
        L.rt_host_app_begin_frame.argtypes = [vp, C.c_double]
        L.rt_app_render.argtypes = [vp, vp, C.c_double]
        _lib = L
    return _lib


def _check(code):
    if code != 0:
        raise RtError(code, lib().rt_last_error().decode("utf-8", "replace"))


# ------------------------------------------------------------- host half
class HostScene:
    """≙ Scene (src/scene.rs:20-27) loaded by the C++ host loader."""

    def __init__(self, path: str):
        h = C.c_void_p()
        _check(lib().rt_host_scene_load(path.encode(), C.byref(h)))
        self.handle = h
        L = lib()
        nv, ni, nm = L.rt_host_scene_vertex_count(h), L.rt_host_scene_index_count(h), L.rt_host_scene_model_count(h)

        def view(ptr, dtype, n):
            if n == 0:
                return np.zeros(0, dtype)
            buf = (C.c_char * (dtype.itemsize * n)).from_address(ptr)
            # copy as raw bytes: a structured copy would leave the padding bytes undefined
            return np.frombuffer(buf, dtype=np.uint8).copy().view(dtype)

        self.vertices = view(L.rt_host_scene_vertices(h), VERTEX_DTYPE, nv)
        self.indices = view(L.rt_host_scene_indices(h), np.dtype(np.uint32), ni)
        self.primitives = view(L.rt_host_scene_primitives(h), PRIMITIVE_DTYPE, nm)
        self.models = view(L.rt_host_scene_models(h), MODEL_DTYPE, nm)
        self.names = [L.rt_host_scene_model_name(h, i).decode() for i in range(nm)]
        self.textures = []          # [h, w, 4] uint8, file row order
        for i in range(L.rt_host_scene_texture_count(h)):
            t = RtTexture()
            _check(L.rt_host_scene_texture(h, i, C.byref(t)))
            buf = (C.c_char * (4 * t.width * t.height)).from_address(t.rgba8)
            self.textures.append(np.frombuffer(buf, np.uint8).copy().reshape(t.height, t.width, 4))

    def close(self):
        if getattr(self, "handle", None):
            lib().rt_host_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def default_uniform():
    """≙ the Uniform literal at src/main.rs:73-84."""
    u = np.zeros(1, UNIFORM_DTYPE)
    lib().rt_host_uniform_default(u.ctypes.data)
    return u


def make_uniform(width, height, position, yaw, pitch=0.0, fov=np.pi / 2, aspect=None, frame=1,
                 accumulation_frame=0, flags=0, exposure=1.0, environment=1.0, max_bounces=10, samples=5):
    u = default_uniform()
    u["window"][0] = (width, height)
    u["position"][0] = position
    u["pitch"] = np.float32(pitch)
    u["yaw"] = np.float32(yaw)
    u["fov"] = np.float32(fov)
    u["aspect"] = np.float32(width) / np.float32(height) if aspect is None else np.float32(aspect)  # src/camera.rs:56
    u["frame"] = frame
    u["accumulation_frame"] = accumulation_frame
    u["flags"] = flags
    u["exposure"] = exposure
    u["environment"] = environment
    u["max_bounces"] = max_bounces
    u["samples"] = samples
    return u


def capture_rgb8(accum):
    a = np.ascontiguousarray(accum, np.float32).reshape(-1, 4)
    out = np.zeros((a.shape[0], 3), np.uint8)
    lib().rt_host_capture_rgb8(a.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def write_png(path, rgb, w, h):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _check(lib().rt_host_write_png(path.encode(), rgb.ctypes.data, w, h))


def bvh_info(vertices, indices, primitives, transforms=None):
    info = RtBvhInfo()
    v = np.ascontiguousarray(vertices)
    i = np.ascontiguousarray(indices, np.uint32)
    p = np.ascontiguousarray(primitives)
    t = None if transforms is None else np.ascontiguousarray(transforms, np.float32)
    _check(lib().rt_host_bvh_info(v.ctypes.data, len(v), i.ctypes.data, len(i), p.ctypes.data, len(p),
                                  None if t is None else t.ctypes.data, C.byref(info)))
    return info


# ----------------------------------------------------------- device half
class Context:
    """One rt_context: the replacement for the reference's ComputePipeline + bound resources."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib().rt_create(device, C.byref(h)))
        self.handle = h
        self.width = self.height = 0
        if _TEST_NODE_FORMAT is not None:
            self.set_option(RT_OPT_NODE_FORMAT, _TEST_NODE_FORMAT)

    def close(self):
        if getattr(self, "handle", None):
            lib().rt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, key, value):
        _check(lib().rt_set_option(self.handle, key, int(value)))

    def set_stream(self, stream_ptr):
        _check(lib().rt_set_stream(self.handle, stream_ptr))

    def scene_upload(self, vertices, indices, primitives, models, transforms=None):
        v = np.ascontiguousarray(vertices)
        i = np.ascontiguousarray(indices, np.uint32)
        p = np.ascontiguousarray(primitives)
        m = np.ascontiguousarray(models)
        t = None if transforms is None else np.ascontiguousarray(transforms, np.float32)
        assert v.dtype.itemsize == 48 and p.dtype.itemsize == 24 and m.dtype.itemsize == 128
        _check(lib().rt_scene_upload(self.handle, v.ctypes.data, len(v), i.ctypes.data, len(i), p.ctypes.data, len(p),
                                     None if t is None else t.ctypes.data, m.ctypes.data, len(m)))

    def spheres_upload(self, spheres):
        sp = np.ascontiguousarray(spheres)
        assert sp.dtype.itemsize == 32
        _check(lib().rt_spheres_upload(self.handle, sp.ctypes.data, len(sp)))

    def textures_upload(self, images):
        """images: list of [h, w, 4] uint8 RGBA arrays in file row order (top row first)."""
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        arr = (RtTexture * max(1, len(imgs)))()
        for i, im in enumerate(imgs):
            assert im.ndim == 3 and im.shape[2] == 4
            arr[i].rgba8, arr[i].width, arr[i].height = im.ctypes.data, im.shape[1], im.shape[0]
        _check(lib().rt_textures_upload(self.handle, arr, len(imgs)))

    def scene_finish(self, host_scene: HostScene):
        _check(lib().rt_host_scene_finish(host_scene.handle, self.handle))

    def models_update(self, models, transforms=None):
        m = np.ascontiguousarray(models)
        t = None if transforms is None else np.ascontiguousarray(transforms, np.float32)
        _check(lib().rt_models_update(self.handle, m.ctypes.data, len(m), None if t is None else t.ctypes.data))

    def set_shard(self, index, count):
        _check(lib().rt_set_shard(self.handle, index, count))

    def local_rows(self, w, h):
        return int(lib().rt_accum_local_rows(self.handle, w, h))

    def accum_resize(self, w, h):
        _check(lib().rt_accum_resize(self.handle, w, h))
        self.width, self.height = w, h

    def accum_bind(self, device_ptr, nbytes, w, h):
        _check(lib().rt_accum_bind(self.handle, device_ptr, nbytes, w, h))
        self.width, self.height = w, h

    def dispatch(self, uniform):
        u = np.ascontiguousarray(uniform)
        _check(lib().rt_dispatch(self.handle, u.ctypes.data))

    def dispatch_bands(self, uniform, band_begin, band_end=0xFFFFFFFF):
        """One piece of a frame: the context's 8-row bands [band_begin, band_end) (rt_abi.h rt_dispatch_bands)."""
        u = np.ascontiguousarray(uniform)
        _check(lib().rt_dispatch_bands(self.handle, u.ctypes.data, band_begin, band_end))

    def dispatch_frames(self, uniform, n_frames):
        """Advances uniform['frame'] / ['accumulation_frame'] in place (src/app.rs:78-96)."""
        assert uniform.flags.c_contiguous and uniform.dtype.itemsize == 80
        _check(lib().rt_dispatch_frames(self.handle, uniform.ctypes.data, n_frames))

    def dispatch_frames_static(self, uniform, n_frames):
        """rt_dispatch_frames without any device allocation (static scratch, size-limited)."""
        assert uniform.flags.c_contiguous and uniform.dtype.itemsize == 80
        _check(lib().rt_dispatch_frames_static(self.handle, uniform.ctypes.data, n_frames))

    def capture_rgb8(self):
        """[local_rows, width, 3] uint8: the accumulation buffer through Capture's conversion, done on the device."""
        rows = self.local_rows(self.width, self.height)
        out = np.zeros((rows, self.width, 3), np.uint8)
        _check(lib().rt_capture_rgb8(self.handle, out.ctypes.data))
        return out

    def frames_reserve(self, n_frames):
        _check(lib().rt_frames_reserve(self.handle, n_frames))

    def sync(self):
        _check(lib().rt_sync(self.handle))

    def accum_download(self):
        rows = self.local_rows(self.width, self.height)
        out = np.zeros((rows, self.width, 4), np.float32)
        _check(lib().rt_accum_download(self.handle, out.ctypes.data))
        return out

    def aov_download(self):
        rows = self.local_rows(self.width, self.height)
        out = np.zeros((rows, self.width, 2), np.uint32)
        _check(lib().rt_aov_download(self.handle, out.ctypes.data))
        return out

    def stats(self):
        s = RtStats()
        _check(lib().rt_stats(self.handle, C.byref(s)))
        return s

    def stats_reset(self):
        _check(lib().rt_stats_reset(self.handle))

    def last_dispatch_ms(self):
        ms = C.c_float()
        _check(lib().rt_last_dispatch_ms(self.handle, C.byref(ms)))
        return float(ms.value)


class Group:
    """One rt_group: n GPUs behind one host object (the scene is built once, every device renders its bands)."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        _check(lib().rt_group_create(ids, len(device_ids), C.byref(h)))
        self.handle = h
        self.width = self.height = 0
        if _TEST_NODE_FORMAT is not None:
            self.set_option(RT_OPT_NODE_FORMAT, _TEST_NODE_FORMAT)

    def close(self):
        if getattr(self, "handle", None):
            lib().rt_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __len__(self):
        return int(lib().rt_group_size(self.handle))

    def peer_access(self):
        """Per shard: RT_PEER_* (0 same device as shard 0, 1 peer access enabled, 2 unavailable, 3 enabling failed)."""
        n = len(self)
        st = (C.c_int * n)()
        _check(lib().rt_group_peer_access(self.handle, st, n))
        return list(st)

    def set_option(self, key, value):
        _check(lib().rt_group_set_option(self.handle, key, int(value)))

    def scene_upload(self, vertices, indices, primitives, models, transforms=None):
        v = np.ascontiguousarray(vertices)
        i = np.ascontiguousarray(indices, np.uint32)
        p = np.ascontiguousarray(primitives)
        m = np.ascontiguousarray(models)
        t = None if transforms is None else np.ascontiguousarray(transforms, np.float32)
        _check(lib().rt_group_scene_upload(self.handle, v.ctypes.data, len(v), i.ctypes.data, len(i), p.ctypes.data, len(p),
                                           None if t is None else t.ctypes.data, m.ctypes.data, len(m)))

    def spheres_upload(self, spheres):
        sp = np.ascontiguousarray(spheres)
        _check(lib().rt_group_spheres_upload(self.handle, sp.ctypes.data, len(sp)))

    def textures_upload(self, images):
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        arr = (RtTexture * max(1, len(imgs)))()
        for i, im in enumerate(imgs):
            assert im.ndim == 3 and im.shape[2] == 4
            arr[i].rgba8, arr[i].width, arr[i].height = im.ctypes.data, im.shape[1], im.shape[0]
        _check(lib().rt_group_textures_upload(self.handle, arr, len(imgs)))

    def models_update(self, models, transforms=None):
        m = np.ascontiguousarray(models)
        t = None if transforms is None else np.ascontiguousarray(transforms, np.float32)
        _check(lib().rt_group_models_update(self.handle, m.ctypes.data, len(m), None if t is None else t.ctypes.data))

    def accum_resize(self, w, h):
        _check(lib().rt_group_accum_resize(self.handle, w, h))
        self.width, self.height = w, h

    def dispatch(self, uniform):
        u = np.ascontiguousarray(uniform)
        _check(lib().rt_group_dispatch(self.handle, u.ctypes.data))

    def dispatch_bands(self, uniform, band_begin, band_end=0xFFFFFFFF):
        u = np.ascontiguousarray(uniform)
        _check(lib().rt_group_dispatch_bands(self.handle, u.ctypes.data, band_begin, band_end))

    def dispatch_frames(self, uniform, n_frames):
        assert uniform.flags.c_contiguous and uniform.dtype.itemsize == 80
        _check(lib().rt_group_dispatch_frames(self.handle, uniform.ctypes.data, n_frames))

    def sync(self):
        _check(lib().rt_group_sync(self.handle))

    def accum_download(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        _check(lib().rt_group_accum_download(self.handle, out.ctypes.data))
        return out

    def accum_gather(self):
        """Gather on the first device without downloading; returns the device address of the whole image there."""
        p = C.c_void_p()
        _check(lib().rt_group_accum_gather(self.handle, C.byref(p)))
        return p.value

    def capture_rgb8(self):
        out = np.zeros((self.height, self.width, 3), np.uint8)
        _check(lib().rt_group_capture_rgb8(self.handle, out.ctypes.data))
        return out

    def stats(self):
        s = RtStats()
        _check(lib().rt_group_stats(self.handle, C.byref(s)))
        return s

    def stats_reset(self):
        _check(lib().rt_group_stats_reset(self.handle))

    def shard_ms(self, shard):
        """Device time of shard's last dispatch (rt_last_dispatch_ms on its context)."""
        h = C.c_void_p()
        _check(lib().rt_group_context(self.handle, shard, C.byref(h)))
        ms = C.c_float()
        _check(lib().rt_last_dispatch_ms(h, C.byref(ms)))
        return float(ms.value)
