import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The ctypes binding of librt_amd.so (built beforehand by __graft_entry__.build())."""
    if not os.path.exists(os.path.join(ROOT, "ray-tracing_amd", "librt_amd.so")) or \
            not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        graft.build()
    mod = graft.load_package()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def obj_loader():
    from oracle import obj_loader
    return obj_loader


def scene_path(name):
    return os.path.join(ROOT, "scenes", name + ".obj")


@pytest.fixture(scope="session")
def scenes(obj_loader):
    """Python-loader scenes, cached."""
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = obj_loader.load(scene_path(name))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def oracle_scenes(oracle_mod, scenes):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = oracle_mod.OracleScene(scenes(name))
        return cache[name]
    return get


def config(name):
    return graft.load_config(name)


def uniforms_for(rt_or_oracle, cfg, width, height, **kw):
    args = dict(max_bounces=cfg["max_bounces"], samples=cfg["samples"])
    args.update(kw)
    return rt_or_oracle.make_uniform(width, height, cfg["position"], cfg["yaw"], cfg["pitch"], **args)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_images_identical(got, want, what=""):
    """Bit-exact, with NaN == NaN (x86 and gfx950 differ in NaN sign/payload)."""
    g, w = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert g.shape == w.shape, (g.shape, w.shape)
    same = (bits(g) == bits(w)) | (np.isnan(g) & np.isnan(w))
    if not same.all():
        bad = np.argwhere(~same)
        raise AssertionError(f"{what}: {len(bad)} of {same.size} values differ; first at {bad[0]}: "
                             f"{g[tuple(bad[0])]!r} vs {w[tuple(bad[0])]!r}; max abs diff "
                             f"{np.nanmax(np.abs(g - w))}")
