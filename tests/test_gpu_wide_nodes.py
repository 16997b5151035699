"""
The 8-ary node format (RT_OPT_NODE_FORMAT = 3, DevNode8) is chosen automatically only for scenes with millions of
primitives (tests/test_gpu_configs_at_size.py).  Here the WHOLE GPU parity suite — every scene, kernel variant, material,
sphere / texture / refit / sharding / multi-frame case of test_gpu_parity.py, test_textures.py and test_gpu_group.py —
is run once more in a child process whose contexts default to that format (RT_AMD_NODE_FORMAT=3), so the wide format is
held to the same bit-exact bar as the binary ones on all of them.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("RT_AMD_NODE_FORMAT") is not None, reason="already the child run")
@pytest.mark.parametrize("fmt", ["3", "4"])       # 80-byte and 64-byte 8-ary records
def test_parity_suite_on_8ary_nodes(fmt):
    env = dict(os.environ, RT_AMD_NODE_FORMAT=fmt)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_parity.py", "tests/test_textures.py", "tests/test_gpu_group.py", "tests/test_zz_no_alloc.py"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


def test_8ary_format_is_really_used(rt, scenes):
    """Guards the test above against passing on a silent fallback: with format 3 the tree has 80-byte nodes and far fewer of them."""
    from conftest import config
    s = scenes("teapot-circle")
    sizes = {}
    for fmt in (1, 3, 4):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.RT_OPT_NODE_FORMAT, fmt)
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
            st = ctx.stats()
            n_prims = len(s.indices) // 3
            sizes[fmt] = (st.bvh_nodes, (st.bvh_bytes - n_prims * (48 + 128)) / st.bvh_nodes, st.stack_entries)
    assert sizes[1][1] == 48 and sizes[3][1] == 80 and sizes[4][1] == 64
    assert sizes[3][0] < sizes[1][0] / 1.8 and sizes[4][0] == sizes[3][0]
    assert sizes[3][2] <= 24 and sizes[4][2] == sizes[3][2]


@pytest.mark.parametrize("fmt", [3, 4])
def test_collapse_cut_optimal_vs_greedy(rt, scenes, fmt):
    """The 8-ary tree is cut out of the BVH2 by dynamic programming (fewest expected node visits); developer knob 112 = 1 gives the
    older greedy cut.  Both are trees over the same primitives, so the images agree bit for bit, and the optimal cut needs fewer
    nodes and fewer node visits per ray."""
    import numpy as np
    from conftest import config
    s = scenes("teapot-circle")
    cfg = config("C3")
    W, H = 160, 90
    u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=4, samples=2)
    got = {}
    for greedy in (0, 1):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.RT_OPT_NODE_FORMAT, fmt)
            ctx.set_option(112, greedy)
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
            ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 1)
            ctx.accum_resize(W, H)
            ctx.dispatch(u)
            st = ctx.stats()
            got[greedy] = (np.array(ctx.accum_download(), copy=True), st.bvh_nodes, st.node_visits, st.rays)
    assert got[0][0].tobytes() == got[1][0].tobytes()
    assert got[0][3] == got[1][3]
    assert got[0][1] < got[1][1], (got[0][1], got[1][1])
    assert got[0][2] < got[1][2], (got[0][2], got[1][2])
    with rt.Context(0) as ctx:
        with pytest.raises(rt.RtError):
            ctx.set_option(112, 2)
