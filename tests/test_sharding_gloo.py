"""Multi-rank path on CPU: world_size-2 (and 3) gloo runs of the band sharding + single gather.
Each rank 'renders' its bands with the oracle (global pixel_idx, so results must equal the
unsharded image bit for bit), packs them as rt_set_shard lays them out, and
rank 0 gathers and de-interleaves."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    graft.load_package()
    from ray_tracing_amd import sharding
    from oracle import obj_loader, oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = graft.load_config("C1")
    sc = oracle.OracleScene(obj_loader.load(os.path.join(ROOT, cfg["scene"])))
    u = oracle.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=2, samples=1)
    full = np.zeros((h, w, 4), np.float32)
    rows = sharding.band_rows_of(rank, world, h)
    for lb in range(sharding.local_bands(h, rank, world)):
        y0 = (lb * world + rank) * 8
        sc.render(u, full, mode=1, rows=(y0, min(y0 + 8, h)))
    local = torch.zeros(sharding.local_buffer_floats(w, h, world), dtype=torch.float32)
    packed = full[rows]                                   # local row order
    local[: packed.size] = torch.from_numpy(packed.reshape(-1))
    img = sharding.gather_framebuffer(local, w, h, rank, world)
    if rank == 0:
        np.save(out_path, img.numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,w,h", [(2, 40, 52), (3, 24, 64), (2, 16, 8)])
def test_sharded_render_equals_unsharded(tmp_path, oracle_mod, oracle_scenes, world, w, h):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    got = np.load(out)
    import __graft_entry__ as graft
    cfg = graft.load_config("C1")
    u = oracle_mod.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=2, samples=1)
    want, _, _ = oracle_scenes("cornell-box").render(u, mode=1)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_band_bookkeeping():
    import __graft_entry__ as graft
    graft.load_package()
    from ray_tracing_amd import sharding
    for h in [1, 7, 8, 9, 64, 1080, 2160]:
        for world in [1, 2, 3, 4, 8]:
            rows = [sharding.band_rows_of(r, world, h) for r in range(world)]
            flat = sorted(y for r in rows for y in r)
            assert flat == list(range(h))
            assert sum(sharding.local_bands(h, r, world) for r in range(world)) == sharding.n_bands(h)
            assert all(sharding.local_bands(h, r, world) <= sharding.max_local_bands(h, world) for r in range(world))
