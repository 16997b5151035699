"""
sharding.py — multi-GPU image sharding for the render path.

The reference is single-GPU.  Pixels are independent (each invocation writes
only accumulation[pixel_idx], shaders/main.wgsl:31-32, and seeds its RNG from
the global pixel index, main.wgsl:23), so the image is cut into bands of 8 rows
(the reference's workgroup rows, main.wgsl:15) dealt round-robin to ranks.  The
scene is replicated; there is no exchange while rendering; one gather of the
accumulated bands to rank 0 ends the job (RCCL over xGMI on GPUs, gloo in the
CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

BAND_ROWS = 8


def n_bands(height: int) -> int:
    return (height + BAND_ROWS - 1) // BAND_ROWS


def local_bands(height: int, rank: int, world: int) -> int:
    b = n_bands(height)
    return 0 if b <= rank else (b - rank + world - 1) // world


def max_local_bands(height: int, world: int) -> int:
    return local_bands(height, 0, world)


def local_buffer_floats(width: int, height: int, world: int) -> int:
    """Every rank allocates the same (largest) size so one gather moves equal pieces."""
    return max_local_bands(height, world) * BAND_ROWS * width * 4


def band_rows_of(rank: int, world: int, height: int):
    """Global row indices, in local order, of the rows `rank` renders."""
    rows = []
    for lb in range(local_bands(height, rank, world)):
        y0 = (lb * world + rank) * BAND_ROWS
        rows.extend(range(y0, min(y0 + BAND_ROWS, height)))
    return rows


def deinterleave(stacked: torch.Tensor, width: int, height: int, world: int) -> torch.Tensor:
    """stacked: [world, max_local_bands*8*width*4] → [height, width, 4] in image order."""
    mb = max_local_bands(height, world)
    t = stacked.view(world, mb, BAND_ROWS, width, 4).permute(1, 0, 2, 3, 4).reshape(mb * world * BAND_ROWS, width, 4)
    return t[:height].contiguous()


def gather_framebuffer(local: torch.Tensor, width: int, height: int, rank: int, world: int, dst: int = 0, force_collective: bool = False):
    """One collective: every rank's packed bands → rank `dst`.  Returns the
    [height, width, 4] image on `dst`, None elsewhere.  force_collective: issue the gather even for one rank (a one-GPU box
    then runs the RCCL path end to end: bench.py RT_BENCH_FORCE_DIST)."""
    assert local.numel() == local_buffer_floats(width, height, world)
    if world == 1 and not force_collective:
        return deinterleave(local.view(1, -1), width, height, 1)
    if rank == dst:
        out = torch.empty((world, local.numel()), dtype=local.dtype, device=local.device)
        dist.gather(local.view(-1), list(out.unbind(0)), dst=dst)
        return deinterleave(out, width, height, world)
    dist.gather(local.view(-1), None, dst=dst)
    return None
