"""Ray sharding across ranks (one process per GPU) + the single histogram all-reduce.

Rays are independent and ray i always draws from Philox stream (seed, i), so ANY partition of
the index range gives the same summed histogram.  Rank r of P traces the
contiguous range shard(n, r, P); the only exchange is one SUM all-reduce of the
[n_theta*n_phi] int64 histogram (+ the 7-word census) — torch.distributed backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

The tracer itself is injected (`trace(cfg, count, seed, first) -> (hits, stats)`): on a GPU
box that is altair_raytracing_amd.fluxmap; there is no CPU tracer in this package.
"""
from typing import Callable, Tuple

import numpy as np


def shard(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of [0,n_total): returns (first, count); the first n_total%world ranks get one extra ray."""
    if world < 1 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_total, world)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def step_slice(step: int, rank: int, world: int, rays_per_rank: int) -> Tuple[int, int]:
    """bench.py's weak-scaling schedule: in step s, rank r traces rays [(s*world + r)*n, +n).  Over `steps` steps and all
    ranks these slices tile [0, steps*world*n) exactly once, and the union over ranks of one step is the contiguous block
    [s*world*n, (s+1)*world*n) -- so the summed histogram of a step does not depend on the number of ranks."""
    if world < 1 or not (0 <= rank < world) or step < 0 or rays_per_rank < 0:
        raise ValueError("bad step slice request")
    return (step * world + rank) * rays_per_rank, rays_per_rank


CENSUS_FIELDS = ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")


def fluxmap_sharded(trace: Callable, cfg, n_total: int, seed: int, first_ray: int = 0, device=None):
    """Trace this rank's shard with `trace`, then all-reduce histogram and census over the default
    process group (if initialised).  Returns (hits[n_theta,n_phi] uint64, census dict) — identical on every rank."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    first, count = shard(n_total, rank, world)
    hits, st = trace(cfg, count, seed, first_ray + first)
    census = np.array([getattr(st, k) for k in CENSUS_FIELDS], dtype=np.int64)
    if world > 1:
        buf = torch.from_numpy(np.concatenate([hits.reshape(-1).astype(np.int64), census]))
        if device is not None:
            buf = buf.to(device)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf = buf.cpu().numpy()
        hits = buf[:-len(CENSUS_FIELDS)].astype(np.uint64).reshape(hits.shape)
        census = buf[-len(CENSUS_FIELDS):]
    return hits, dict(zip(CENSUS_FIELDS, (int(x) for x in census)))


def disc_sweep_sharded(sweep: Callable, cfg, discs, radius: float, half_thick: float, n_total: int, seed: int,
                       first_ray: int = 0, device=None):
    """BASELINE.json configs[3] (integratingSphereDetectorSweep.C ray-sharded): every rank traces its slice of the
    n_total rays against ALL discs with `sweep(cfg, discs, radius, half_thick, count, seed, first) -> (hits, stats)`
    (altair_raytracing_amd.disc_sweep on a GPU box), then the per-disc counts and the census are summed by the same
    single all-reduce as the flux map."""
    return fluxmap_sharded(lambda c, count, s, first: sweep(c, discs, radius, half_thick, count, s, first),
                           cfg, n_total, seed, first_ray, device)
