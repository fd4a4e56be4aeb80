"""Multi-GPU sharding of the ocean path: independent units, no data-path collective.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
Two shardable axes exist in FFTMesh semantics, because a step is a pure function of (h0, h0conj, t)
(S/FFTMesh.cs:178-190):
  * tiles  -- distinct (seed, wind) oceans, one per rank (BASELINE configs[2]);
  * steps  -- time-steps of ONE ocean, contiguous blocks per rank after h0 has been broadcast once.
OceanRenderer semantics iterates its phase (F/FFTCommon.cginc:101-104), so only the tile axis shards there.
The only collective is the optional gather of finished outputs, issued once per batch (29.4 MB per 1024^2 tile
~ 190 us on one 153 GB/s xGMI link, i.e. ~8 steps of synthesis) -- never per step.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def tile_seed(base_seed: int, rank: int) -> int:
    """Seed of the tile owned by `rank` (bench.py: seed = 1 + rank)."""
    return int(base_seed) + int(rank)


def shard_steps(nsteps: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of time-step indices for `rank`; blocks differ by at most one step."""
    if world < 1 or not (0 <= rank < world) or nsteps < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(nsteps, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def step_times(lo: int, hi: int, dt: float = 1.0 / 60.0) -> List[float]:
    """t_k = (k+1)*dt, the bench's time axis."""
    return [(k + 1) * dt for k in range(lo, hi)]


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """The bench's timing rule: the job took as long as its slowest rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value: float, dist=None, device=None) -> List[float]:
    """`value` of every rank, in rank order, on every rank (one all-reduce of a one-hot vector: works on gloo and on RCCL alike)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=device)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]
