"""CPU tier: the N>1 path (tile / time-step sharding, max-over-ranks timing, gather) with world_size-2 gloo
processes.  No GPU: each rank's "synthesis" is done by the oracle at a tiny N, and the gathered result must equal a
single-process evaluation of all tiles / all steps -- i.e. the sharding logic itself is what is under test."""
import os
import socket
import sys

import numpy as np
import pytest

import workloads


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---- test support: a torch.distributed gather of what the ranks computed (the product gathers over RCCL inside the library,
# mw_tiles_gather; these two stand in for it on the CPU tier so that the SHARDING ARITHMETIC of mistral_water.parallel -- which seed,
# which step block, which order -- can be checked end to end without a GPU) ------------------------------------------------------
def gather_tiles(local, dist, dst: int = 0):
    """Collect every rank's tile outputs (a flat tensor) on `dst`; returns the list there, None elsewhere."""
    world = dist.get_world_size()
    bufs = [local.new_empty(local.shape) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(local, bufs, dst=dst)
    return bufs


def gather_step_blocks(local_rows, nsteps: int, dist, shard_steps):
    """All-gather variable-length per-rank step blocks back into time order ([nsteps, ...] on every rank)."""
    import torch
    world = dist.get_world_size()
    sizes = [shard_steps(nsteps, world, r) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad = local_rows.new_zeros((longest,) + tuple(local_rows.shape[1:]))
    pad[: local_rows.shape[0]] = local_rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([out[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (repo, os.path.join(repo, "mistral-water_amd"), os.path.join(repo, "tests")):
        sys.path.insert(0, p)
    from mistral_water import parallel
    from oracle import oracle as O
    import workloads as W
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = W.fftmesh_params(16)
        # --- tile axis: one ocean per rank (seed = base + rank), gathered on rank 0 ---------------------------
        h0, h0c = O.generate_spectrum(p, parallel.tile_seed(1, rank))
        v, n, c = O.eval_f64(p, h0, h0c, 0.5)
        local = torch.from_numpy(np.concatenate([v.ravel(), n.ravel(), c[:, 0]]))
        tiles = gather_tiles(local, dist, dst=0)
        # --- time axis: one ocean, contiguous blocks of 5 steps over 2 ranks ----------------------------------
        g0, g0c = O.generate_spectrum(p, 7)
        lo, hi = parallel.shard_steps(5, world, rank)
        rows = np.stack([O.eval_f64(p, g0, g0c, t)[0][:, 1] for t in parallel.step_times(lo, hi)]) if hi > lo \
            else np.zeros((0, 256))
        allrows = gather_step_blocks(torch.from_numpy(rows), 5, dist, parallel.shard_steps)
        slow = parallel.max_over_ranks(1.0 + rank, dist)
        # --- OceanRenderer semantics, time axis: only the phase links the frames, so rank r SEEKS to frame lo with the Dispersion pass alone
        #     (what mw_ocean_advance_phase does on the device) and renders its block (bench.py --workload renderer1024 --shard steps) ----------
        rp = O.RendererParams(resolution=8, length=27.155, wind_x=14.45, wind_y=12.0, amplitude=0.41, choppiness=0.46, gravity=9.81, mult=1.5)
        init4 = O.renderer_initial_spectrum(rp, 3)
        dts = [0.016, 0.3, 0.0, 600.0, 0.02]
        ph = np.zeros((rp.M, rp.M), np.float32)
        for k in range(lo):
            O.renderer_advance_phase(rp, init4, ph, dts[k])
        frames = np.stack([O.renderer_step_f64(rp, init4, ph, dts[k], literal_passes=False)[0].ravel() for k in range(lo, hi)]) if hi > lo \
            else np.zeros((0, rp.M * rp.M))
        allframes = gather_step_blocks(torch.from_numpy(frames), 5, dist, parallel.shard_steps)
        for k in range(hi, 5):
            O.renderer_advance_phase(rp, init4, ph, dts[k])      # ... and walks on to frame K: every rank ends in the same state
        q.put((rank, None if tiles is None else [t.numpy() for t in tiles], allrows.numpy(), slow, (lo, hi), allframes.numpy(), ph))
    finally:
        dist.destroy_process_group()


def test_shard_steps_partition():
    from mistral_water import parallel
    for n in (0, 1, 5, 1000, 1001):
        for w in (1, 2, 3, 8):
            blocks = [parallel.shard_steps(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_steps(4, 2, 2)


def test_two_rank_gloo_tiles_and_steps(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    p = workloads.fftmesh_params(16)
    # tiles: rank 0 holds both tiles, each equal to a single-process evaluation with that tile's seed
    tiles = res[0][1]
    assert res[1][1] is None and len(tiles) == 2
    for r in range(2):
        h0, h0c = oracle.generate_spectrum(p, 1 + r)
        v, n, c = oracle.eval_f64(p, h0, h0c, 0.5)
        assert np.array_equal(tiles[r], np.concatenate([v.ravel(), n.ravel(), c[:, 0]]))
    assert not np.array_equal(tiles[0], tiles[1])
    # steps: both ranks end with all 5 steps in time order
    g0, g0c = oracle.generate_spectrum(p, 7)
    want = np.stack([oracle.eval_f64(p, g0, g0c, (k + 1) / 60.0)[0][:, 1] for k in range(5)])
    for r in range(2):
        assert np.array_equal(res[r][2], want)
        assert res[r][3] == 2.0  # max over ranks of (1 + rank)
    assert res[0][4] == (0, 3) and res[1][4] == (3, 5)
    # OceanRenderer frames sharded over the ranks: the same five height textures as ONE process rendering them in sequence, and the same
    # phase texture at the end on every rank
    rp = oracle.RendererParams(resolution=8, length=27.155, wind_x=14.45, wind_y=12.0, amplitude=0.41, choppiness=0.46, gravity=9.81, mult=1.5)
    init4 = oracle.renderer_initial_spectrum(rp, 3)
    ph = np.zeros((rp.M, rp.M), np.float32)
    seq = np.stack([oracle.renderer_step_f64(rp, init4, ph, dt, literal_passes=False)[0].ravel() for dt in (0.016, 0.3, 0.0, 600.0, 0.02)])
    for r in range(2):
        assert np.array_equal(res[r][5], seq) and np.array_equal(res[r][6], ph)
