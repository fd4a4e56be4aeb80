#!/usr/bin/env python
"""bench.py -- ocean grid-points/sec (spectrum -> transform -> displacement -> Jacobian) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one time-step of one 1024 x 1024 FFTMesh-semantics ocean tile
(BASELINE.json configs[1]); with N GPUs every rank owns an independent tile (seed = 1 + rank, configs[2]),
no data-path collective, weak scaling.  Inputs (h0/h0conj -> packed spectrum tables) are resident in HBM
before the timed region; outputs (vertices, normals, whitecap) stay in device memory.

Extra objects on the same line:
  roofline      dominant kernel (k_pass2) algorithmic bytes / its mean launch duration measured live with
                hipEvents on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  the oracle's literal restatement of the reference CPU path (S/FFTMesh.cs:192-249, O(N^4)),
                single thread, on a bounded vertex sample of the same 1024^2 step.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "mistral-water_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
# BASELINE.json's metric string, verbatim
BASELINE_METRIC = "ocean grid-points/sec (full spectrum\u2192IFFT\u2192disp\u2192Jacobian), 1024\u00b2 grid, 1/2/4/8 GPUs"
BYTES_PER_POINT = 92       # SURVEY.md 8d canonical algorithmic traffic of one FFTMesh step
BYTES_PASS1 = 16 + 24      # read (P,Q) + write 3 packed complex fields
BYTES_PASS2 = 24 + 28      # read 3 packed complex fields + write vertex 12 + normal 12 + whitecap 4
BYTES_POND = 24            # read position 12 + write position 12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="ocean1024", choices=["ocean1024", "ocean4096", "ocean2048", "ocean512", "ocean256", "pond", "renderer1024"])
    ap.add_argument("--batch", type=int, default=32, help="time-steps per enqueue (FFTMesh steps are independent in t)")
    ap.add_argument("--preheat-ms", type=float, default=150.0,
                    help="untimed: keep the device busy with the workload this long before the W warm-up steps, so the "
                         "timed region starts at steady clocks (the first ~5 ms after idle run ~25 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time the RCCL gather of the last step's tiles (configs[2])")
    return ap.parse_args()


def pmc_traffic(N, B):
    """HBM-side bytes per k_pass2 launch from the committed rocprofv3 PMC passes (tools/profile_r1.sh ->
    profiles/r01_ocean{N}_b{B}_pmc.json): (2 * FETCH_SIZE + WRITE_SIZE) KiB, i.e. with the gfx950 correction the
    micro-architecture guide prescribes (FETCH_SIZE counts 128-B requests at 64 B).  Counters cannot be read from
    inside this process, so this is the value measured by the same command under the profiler; None if absent."""
    path = os.path.join(REPO, "profiles", f"r01_ocean{N}_b{B}_pmc.json")
    try:
        d = json.load(open(path))["pmc_mean_per_launch"]
        k = [v for name, v in d.items() if "k_pass2" in name][0]
        return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0, f"rocprofv3 --pmc, {os.path.relpath(path, REPO)}"
    except Exception:
        return None, "no committed PMC pass for this workload/batch"


def preheat(enqueue, torch, ms):
    """Untimed: keep enqueueing the workload until `ms` of wall-clock have passed, then drain.  After any idle period
    the device runs the first few milliseconds ~25 % slower (clock/power ramp, measured: 19.5 us/step with a 1 ms
    warm-up vs 15.9 us/step steady state at 1024^2), which a short W would otherwise fold into the timed region."""
    if ms <= 0:
        return 0.0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        enqueue()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def cpu_baseline(p, h0, h0c, budget_s=12.0):
    """Literal O(N^4) port (oracle) on a vertex sample sized for ~10-20 s of one host core."""
    from oracle import oracle as O
    N = p.N
    rng = np.random.default_rng(0)
    probe = rng.choice(N * N, 8, replace=False).astype(np.int32)
    t0 = time.perf_counter()
    O.displacement_subset_f32(p, h0, h0c, 1.0, probe)
    per_vertex = (time.perf_counter() - t0) / probe.size
    count = int(max(8, min(4096, budget_s / per_vertex)))
    idx = rng.choice(N * N, count, replace=False).astype(np.int32)
    t0 = time.perf_counter()
    O.displacement_subset_f32(p, h0, h0c, 1.0, idx)
    el = time.perf_counter() - t0
    # the same model through numpy's FFT (f64, 1 thread): what a CPU *port with an FFT* achieves
    t1 = time.perf_counter()
    O.eval_fft_f64(p, h0, h0c, 1.0)
    el_fft = time.perf_counter() - t1
    # a float32 radix-2 Stockham port in C (oracle/cpu_fft_baseline.c), 1 thread and all host cores
    def time_c(nthreads, reps):
        O.cpu_fft_step_f32(p, h0, h0c, 1.0, nthreads)
        t2 = time.perf_counter()
        for r in range(reps):
            O.cpu_fft_step_f32(p, h0, h0c, 1.0 + r / 60.0, nthreads)
        return (time.perf_counter() - t2) / reps
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    el_c1 = time_c(1, 2)
    # more threads than the memory system can feed only add barrier cost: report the best thread count, name it
    cands = sorted({c for c in (cores, cores // 2, cores // 4, cores // 8, 32, 16, 8) if 1 <= c <= cores})
    el_call, best_threads = min((time_c(c, 3), c) for c in cands)
    return {
        "value": count / el, "unit": "grid-points/s", "cores": 1, "kind": "port",
        "sample": f"{count} of {N * N} vertices of one {N}x{N} step through the literal O(N^4) "
                  f"FFTMesh.Displacement restatement (oracle/fftmesh_oracle.c), {el:.1f} s; "
                  f"host has {os.cpu_count()} cores",
        "fft_port": {"value": N * N / el_fft, "unit": "grid-points/s", "cores": 1,
                     "what": "same model via numpy ifft2 in f64 (oracle.eval_fft_f64), one full step"},
        "fft_port_c": {"value": N * N / el_c1, "unit": "grid-points/s", "cores": 1,
                       "what": "float32 radix-2 Stockham port, 5 unpacked fields (oracle/cpu_fft_baseline.c), full steps"},
        "fft_port_c_all_cores": {"value": N * N / el_call, "unit": "grid-points/s", "cores": best_threads,
                                 "what": f"the same over pthreads; best of {cands} threads on the {cores}-core host"},
    }


def main():
    a = parse()
    import torch
    import mistral_water as mw
    import workloads
    from oracle import oracle as O

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # test hooks for a 1-GPU box (never set by the driver): MW_BENCH_BACKEND=gloo + MW_BENCH_SAME_DEVICE=1 run the
        # N > 1 control flow (barriers, max-over-ranks, rank-0 reporting) with every rank on cuda:0
        backend = os.environ.get("MW_BENCH_BACKEND", "nccl")
        if os.environ.get("MW_BENCH_SAME_DEVICE") == "1":
            local_rank = 0
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")
    stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.workload == "pond":
        return pond(a, mw, torch, dev, stream, barrier, dist, rank, world)
    if a.workload == "renderer1024":
        return renderer(a, mw, torch, dev, stream, barrier, dist, rank, world)

    N = {"ocean1024": 1024, "ocean4096": 4096, "ocean2048": 2048, "ocean512": 512, "ocean256": 256}[a.workload]
    NN = N * N
    p = workloads.fftmesh_params(N)
    seed = 1 + rank
    ocean = mw.Ocean(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y),
                     amplitude=p.amplitude, choppiness=p.choppiness, gravity=p.gravity, seed=seed, device=local_rank)
    ocean.set_stream(stream.cuda_stream)
    B = max(1, min(a.batch, ocean.max_batch))
    dv = torch.empty((B, NN, 3), dtype=torch.float32, device=dev)
    dn = torch.empty((B, NN, 3), dtype=torch.float32, device=dev)
    dw = torch.empty((B, NN), dtype=torch.float32, device=dev)

    # ---- parity gate before any timing (same run, same inputs) ----------------------------------------
    parity = None
    if not a.no_parity and rank == 0:
        h0, h0c = ocean.get_spectrum()
        ocean.evaluate_device([1.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        ocean.synchronize()
        vf, nf, cf, hds = O.eval_fft_f64(p, h0, h0c, 1.0, return_hds=True)
        workloads.assert_parity(dv[0].cpu().numpy(), dn[0].cpu().numpy(), dw[0].cpu().numpy()[:, None], vf, nf,
                                cf[:, :1], O.rest_mesh(p)[0], np.abs(hds).max(), tag="bench parity gate")
        parity = "ok (vs oracle f64, tol workloads.REL_TOL)"

    def run(nsteps, k0):
        k = k0
        while k < k0 + nsteps:
            nb = min(B, k0 + nsteps - k)
            ocean.evaluate_device([(kk + 1) / 60.0 for kk in range(k, k + nb)], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
            k += nb

    barrier()   # rank 0 may have spent seconds in the parity gate: line the ranks up BEFORE warming the clocks
    preheat_ms = preheat(lambda: run(B, 0), torch, a.preheat_ms)
    run(a.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    run(a.steps, a.warmup)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # ---- kernel-level timing with HIP events on the launch stream (rank 0) ----------------------------
    preheat(lambda: run(B, 0), torch, a.preheat_ms)     # the gather/all-reduce above may have let the clocks drop
    kern = ocean.profile_kernels(nsteps=B, iters=100)   # in situ: pass1/pass2 alternate as in the timed loop
    k2_ms = kern[1][1]
    roof_ach = BYTES_PASS2 * NN * B / (k2_ms * 1e-3)
    traffic, traffic_note = pmc_traffic(N, B)
    roofline = {"bound": "hbm", "kernel": "k_pass2", "achieved": roof_ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": roof_ach / HBM_PEAK, "traffic": traffic, "traffic_note": traffic_note,
                "bytes_per_launch": BYTES_PASS2 * NN * B, "launch_us": k2_ms * 1e3,
                "kernels": [{"name": nm, "us_per_launch": ms * 1e3,
                             "algorithmic_GBps": (BYTES_PASS1 if i == 0 else BYTES_PASS2) * NN * B / (ms * 1e-3) / 1e9}
                            for i, (nm, ms) in enumerate(kern)]}

    gather_ms = None
    if a.gather and dist is not None:
        last = torch.cat([dv[B - 1].reshape(-1), dn[B - 1].reshape(-1), dw[B - 1].reshape(-1)])
        bufs = [torch.empty_like(last) for _ in range(world)] if rank == 0 else None
        barrier()
        t1 = time.perf_counter()
        dist.gather(last, bufs, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3

    value = world * a.steps * NN / el
    out = {
        "metric": BASELINE_METRIC if N == 1024
        else f"ocean grid-points/sec (full spectrum->IFFT->disp->Jacobian), {N}^2 grid",
        "value": value, "unit": "grid-points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "preheat_ms": preheat_ms,
        "config": {"workload": f"FFTMesh-semantics ocean tile {N}x{N}, height+choppy+normals+Jacobian whitecap, "
                               f"t_k = k/60 s, one independent tile per GPU (seed = 1 + rank)",
                   "grid": N, "steps_per_enqueue": B, "tiles": world, "semantics": "MW_SEM_FFTMESH",
                   "parallelism": f"tile{world}"},
        "hbm_roofline_frac_whole_step": value / world * BYTES_PER_POINT / HBM_PEAK,
        "roofline": roofline,
        "parity": parity,
    }
    if gather_ms is not None:
        out["gather_last_step_ms"] = gather_ms
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        h0, h0c = ocean.get_spectrum()
        out["cpu_baseline"] = cpu_baseline(p, h0, h0c)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    ocean.close()
    if dist is not None:
        dist.destroy_process_group()


def renderer(a, mw, torch, dev, stream, barrier, dist, rank, world):
    """OceanRenderer semantics at the reference's shipped configuration (D/Ocean Demo.unity:296-302): 1024^2 textures,
    one GenerateTexture() per step.  The phase is stateful, so steps cannot be batched (F/FFTCommon.cginc:101-104).
    Algorithmic bytes per texel: 20 (spectrum + phase in) + 4 (phase out) + 24 + 24 (exchange) + 16 (height, disp.rgb out)
    + 16 (re-read by the normal/whitecap passes) + 16 (normal, white out) = 120."""
    import ctypes as C
    from mistral_water import _native as nat
    o = mw.Ocean(resolution=128, length=434.48, wind=(14.45, 12.0), amplitude=0.41, choppiness=0.46, mult=1.5,
                 seed=1 + rank, semantics=mw.MW_SEM_OCEANRENDERER, device=dev.index)
    o.set_stream(stream.cuda_stream)
    M = o.N

    def step():
        nat.check(nat.lib().mw_ocean_generate_texture_device(o.handle, C.c_float(1.0 / 60.0), None, None, None, None))
    preheat(lambda: [step() for _ in range(8)], torch, a.preheat_ms)
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    if rank == 0:
        v = world * a.steps * M * M / el
        print(json.dumps({
            "metric": "OceanRenderer-semantics texels/sec (dispersion+spectrum -> 2-D Stockham -> normal -> whitecap), 1024^2",
            "value": v, "unit": "texels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": "OceanRenderer GenerateTexture(), 1024x1024 textures, shipped demo parameters",
                                            "semantics": "MW_SEM_OCEANRENDERER"},
            "roofline": {"bound": "hbm", "kernel": "whole frame (3 kernels)", "achieved": v * 120 / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": v * 120 / HBM_PEAK, "traffic": None}, "cpu_baseline": None}))
    o.close()


def pond(a, mw, torch, dev, stream, barrier, dist, rank, world):
    """BASELINE configs[4]: 1M-vertex, 8-wave Gerstner displacement; 24 B/vertex algorithmic."""
    import ctypes as C
    import workloads
    from mistral_water import _native as nat
    nv = 1000 * 1000
    g = torch.linspace(-50, 50, 1001, device=dev)[:-1]
    pos = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1)
    pos = torch.stack([pos[..., 0], torch.zeros_like(pos[..., 0]), pos[..., 1]], -1).reshape(-1, 3).contiguous()
    W = np.ascontiguousarray(workloads.pond_waves8(), np.float32)
    P = workloads.POND
    B = max(1, min(a.batch, nat.lib().mw_gerstner_max_steps(8)))   # time values per launch (steps are independent in t)
    out_t = torch.empty((B, nv, 3), dtype=torch.float32, device=dev)

    def run(nsteps, k0):
        k = k0
        while k < k0 + nsteps:
            nb = min(B, k0 + nsteps - k)
            tt = np.array([(kk + 1) / 60.0 for kk in range(k, k + nb)], np.float32)
            if nb == 1:     # the single-step entry point (one sincos per wave and vertex)
                nat.check(nat.lib().mw_gerstner_displace_device(
                    C.c_void_p(pos.data_ptr()), nv, W.ctypes.data_as(C.c_void_p), 8, C.c_float(P["amplitude"]),
                    C.c_float(P["frequency"]), C.c_float(P["steepness"]), C.c_float(float(tt[0])), C.c_void_p(out_t.data_ptr()),
                    C.c_void_p(stream.cuda_stream)))
            else:
                nat.check(nat.lib().mw_gerstner_displace_steps_device(
                    C.c_void_p(pos.data_ptr()), nv, W.ctypes.data_as(C.c_void_p), 8, C.c_float(P["amplitude"]),
                    C.c_float(P["frequency"]), C.c_float(P["steepness"]), tt.ctypes.data_as(C.c_void_p), nb,
                    C.c_void_p(out_t.data_ptr()), C.c_void_p(stream.cuda_stream)))
            k += nb

    # parity gate on a sample of the lattice, first and last step of one launch
    parity = None
    if rank == 0 and not a.no_parity:
        from oracle import oracle as O
        run(B, 0)
        torch.cuda.synchronize()
        idx = np.random.default_rng(0).integers(0, nv, 4096)
        hp = pos.cpu().numpy()[idx]
        for kk in (0, B - 1):
            want = O.gerstner_f64(hp, W, P["amplitude"], P["frequency"], P["steepness"], np.float32((kk + 1) / 60.0))
            assert np.abs(out_t[kk].cpu().numpy()[idx] - want).max() < 8e-6, "pond parity gate"
        parity = "ok (4096-vertex sample vs oracle f64, tol 8e-6)"
    barrier()
    preheat(lambda: run(B, 0), torch, a.preheat_ms)
    run(a.warmup, 0)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(a.steps, a.warmup)
    e1.record(stream)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    step_us = e0.elapsed_time(e1) * 1e3 / a.steps
    ach = BYTES_POND * nv / (step_us * 1e-6)
    real = (12.0 / B + 12.0) * nv / (step_us * 1e-6)
    if rank == 0:
        print(json.dumps({
            "metric": "pond Gerstner vertices/sec (1M vertices, 8 waves)", "value": world * a.steps * nv / el,
            "unit": "vertices/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pond: 1000x1000 vertex lattice, 8 Gerstner waves (SURVEY.md 8d config 5), t_k = k/60 s",
                       "steps_per_launch": B},
            "roofline": {"bound": "hbm", "kernel": "k_gerstner_steps<8>" if B > 1 else "k_gerstner",
                         "achieved": real / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": real / HBM_PEAK, "traffic": None,
                         "us_per_step": step_us,
                         "note": f"one launch of {B} time values must move 12 B/vertex of positions once and 12 B/vertex per "
                                 f"step of results: (12/{B} + 12) B/vertex/step.  At SURVEY 8d's per-step figure of 24 B/vertex "
                                 f"(position read counted every step) the same time is {ach / 1e9:.0f} GB/s"},
            "parity": parity, "cpu_baseline": None}))


if __name__ == "__main__":
    main()
