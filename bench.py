#!/usr/bin/env python
"""bench.py -- ocean grid-points/sec (spectrum -> transform -> displacement -> Jacobian) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one time-step of one 1024 x 1024 FFTMesh-semantics ocean tile
(BASELINE.json configs[1]); with N GPUs every rank owns an independent tile (seed = 1 + rank, configs[2]),
no data-path collective, weak scaling.  Inputs (h0/h0conj -> packed spectrum tables) are resident in HBM
before the timed region; outputs (vertices, normals, whitecap) stay in device memory.

N = 1 drives one mw_ocean handle; N > 1 drives the product's tile API (include/mistral_water.h, mw_tiles_*): every
rank owns one tile and one rank of an RCCL communicator that the LIBRARY creates (mw_comm_unique_id on rank 0, the 128
bytes broadcast through torch.distributed, mw_tiles_create_rank everywhere).  torch.distributed carries only the id,
the barriers and the max-over-ranks of the timing; `--gather` adds the library's own gather of the last step of every
batch to rank 0 (ncclSend/ncclRecv on a side stream behind an event), overlapped with the next batch.

The default run (no --workload; what the driver runs) times the 1024^2 headline and then, at N = 1, BASELINE configs[3] (4096^2, 64 steps
in 32-step enqueues) and configs[4] (the pond, 1M vertices x 8 waves) -- each with its own parity gate, `roofline` and bounded
`cpu_baseline` -- under "configs": {"ocean4096": {...}, "pond": {...}} of the same line (--workload ocean1024 = the headline alone).
Every K-step timed region is repeated until at least --min-timed-ms (50 ms) of timed work exist, whatever K the driver passes.

Extra objects on the same line:
  roofline      dominant kernel (k_pass2) ALGORITHMIC bytes / its mean launch duration measured live with hipEvents on the
                launch stream, against the 8 TB/s HBM peak (`frac`); `real_frac` = the HBM-side bytes the committed rocprofv3
                PMC pass counted for the same launch / the same duration / 8 TB/s (what the kernel physically moves).
  cpu_baseline  the oracle's restatement of the reference CPU path timed on this box's host cores, on a bounded sample.
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

import memguard  # noqa: E402  host-memory cap (tests/memguard.py): a harness bug must end this process, not the GPU box
memguard.install()
# the host driver of these boxes supports dmabuf IPC only: without this RCCL's communicator bootstrap between processes fails with
# `hipIpcGetMemHandle: invalid argument`.  Exported by the image already; set here as well so that a launcher with a scrubbed environment works.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
# BASELINE.json's metric string, verbatim
BASELINE_METRIC = "ocean grid-points/sec (full spectrum→IFFT→disp→Jacobian), 1024² grid, 1/2/4/8 GPUs"
BYTES_PER_POINT = 92       # SURVEY.md 8d canonical algorithmic traffic of one FFTMesh step
BYTES_PASS1 = 16 + 24      # read (P,Q) + write 3 packed complex fields
BYTES_PASS2 = 24 + 28      # read 3 packed complex fields + write vertex 12 + normal 12 + whitecap 4
BYTES_POND = 24            # read position 12 + write position 12
BYTES_RENDERER = 120       # see renderer()
OR_STEPS_CHUNK = 8         # frames per pass-2 / normal-pass launch of a steps call at 1024^2 (MW_OR_STEPS_CHUNK, csrc/ocean_renderer_device.h)
PROFILE_ROUND = "r06"      # profiles/<round>_<workload>_b<B>_pmc.json carry the counters of ONE build (its build_id inside)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="all", choices=["all", "ocean1024", "ocean4096", "ocean2048", "ocean512", "ocean256", "pond", "renderer1024", "direct"],
                    help="all (default) = the ocean1024 headline + at N = 1 BASELINE configs[3] (ocean4096) and [4] (pond) under `configs`")
    ap.add_argument("--direct-n", type=int, default=1000,
                    help="direct: grid size of the non-FFT FFTMesh case (50 = the Inspector default S/FFTMesh.cs:13, 100, 1000)")
    ap.add_argument("--batch", type=int, default=32, help="time-steps per enqueue (FFTMesh steps are independent in t); renderer1024: consecutive "
                                                           "GenerateTexture() frames per enqueue (1 = one call per frame)")
    ap.add_argument("--preheat-ms", type=float, default=150.0,
                    help="untimed: keep the device busy with the workload this long before the W warm-up steps, so the "
                         "timed region starts at steady clocks (the first ~5 ms after idle run ~25 %% slower)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times (each between its own barrier + synchronize pairs); "
                         "`value` / `ms_per_step` are the MEDIAN region, min / max are printed beside it")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="the K-step region is repeated at least until this much timed work exists (R = max(--repeats, ceil(this / region)))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the frame-at-a-time figures (one time-step per enqueue on device pointers; FFTMesh.Update through host "
                         "pointers): the rocprofv3 counter passes use it so that every launch of a kernel has the same size")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "steps"],
                    help="N > 1: 'tiles' = one independent ocean per rank (BASELINE configs[2], weak scaling); 'steps' = ONE ocean, "
                         "contiguous blocks of the K time-steps per rank (SURVEY 8e axis 2, strong scaling)")
    ap.add_argument("--tiles", type=int, default=1,
                    help="renderer1024: independent oceans per GenerateTexture() (mw_ocean_create_batch), one frame of each per call")
    ap.add_argument("--gather", action="store_true",
                    help="the library's RCCL gather of the last step of every batch to rank 0, on the side stream, as a second set of timed "
                         "regions (`with_gather`).  N > 1 runs it by default (SURVEY 8d config 3: with AND without the gather)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the with_gather regions")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1: skip the `strong` object (ONE ocean, the K time-steps sharded over the ranks: --shard steps as a sub-run)")
    return ap.parse_args()


def emit(obj):
    """The ONE JSON line, as the LAST line of stdout: RCCL writes a version banner through C stdio when a communicator is
    created; it sits in libc's buffer until exit unless it is flushed first."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if obj is not None:
        print(json.dumps(obj), flush=True)


def host_cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def pmc_traffic_ex(workload, B, kernel, build_id, tgroup=None):
    """HBM-side bytes per launch of `kernel` (a substring; several kernels: their sum) from the committed rocprofv3 PMC passes
    (tools/prof_workload.sh -> profiles/<round>_<workload>_b<B>_pmc.json): (2 * FETCH_SIZE + WRITE_SIZE) KiB, i.e. with the gfx950
    correction the micro-architecture guide prescribes (FETCH_SIZE counts 128-B requests at 64 B).  Counters cannot be read from
    inside this process: this is the value measured by the same command under the profiler -- and it is only quoted when the
    counter file was taken on THIS build of the library (the bench line stored in it carries mw_build_id()).
    Returns {"traffic", "note", "scaled_from_b"}: scaled_from_b = the batch size of the pass that was scaled per step (None = a pass at
    exactly this batch size).  Scaling holds for kernels whose workgroups are independent per time-step (pass 2, the pond, the renderer);
    pass 1 reads the spectrum once per TIME GROUP, so its counters are only quoted from a pass with the same group size (`tgroup`)."""
    path = os.path.join(REPO, "profiles", f"{PROFILE_ROUND}_{workload}_b{B}_pmc.json")
    rel = os.path.relpath(path, REPO)
    scale, scaled_note, b_file = 1.0, "", None
    try:
        j = json.load(open(path))
    except Exception:
        # No counter pass at exactly this batch size (the passes are taken at the batch sizes the default and the driver's command
        # use): quote the nearest one PER STEP and say so in the note.
        import glob
        import re
        cands = []
        for f in glob.glob(os.path.join(REPO, "profiles", f"{PROFILE_ROUND}_{workload}_b*_pmc.json")):
            m = re.search(r"_b(\d+)_pmc\.json$", f)
            if m:
                cands.append((abs(int(m.group(1)) - B), int(m.group(1)), f))
        if not cands:
            return {"traffic": None, "note": f"no committed PMC pass for this workload/batch ({rel})", "scaled_from_b": None}
        _, b_file, path = min(cands)
        rel = os.path.relpath(path, REPO)
        try:
            j = json.load(open(path))
        except Exception:
            return {"traffic": None, "note": f"unreadable PMC pass ({rel})", "scaled_from_b": None}
        scale = B / float(b_file)
        scaled_note = f"; no pass at {B} steps per launch: the {b_file}-step pass scaled per step (x {scale:.3f})"
    theirs = (j.get("bench_line") or {}).get("build_id")
    if theirs != build_id:
        return {"traffic": None, "note": f"{rel} was measured on build {theirs!r}, this run is build {build_id!r}: not quoted", "scaled_from_b": None}
    if b_file is not None and "k_pass1" in kernel:
        theirs_tg = ((j.get("bench_line") or {}).get("config") or {}).get("pass1_time_group")
        if tgroup is None or theirs_tg != tgroup:
            return {"traffic": None, "scaled_from_b": None,
                    "note": f"{rel}: pass 1 reads the spectrum once per time group ({theirs_tg} there, {tgroup} here): its counters do not scale per step, not quoted"}
    ks = [v for name, v in j["pmc_mean_per_launch"].items() if kernel in name]
    if not ks or any("FETCH_SIZE" not in k or "WRITE_SIZE" not in k for k in ks):
        return {"traffic": None, "note": f"{rel} has no FETCH_SIZE / WRITE_SIZE for {kernel}", "scaled_from_b": None}
    return {"traffic": scale * sum((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 for k in ks),
            "note": f"rocprofv3 --pmc, {rel} (same build){scaled_note}", "scaled_from_b": b_file}


def pmc_traffic(workload, B, kernel, build_id, tgroup=None):
    r = pmc_traffic_ex(workload, B, kernel, build_id, tgroup)
    return r["traffic"], r["note"]


def pcts(xs):
    """median / p10 / p90 / min / max of a list (nearest rank)."""
    v = sorted(xs)
    n = len(v)

    def q(f):
        return v[min(n - 1, int(round(f * (n - 1))))]
    return {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "min": v[0], "max": v[-1], "n": n}


def scaled_repeats(region_s, repeats, min_timed_ms, cap=2000):
    """How many times the K-step region is run: at least `repeats`, and enough for `min_timed_ms` of timed work in total."""
    import math
    need = int(math.ceil(min_timed_ms * 1e-3 / max(region_s, 1e-7)))
    return int(max(1, repeats, min(cap, need)))


_hip = None


def d2h(ptr, shape, dtype=np.float32):
    """Copy a raw device pointer (an int handed out by the C ABI) into a new host array."""
    global _hip
    import ctypes as C
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(shape, dtype)
    rc = _hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2)
    assert rc == 0, f"hipMemcpy D2H failed ({rc})"
    return out


def batch_plan(K, max_batch):
    """How K timed steps are enqueued: whole enqueues of B steps (+ one shorter one only when K has no usable divisor).
    B = the largest divisor of K up to max_batch when that is >= 16 or K itself fits one enqueue, else max_batch."""
    if K <= max_batch:
        return K, [K]
    div = max(d for d in range(1, max_batch + 1) if K % d == 0)
    B = div if div >= 16 else max_batch
    sizes = [B] * (K // B) + ([K % B] if K % B else [])
    return B, sizes


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


def cpu_baseline_ocean(p, h0, h0c, gpu_step, budget_s=12.0, quick=False):
    """CPU legs of an FFTMesh workload, all on the same (h0, h0conj, t = 1.0) step the GPU parity gate evaluated:
      value         the literal O(N^4) port (oracle/fftmesh_oracle.c, one core) on a vertex sample sized for ~budget_s seconds;
      literal_f32_distance   GPU output vs that literal float32 sum at the sampled vertices (the north star's "match the
                    reference C# CPU path": measured, not assumed);
      fft_port*     numpy ifft2 in f64, and the float32 radix-2 Stockham port in C on 1 and on the best number of host threads.
    quick (the extra configs of the default line): fewer thread counts tried, one repetition each."""
    from oracle import oracle as O
    N = p.N
    rng = np.random.default_rng(0)
    probe = rng.choice(N * N, 2 if quick else 4, replace=False).astype(np.int32)
    t0 = time.perf_counter()
    O.displacement_subset_f32(p, h0, h0c, 1.0, probe)
    per_vertex = (time.perf_counter() - t0) / probe.size
    count = int(max(8, min(4096, budget_s / per_vertex)))
    idx = np.sort(rng.choice(N * N, count, replace=False)).astype(np.int32)
    t0 = time.perf_counter()
    hd, nor = O.displacement_subset_f32(p, h0, h0c, 1.0, idx)
    el = time.perf_counter() - t0
    dist = None
    if gpu_step is not None:
        v, n = gpu_step
        rest = O.rest_mesh(p)[0]
        scale = max(float(np.abs(hd).max()), 1e-30)
        dist = {"vertices": count, "t": 1.0,
                "height_rel": float(np.abs(v[idx, 1] - hd[:, 1]).max() / scale),
                "disp_rel": float(max(np.abs((rest[idx, 0] - v[idx, 0]) - hd[:, 0] * p.choppiness).max(),
                                      np.abs((rest[idx, 2] - v[idx, 2]) - hd[:, 2] * p.choppiness).max()) / scale),
                "normal_abs": float(np.abs(n[idx] - nor).max()),
                "note": "relative to max |displacement| of the sample; the literal float32 sum of N^2 terms is itself only "
                        "~1e-4 accurate (tests/test_gpu_parity.py bounds it at 3e-4)"}
        del rest
    t1 = time.perf_counter()
    O.eval_fft_f64(p, h0, h0c, 1.0)
    el_fft = time.perf_counter() - t1

    def time_c(nthreads, reps):
        O.cpu_fft_step_f32(p, h0, h0c, 1.0, nthreads)
        t2 = time.perf_counter()
        for r in range(reps):
            O.cpu_fft_step_f32(p, h0, h0c, 1.0 + r / 60.0, nthreads)
        return (time.perf_counter() - t2) / reps
    cores = host_cores()
    el_c1 = time_c(1, 2 if (N <= 1024 and not quick) else 1)
    # more threads than the memory system can feed only add barrier cost: report the best thread count, name it
    if quick:
        cands = sorted({c for c in (cores // 2, cores // 4, 32) if 1 <= c <= cores})
    else:
        cands = sorted({c for c in (cores, cores // 2, cores // 4, cores // 8, 32, 16, 8) if 1 <= c <= cores})
    el_call, best_threads = min((time_c(c, 3 if (N <= 1024 and not quick) else 1), c) for c in cands)
    return {
        "value": count / el, "unit": "grid-points/s", "cores": 1, "kind": "port",
        "sample": f"{count} of {N * N} vertices of one {N}x{N} step through the literal O(N^4) "
                  f"FFTMesh.Displacement restatement (oracle/fftmesh_oracle.c), {el:.1f} s; "
                  f"host has {os.cpu_count()} cores",
        "literal_f32_distance": dist,
        "fft_port": {"value": N * N / el_fft, "unit": "grid-points/s", "cores": 1,
                     "what": "same model via numpy ifft2 in f64 (oracle.eval_fft_f64), one full step"},
        "fft_port_c": {"value": N * N / el_c1, "unit": "grid-points/s", "cores": 1,
                       "what": "float32 radix-2 Stockham port, 5 unpacked fields (oracle/cpu_fft_baseline.c), full steps"},
        "fft_port_c_all_cores": {"value": N * N / el_call, "unit": "grid-points/s", "cores": best_threads,
                                 "what": f"the same over pthreads; best of {cands} threads on the {cores}-core host"},
    }


class PhaseGuard:
    """Watchdog around the phases that have never met a multi-GPU node (the RCCL gather, the strong-scaling sub-run): if one of them does
    not come back in time, or fails on one rank while the others sit in its collectives, rank 0 still prints the ONE line -- the last object
    handed to checkpoint(), with the phase named in it -- and the process leaves (every rank arms the same deadline, so the ranks left
    behind are released by their own timer).  A hung or failed collective must not cost the driver its scaling curve."""

    def __init__(self, rank, emit_fn, exit_fn=os._exit):
        self.rank, self.emit, self.exit = rank, emit_fn, exit_fn
        self.partial, self.timer = None, None

    def checkpoint(self, obj):
        self.partial = obj

    def _leave(self, info):
        if self.rank == 0 and self.partial is not None:
            self.partial["aborted_phase"] = info
            self.emit(self.partial)
        self.exit(0 if self.partial is not None or self.rank != 0 else 3)

    def arm(self, seconds, phase):
        import threading
        self.disarm()
        self.timer = threading.Timer(seconds, lambda: self._leave({"phase": phase, "after_s": seconds,
                                                                   "note": "did not return in time; the line carries everything measured before it"}))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def abandon(self, why):
        self.disarm()
        self._leave({"error": why, "note": "the line carries everything measured before the failing phase"})


class Env:
    """What every workload needs from the process: torch, the package, the process group and this rank's device and stream."""
    pass


def setup(a):
    import torch
    import mistral_water as mw
    from mistral_water import _native as _nat
    _nat.require_product_build("bench.py")      # numbers name a product build; an A/B variant (MW_LIB=variants/X.so) needs MW_ALLOW_LAB=1
    e = Env()
    e.torch, e.mw = torch, mw
    e.world = int(os.environ.get("WORLD_SIZE", "1"))
    e.rank = int(os.environ.get("RANK", "0"))
    e.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != e.world and e.world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={e.world}")
    e.dist = None
    e.same_device = os.environ.get("MW_BENCH_SAME_DEVICE") == "1"
    if e.world > 1:
        import torch.distributed as dist
        e.dist = dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # test hooks for a 1-GPU box (never set by the driver): MW_BENCH_BACKEND=gloo + MW_BENCH_SAME_DEVICE=1 run the
        # N > 1 control flow (barriers, max-over-ranks, rank-0 reporting) with every rank on cuda:0; an RCCL communicator
        # cannot hold one device twice, so that mode drives plain mw_ocean handles instead of the tile API
        backend = os.environ.get("MW_BENCH_BACKEND", "nccl")
        if e.same_device:
            e.local_rank = 0
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", e.local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(e.local_rank)
    e.dev = torch.device("cuda", e.local_rank)
    e.red_dev = e.dev if (e.dist is None or e.dist.get_backend() == "nccl") else torch.device("cpu")
    e.stream = torch.cuda.current_stream()

    def barrier():
        torch.cuda.synchronize()
        if e.dist is not None:
            e.dist.barrier()
        torch.cuda.synchronize()
    e.barrier = barrier
    e.hung = False

    guard = PhaseGuard(e.rank, emit)
    e.checkpoint, e.arm, e.disarm, e.abandon = guard.checkpoint, guard.arm, guard.disarm, guard.abandon
    return e


def main():
    a = parse()
    e = setup(a)
    extra = (a.workload == "all")
    if a.workload == "pond":
        out = pond(a, e)
    elif a.workload == "renderer1024":
        out = renderer(a, e)
    elif a.workload == "direct":
        out = direct(a, e)
    else:
        N = {"all": 1024, "ocean1024": 1024, "ocean4096": 4096, "ocean2048": 2048, "ocean512": 512, "ocean256": 256}[a.workload]
        out = ocean(a, e, N)
        if extra and e.world == 1 and e.rank == 0:
            # BASELINE configs[3] and [4] on the same line: their own step counts (the driver's K belongs to the
            # headline), their own parity gates, roofline objects and bounded CPU baselines; a failure of one is reported in its
            # place and never takes the headline down
            import copy
            cfgs = {}
            t_extra = time.perf_counter()
            for name, fn, over in (("ocean4096", lambda aa: ocean(aa, e, 4096, extra=True), dict(steps=64, warmup=32, batch=32)),
                                   ("pond", lambda aa: pond(aa, e, extra=True), dict(steps=3200, warmup=320, batch=32)),
                                   # the reference's shipped demo scene in its own (OceanRenderer) semantics: 32 consecutive frames per
                                   # enqueue, and one GenerateTexture() per call -- what OceanRenderer.Update drives
                                   ("renderer1024", lambda aa: renderer(aa, e, extra=True), dict(steps=640, warmup=64, batch=32, tiles=1)),
                                   ("renderer1024_frame", lambda aa: renderer(aa, e, extra=True),
                                    dict(steps=400, warmup=50, batch=1, tiles=1, no_cpu_baseline=True))):
                aa = copy.copy(a)
                for k, v in over.items():
                    setattr(aa, k, v)
                t0 = time.perf_counter()
                try:
                    cfgs[name] = fn(aa)
                except Exception as ex:      # noqa: BLE001 -- reported in the line
                    import traceback
                    cfgs[name] = {"error": repr(ex), "trace": traceback.format_exc()[-800:]}
                cfgs[name]["bench_wall_s"] = time.perf_counter() - t0
                e.torch.cuda.empty_cache()
            out["configs"] = cfgs
            out["configs_wall_s"] = time.perf_counter() - t_extra
        elif extra:
            if e.rank == 0:
                out["configs"] = None      # N > 1: the headline only (configs[3] / [4] are single-GPU configurations)
        if e.world > 1 and a.shard == "tiles" and not a.no_strong:
            # SURVEY 8e axis 2 in the same line: ONE ocean, the K time-steps in contiguous blocks per rank (no collective at all)
            import copy
            aa = copy.copy(a)
            aa.shard, aa.no_parity, aa.no_latency, aa.no_cpu_baseline = "steps", True, True, True
            e.checkpoint(dict(out, strong={"error": "the sub-run did not finish"}) if e.rank == 0 else None)
            e.arm(float(os.environ.get("MW_BENCH_PHASE_TIMEOUT", "240")), "strong (--shard steps sub-run)")
            try:
                st = ocean(aa, e, N, extra=True)
                strong = {k: st[k] for k in ("value", "unit", "ms_per_step", "scaling", "repeats", "timed_ms_total", "region_ms_stats", "per_rank")}
                strong["config"] = {k: st["config"][k] for k in ("workload", "steps_per_enqueue", "enqueues_per_region", "tiles", "parallelism")}
                strong["vs_tiles_value"] = st["value"] / out["value"] if e.rank == 0 else None
                strong["what"] = "the same K steps of ONE ocean (seed 1), rank r runs the contiguous block [lo, hi): whole-job rate = K * N^2 / the slowest rank's time"
            except Exception as ex:      # noqa: BLE001 -- the other ranks sit in the sub-run's barriers: print what exists and leave
                e.abandon(f"strong: {ex!r}")
            e.disarm()
            if e.rank == 0:
                out["strong"] = strong
    if e.dist is not None:
        e.dist.destroy_process_group()
    emit(out if e.rank == 0 else None)
    if e.hung:            # a worker thread is still blocked inside the communicator bootstrap: do not wait for it at exit
        os._exit(0)


def ocean(a, e, N, extra=False):
    """One FFTMesh-semantics ocean workload (SURVEY 8d config 2 at N = 1024, config 4 at N = 4096): returns the result object.
    extra = True: one of the additional configs of the default line (compact CPU baseline, no host-pointer frames)."""
    torch, mw, dist, rank, world, local_rank = e.torch, e.mw, e.dist, e.rank, e.world, e.local_rank
    dev, red_dev, stream, barrier, same_device = e.dev, e.red_dev, e.stream, e.barrier, e.same_device
    import workloads
    from oracle import oracle as O
    from mistral_water import parallel as par
    from mistral_water import _native as nat
    NN = N * N
    p = workloads.fftmesh_config2(N)      # SURVEY 8d config 2 / 4, literally: amplitude 0.41, choppiness 0.46, wind (14.45, 12)
    shard_steps = (a.shard == "steps" and world > 1)
    seed = 1 if shard_steps else par.tile_seed(1, rank)     # --shard steps: every rank holds the SAME ocean
    kw = dict(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
              choppiness=p.choppiness, gravity=p.gravity)
    build_id = nat.build_id()
    # K timed steps: --shard tiles -> every rank runs all K on its own tile (weak); --shard steps -> rank r runs the
    # contiguous block [lo, hi) of the K steps of the one ocean (strong)
    lo, hi = par.shard_steps(a.steps, world, rank) if shard_steps else (0, a.steps)
    k_local = hi - lo
    # MW_BENCH_FORCE_TILES=1 (test hook): the tile API path with a one-rank communicator on a 1-GPU box
    use_tiles = (world > 1 or os.environ.get("MW_BENCH_FORCE_TILES") == "1") and not shard_steps and not extra
    # test hook (gloo, every rank on cuda:0): an RCCL communicator cannot hold one device twice, so every process drives the SINGLE-process
    # form of the tile API -- its own tile behind its own one-rank communicator; the control flow (tiles, gather, strong) is the real one
    private_comm = use_tiles and same_device
    tiles = ocean = None
    tiles_note = None
    B, sizes = batch_plan(max(k_local, 1), max(1, min(a.batch, 32)))
    if private_comm:
        tiles = mw.Tiles(ntiles=1, devices=[local_rank], max_steps=B, seed=seed, **kw)
    elif use_tiles:
        # the product's tile API: the LIBRARY owns the RCCL communicator; torch.distributed only carries its 128-byte id
        box = [None]
        if rank == 0:
            try:
                box[0] = mw.Tiles.unique_id()
            except mw.MistralWaterError as ex:      # RCCL not loadable: say so in the result instead of dying on every rank
                box[0] = "ERR:" + str(ex)
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], str):
            use_tiles, tiles_note = False, box[0]
        else:
            # ncclCommInitRank is collective: if it fails or hangs on ANY rank the whole job would die without a result.
            # Create the tiles in a worker thread with a deadline, agree on the outcome over torch.distributed, and fall
            # back to plain per-rank handles (still one tile per GPU, no collective) if any rank did not make it.
            import threading
            made = {}

            def _make():
                try:
                    made["tiles"] = mw.Tiles(max_steps=B, seed=1, comm_id=box[0], rank=rank, nranks=world, device=local_rank, **kw)
                except Exception as ex:      # noqa: BLE001 -- reported in the result line
                    made["err"] = repr(ex)
            th = threading.Thread(target=_make, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("MW_BENCH_TILES_TIMEOUT", "120")))
            ok_local = 1 if "tiles" in made else 0
            if dist is not None:
                flag = torch.tensor([ok_local], dtype=torch.int32, device=red_dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok_all = int(flag.item())
            else:
                ok_all = ok_local
            if ok_all:
                tiles = made["tiles"]
            else:
                use_tiles = False
                tiles_note = made.get("err", "mw_tiles_create_rank did not return in time on some rank" if ok_local else "timed out")
                if "tiles" in made:
                    made["tiles"].close()
                e.hung = e.hung or th.is_alive()
    if not use_tiles:
        ocean = mw.Ocean(seed=seed, device=local_rank, **kw)
        ocean.set_stream(stream.cuda_stream)
        dv = torch.empty((B, NN, 3), dtype=torch.float32, device=dev)
        dn = torch.empty((B, NN, 3), dtype=torch.float32, device=dev)
        dw = torch.empty((B, NN), dtype=torch.float32, device=dev)

    import ctypes as C
    # Arguments of an enqueue are marshalled ONCE, outside every timed region (the float32 time array, the ctypes pointers: ~5-8 us of
    # interpreter time per call, 2-3 % of a 20-step region): inside a region there is the C-ABI call itself and nothing else.
    _lib = nat.lib()
    _prepared = {}

    def prepare(times):
        key = tuple(times)
        if key not in _prepared:
            tt = np.ascontiguousarray(times, np.float32)
            if use_tiles:
                call = (_lib.mw_tiles_evaluate, (tiles._h, tt.ctypes.data_as(C.c_void_p), int(tt.size), nat.MW_OUT_WHITE_SCALAR))
            else:
                call = (_lib.mw_ocean_evaluate_device,
                        (ocean._h, tt.ctypes.data_as(C.c_void_p), int(tt.size), C.c_void_p(dv.data_ptr()), C.c_void_p(dn.data_ptr()),
                         C.c_void_p(dw.data_ptr()), nat.MW_OUT_WHITE_SCALAR))
            _prepared[key] = (tt, call)      # tt kept alive with the call
        return _prepared[key][1]

    def enqueue(times):
        fn, args = prepare(times)
        nat.check(fn(*args))

    def sync():
        if use_tiles:
            tiles.synchronize()
        torch.cuda.synchronize()

    # ---- parity gate before any timing (same run, same inputs), rank 0, on whichever path is timed: the single handle, or
    # the rank's tile THROUGH the tile API (its spectrum, its output buffers) -------------------------------------------------
    parity, gpu_step, h0 = None, None, None
    if not a.no_parity and rank == 0:
        if use_tiles:
            h0, h0c = tiles.get_spectrum(0)
            tiles.evaluate([1.0])
            tiles.synchronize()
            pv, pn, pw = tiles.outputs(0)
            gpu_step = (d2h(pv, (NN, 3)), d2h(pn, (NN, 3)))
            gw = d2h(pw, (NN,))
        else:
            h0, h0c = ocean.get_spectrum()
            ocean.evaluate_device([1.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
            ocean.synchronize()
            gpu_step = (dv[0].cpu().numpy(), dn[0].cpu().numpy())
            gw = dw[0].cpu().numpy()
        vf, nf, cf, hds = O.eval_fft_f64(p, h0, h0c, 1.0, return_hds=True)
        workloads.assert_parity(gpu_step[0], gpu_step[1], gw[:, None], vf, nf,
                                cf[:, :1], O.rest_mesh(p)[0], np.abs(hds).max(), tag="bench parity gate", hds=hds)
        parity = "ok (vs oracle f64, tol workloads.REL_TOL" + (", through mw_tiles_*)" if use_tiles else ")")
        del vf, nf, cf, hds, gw

    gathers = [0]

    _plans = {}

    def plan(sz, k0):
        key = (tuple(sz), k0)
        if key not in _plans:
            calls, k = [], k0
            for nb in sz:
                calls.append((prepare(par.step_times(k, k + nb)), nb))
                k += nb
            _plans[key] = calls
        return _plans[key]

    def run(sz, k0, gather=False):
        """Enqueue the batches `sz` (a list of enqueue sizes) starting at time-step index k0."""
        for (fn, args), nb in plan(sz, k0):
            nat.check(fn(*args))
            if gather:      # the previous batch's tiles travel on the side stream while this batch computes
                tiles.gather(step=nb - 1, root=0)
                gathers[0] += 1
    warm_sizes = [B] * max(1, -(-a.warmup // B)) if a.warmup > 0 else []     # whole batches, >= W steps

    plan(sizes, lo)      # the timed plan's calls exist before the first region
    barrier()   # rank 0 may have spent seconds in the parity gate: line the ranks up BEFORE warming the clocks
    preheat_ms = preheat(lambda: run([B], 0), torch, a.preheat_ms)
    run(warm_sizes, 0)
    sync()

    def wall_region(gather=False):
        """EXACTLY the K steps between barrier + synchronize pairs, by the wall clock (the contract's figure).  Nothing but the
        enqueues sits inside: no event records (they are a stream operation each, ~5 us of host time before the first launch)."""
        barrier()
        t0 = time.perf_counter()
        run(sizes, lo, gather=gather)
        if not use_tiles:
            while not stream.query():   # spin until the stream drains: a blocking synchronize adds its wake-up latency (tens of
                pass                    # microseconds) to a region that is a fraction of a millisecond at the driver's K = 20
        sync()
        el = time.perf_counter() - t0
        barrier()
        (local_gather if gather else local_regions).append(el)
        return par.max_over_ranks(el, dist, red_dev)      # the job took as long as its slowest rank
    local_regions, local_gather = [], []

    def event_region():
        """The same K steps by HIP events on the launch stream (what the device spent; launch latency of the first enqueue and the
        host's wake-up are outside)."""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        ev0.record(stream)
        run(sizes, lo)
        ev1.record(stream)
        sync()
        el = ev0.elapsed_time(ev1) * 1e-3
        barrier()
        return par.max_over_ranks(el, dist, red_dev) if dist is not None else el

    # R: at least --repeats regions and at least --min-timed-ms of timed work in total (at the driver's K = 20 a
    # region is 0.27 ms -- five of them were 1.4 ms of evidence).  The pilot region sizes R and is not counted; every rank uses
    # rank 0's figure (wall_region already returns the max over ranks, identical everywhere).
    pilot = wall_region()
    R = scaled_repeats(pilot, a.repeats, a.min_timed_ms)
    regions = [wall_region() for _ in range(R)]
    while sum(regions) * 1e3 < a.min_timed_ms and len(regions) < 4000:      # regions faster than the pilot: keep going until the total is there
        regions.append(wall_region())
    R = len(regions)
    el = float(np.median(regions))
    del local_regions[0]            # the pilot
    R_ev = min(R, 64)
    ev_regions = [event_region() for _ in range(R_ev)] if not use_tiles else None
    el_events = float(np.median(ev_regions)) if ev_regions else None
    # ---- kernel-level timing with HIP events on the launch stream (rank 0) ----------------------------
    prof = tiles_ocean = None
    if use_tiles:
        # the tile's own mw_ocean handle (borrowed) runs the in-situ profile on the tile's compute stream
        tiles_ocean = mw.Ocean.__new__(mw.Ocean)
        tiles_ocean._h = C.c_void_p(mw.lib().mw_tiles_ocean(tiles._h, 0))
        prof = tiles_ocean
    else:
        prof = ocean
    preheat(lambda: run([B], 0), torch, a.preheat_ms)     # the all-reduce above may have let the clocks drop
    iters = 100 if N <= 1024 else (40 if N == 2048 else 16)
    kstats = prof.profile_kernels_stats(nsteps=B, iters=iters)     # in situ: pass1/pass2 alternate as in the timed loop, SAME batch size
    kern = [(nm, st["mean"]) for nm, st in kstats]
    tgroup = int(mw.lib().mw_debug_pass1_time_group(prof._h, B))
    kern32 = prof.profile_kernels(nsteps=32, iters=50) if (B != 32 and not extra) else None   # context only: the same kernels at the full batch
    if tiles_ocean is not None:
        tiles_ocean._h = None                            # borrowed: the tiles own it
    wl_name = f"ocean{N}"
    k2_ms = kern[1][1]
    k2_med = kstats[1][1]["median"]
    roof_ach = BYTES_PASS2 * NN * B / (k2_ms * 1e-3)
    tr2 = pmc_traffic_ex(wl_name, B, "k_pass2", build_id)
    tr1 = pmc_traffic_ex(wl_name, B, "k_pass1", build_id, tgroup=tgroup)
    traffic, traffic_note, traffic1 = tr2["traffic"], tr2["note"], tr1["traffic"]
    stale = None
    if traffic is None:      # context only, never `traffic`: the last committed counter pass of this kernel, whatever build it was
        for rnd in ("r06", "r05", "r04", "r02"):
            try:
                j = json.load(open(os.path.join(REPO, "profiles", f"{rnd}_{wl_name}_b32_pmc.json")))["pmc_mean_per_launch"]
                k = [v for name, v in j.items() if "k_pass2" in name][0]
                stale = {"bytes_per_point": (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / (NN * 32),
                         "source": f"profiles/{rnd}_{wl_name}_b32_pmc.json",
                         "note": "counters of an EARLIER build of the same kernel (32-step launches): not this build's traffic"}
                break
            except Exception:
                pass
    roofline = {"bound": "hbm", "kernel": "k_pass2", "achieved": roof_ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": roof_ach / HBM_PEAK, "traffic": traffic, "traffic_note": traffic_note, "traffic_scaled_from_b": tr2["scaled_from_b"],
                "real_frac": (traffic / (k2_ms * 1e-3) / HBM_PEAK) if traffic else None,
                "traffic_previous_build": stale,
                "physical_bytes_per_point": (traffic / (NN * B)) if traffic else None,
                "bytes_per_launch": BYTES_PASS2 * NN * B, "algorithmic_bytes_per_point": BYTES_PASS2, "launch_us": k2_ms * 1e3,
                # the same launches as a distribution (mw_ocean_profile_kernels_stats): `frac` above is bytes / the MEAN launch duration as
                # the contract prescribes; a rocprofv3 kernel trace of this command (profiles/<round>_*_kernel_pcts.json: median, p10, p90
                # after warm-up) reproduces the median
                "launch_us_stats": {k: v * 1e3 for k, v in kstats[1][1].items()}, "launches_timed": iters,
                "frac_at_median_launch": BYTES_PASS2 * NN * B / (k2_med * 1e-3) / HBM_PEAK,
                "steps_per_launch": B,
                "at_full_batch": None if kern32 is None else {
                    "steps_per_launch": 32, "launch_us": kern32[1][1] * 1e3, "frac": BYTES_PASS2 * NN * 32 / (kern32[1][1] * 1e-3) / HBM_PEAK,
                    "pass1_launch_us": kern32[0][1] * 1e3,
                    "what": "context, NOT the timed region: the same two kernels in 32-step launches (the library's largest enqueue); "
                            f"the timed region and `frac` above are {B}-step launches"},
                # per kernel: its time, its SHARE of the 92-B bookkeeping figure (pass 1: 40 B, pass 2: 52 B) and what the counters
                # say it physically moved.  Pass 1 does not move its 40-B share (the spectrum is read once per time group and two of
                # its three fields are half-stored: 18.6 B physical), so share / time exceeds the HBM peak BY CONSTRUCTION there: it is
                # printed as `share_GBps_bookkeeping`, never as a bandwidth; `physical_GBps` is the bandwidth.
                "kernels": [{"name": nm, "us_per_launch": ms * 1e3, "us_per_launch_median": kstats[i][1]["median"] * 1e3,
                             "us_per_launch_p10": kstats[i][1]["p10"] * 1e3, "us_per_launch_p90": kstats[i][1]["p90"] * 1e3,
                             "share_bytes_per_point": (BYTES_PASS1 if i == 0 else BYTES_PASS2),
                             "share_GBps_bookkeeping": (BYTES_PASS1 if i == 0 else BYTES_PASS2) * NN * B / (ms * 1e-3) / 1e9,
                             "physical_bytes_per_point": (tr / (NN * B)) if tr else None,
                             "physical_GBps": (tr / (ms * 1e-3) / 1e9) if tr else None,
                             "traffic_scaled_from_b": sc}
                            for i, ((nm, ms), tr, sc) in enumerate(zip(kern, (traffic1, traffic), (tr1["scaled_from_b"], tr2["scaled_from_b"])))]}

    # ---- the frame-at-a-time path (what FFTMesh.Update, S/FFTMesh.cs:60-73, drives): untimed region, rank 0 ------------
    frame = None
    if not a.no_latency and not use_tiles and rank == 0:
        for _ in range(20):
            enqueue([1.0])
        torch.cuda.synchronize()
        nfr = 200 if N <= 1024 else 50
        t2 = time.perf_counter()
        for k in range(nfr):
            enqueue([(k + 1) / 60.0])
        torch.cuda.synchronize()
        single_us = (time.perf_counter() - t2) / nfr * 1e6
        frame = {"device_us_per_step": single_us, "device_points_per_s": NN / (single_us * 1e-6),
                 "device_frac_of_hbm_roofline": NN * BYTES_PER_POINT / (single_us * 1e-6) / HBM_PEAK,
                 "what": f"one time-step per call: device pointers, {nfr} calls back to back (mw_ocean_evaluate_device, nsteps = 1)"}
        if not extra:
            hv, hn, hc = (np.empty((NN, 3), np.float32), np.empty((NN, 3), np.float32), np.empty((NN, 4), np.float32))

            def host_frames(n):
                ocean.evaluate_into(1.0, hv, hn, hc)
                t3 = time.perf_counter()
                for k in range(n):
                    ocean.evaluate_into((k + 1) / 60.0, hv, hn, hc)
                return (time.perf_counter() - t3) / n * 1e3
            reps = 10 if N <= 1024 else 2
            ms_pageable = host_frames(reps)
            for arr in (hv, hn, hc):
                mw.host_register(arr)
            ms_registered = host_frames(reps)
            for arr in (hv, hn, hc):
                mw.host_unregister(arr)
            del hv, hn, hc
            frame.update({"host_ms_per_frame_pageable": ms_pageable, "host_ms_per_frame_registered": ms_registered,
                          "what": frame["what"] + "; host pointers = mw_ocean_evaluate into Vector3[] / Vector3[] / Color[] arrays (40 B per point "
                                                  "over PCIe), pageable and page-locked once with mw_host_register"})

    k_total = a.steps if shard_steps else world * a.steps
    value = k_total * NN / el
    per_gpu = value / world
    phys_pt = ((traffic or 0) + (traffic1 or 0)) / (NN * B) if (traffic and traffic1) else None
    rp = pcts(regions)
    out = {
        "metric": BASELINE_METRIC if N == 1024
        else f"ocean grid-points/sec (full spectrum->IFFT->disp->Jacobian), {N}^2 grid",
        "value": value, "unit": "grid-points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if shard_steps else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "preheat_ms": preheat_ms, "build_id": build_id,
        "event_ms_per_step": (el_events / a.steps * 1e3) if el_events is not None else None,
        "repeats": R, "timed_ms_total": sum(regions) * 1e3,
        "region_ms": [round(x * 1e3, 5) for x in regions[:8]] + (["..."] if R > 8 else []),
        "region_ms_stats": {k: (round(v * 1e3, 5) if k != "n" else v) for k, v in rp.items()},
        "ms_per_step_min": min(regions) / a.steps * 1e3, "ms_per_step_max": max(regions) / a.steps * 1e3,
        "ms_per_step_p10": rp["p10"] / a.steps * 1e3, "ms_per_step_p90": rp["p90"] / a.steps * 1e3,
        "event_region_ms": [round(x * 1e3, 5) for x in ev_regions[:8]] if ev_regions else None,
        "event_region_ms_stats": {k: (round(v * 1e3, 5) if k != "n" else v) for k, v in pcts(ev_regions).items()} if ev_regions else None,
        "wall_over_events": (el / el_events) if el_events else None,
        "timing": f"median of {R} runs of the K-step region (>= --repeats {a.repeats} and >= {a.min_timed_ms:g} ms of timed work in total), each "
                  f"between barrier + torch.cuda.synchronize() pairs (wall clock, max over ranks); event_* = the same region by HIP events on "
                  f"the launch stream, {R_ev} separate runs",
        "config": {"workload": f"SURVEY 8d config {2 if N == 1024 else 4 if N == 4096 else '2 at another N'}, literally: "
                               f"FFTMesh-semantics ocean {N}x{N}, height+choppy+normals+Jacobian whitecap, "
                               f"unit_width {p.unit_width:g}, length {p.length:g}, wind ({p.wind_x:g}, {p.wind_y:g}), "
                               f"amplitude {p.amplitude:g}, choppiness {p.choppiness:g}, t_k = k/60 s, "
                               + ("ONE ocean, the K time-steps sharded over the ranks in contiguous blocks; " if shard_steps
                                  else "one independent tile per GPU (seed = 1 + rank); ")
                               + f"the timed region is {len(sizes)} enqueue(s) of {sorted(set(sizes), reverse=True)} time-steps "
                                 f"(whole batches; a throughput figure -- the per-frame figures are in `frame_at_a_time`)",
                   "grid": N, "steps_per_enqueue": B, "enqueue_sizes_timed": sizes if len(sizes) <= 4 else [sizes[0], "...", sizes[-1]],
                   "enqueues_timed": len(sizes) * R, "enqueues_per_region": len(sizes), "warmup_steps_run": sum(warm_sizes), "pass1_time_group": tgroup,
                   "tiles": 1 if shard_steps else world, "semantics": "MW_SEM_FFTMESH",
                   "parallelism": (f"steps{world}" if shard_steps else f"tile{world}"),
                   "api": "mw_tiles_* (library-owned RCCL communicator)" if use_tiles
                   else ("mw_ocean_*" + (f" (tile API unavailable: {tiles_note})" if tiles_note else ""))},
        "frame_at_a_time": frame,
        "single_step_us": frame["device_us_per_step"] if frame else None,
        "hbm_roofline_frac_whole_step": per_gpu * BYTES_PER_POINT / HBM_PEAK,
        "hbm_real_frac_whole_step": (per_gpu * phys_pt / HBM_PEAK) if phys_pt else None,
        "physical_bytes_per_point_whole_step": phys_pt,
        "roofline": roofline,
        "parity": parity,
    }
    # ---- the same K steps again with the per-batch gather to rank 0 overlapped (default at N > 1), under the watchdog ----
    el_gather = gather_regions = None
    do_gather = use_tiles and (a.gather or world > 1) and not a.no_gather
    if do_gather:
        e.checkpoint(dict(out, with_gather={"error": "the gather regions did not finish"}) if rank == 0 else None)
        e.arm(float(os.environ.get("MW_BENCH_PHASE_TIMEOUT", "240")), "with_gather (mw_tiles_gather regions)")
        try:
            run(warm_sizes, 0, gather=True)
            sync()
            gather_regions = [wall_region(gather=True) for _ in range(R)]
            el_gather = float(np.median(gather_regions))
        except Exception as ex:      # noqa: BLE001
            e.abandon(f"with_gather: {ex!r}")      # the other ranks sit in this phase's barriers: nobody can go on together
        e.disarm()
    # N > 1: what the communicator saw and what every rank did (the driver computes efficiency from `value`; these say WHY)
    rank_el = par.all_ranks(float(np.median(local_regions)), dist, red_dev)
    k_rank = par.all_ranks(float(k_local), dist, red_dev)
    out["rccl_ranks"] = int(tiles.count) if use_tiles else 0          # mw_tiles_count(): ranks of the library's communicator (0: tile API not in use)
    out["tile_api"] = {"in_use": bool(use_tiles), "communicator": ("one per process, 1 rank each (test hook: every rank on one device)" if private_comm
                                                                    else ("ncclCommInitRank over the job's ranks" if use_tiles else None)),
                       "note": tiles_note}
    out["per_rank"] = {"median_region_ms": [round(x * 1e3, 5) for x in rank_el], "steps": [int(k) for k in k_rank],
                       "grid_points_per_s": [k * NN / x for k, x in zip(k_rank, rank_el)],
                       "slowest_over_fastest": max(rank_el) / min(rank_el)}
    if el_gather is not None:
        g_el = par.all_ranks(float(np.median(local_gather)), dist, red_dev)
        out["with_gather"] = {"value": world * a.steps * NN / el_gather, "ms_per_step": el_gather / a.steps * 1e3,
                              "relative_to_value": (world * a.steps * NN / el_gather) / value,
                              "gathers": gathers[0], "gathers_per_region": len(sizes), "bytes_per_gather_per_tile": NN * 28,
                              "root": 0, "rccl_ranks": int(tiles.count),
                              "region_ms": [round(x * 1e3, 5) for x in gather_regions[:8]],
                              "per_rank_median_region_ms": [round(x * 1e3, 5) for x in g_el],
                              "what": "the same K steps with the library's RCCL gather of every batch's last step to rank 0 "
                                      "(mw_tiles_gather: ncclSend/ncclRecv on the side stream behind an event)"}
    elif world > 1 and not shard_steps:
        out["with_gather"] = {"skipped": "--no-gather" if a.no_gather else f"tile API not in use: {tiles_note}"}
    # free the device before the CPU legs (and before the next config of the default line)
    if tiles is not None:
        tiles.close()
    if not use_tiles:
        del dv, dn, dw
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if h0 is None:
            h0, h0c = ocean.get_spectrum()
        if ocean is not None:
            ocean.close()
            ocean = None
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline_ocean(p, h0, h0c, gpu_step, budget_s=(6.0 if extra else 12.0), quick=extra)
    elif rank == 0:
        out["cpu_baseline"] = None
    if ocean is not None:
        ocean.close()
    _prepared.clear()
    _plans.clear()
    torch.cuda.empty_cache()
    return out


MFMA_F32_PEAK = 157.3e12   # FLOP/s, v_mfma_f32_32x32x2_f32 = the f32 vector rate (MI355X_MICROARCH.md)


def direct(a, e):
    """FFTMesh grids the FFT cannot express (non-power-of-two N, unit_width != length / N: the reference's
    shipped scene and its Inspector defaults) through the separable direct sum as matrix products on v_mfma_f32_32x32x2_f32.
    One step = one EvaluateWaves(t) of an N x N grid; algorithmic work 60 N^3 flop per step (csrc/direct_kernels.h): compute-
    bound, so the roofline object is against the dense f32 MFMA peak.  N = 50: the Inspector defaults (resolution 50, length 1,
    unitWidth 1, wind (1,1), amplitude 1); otherwise the config-2 sea on an N-point grid of length N (N not a power of two)."""
    from oracle import oracle as O
    import workloads
    from mistral_water import _native as nat
    mw, torch, dev, stream, barrier, dist, rank, world = e.mw, e.torch, e.dev, e.stream, e.barrier, e.dist, e.rank, e.world
    N = a.direct_n
    if N == 50:
        p = O.Params(N=50, unit_width=1.0, length=1.0, wind_x=1.0, wind_y=1.0, amplitude=1.0, choppiness=1.0, gravity=9.81)
    else:
        p = O.Params(N=N, unit_width=1.0, length=float(N), wind_x=14.45, wind_y=12.0, amplitude=1.5e-8 * (1024.0 / N) ** 2, choppiness=0.46)
    NN = N * N
    o = mw.Ocean(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                 choppiness=p.choppiness, gravity=p.gravity, seed=1 + rank, device=dev.index)
    assert o.max_batch == 1, "this grid is FFT-expressible: not the direct path"
    o.set_stream(stream.cuda_stream)
    dv = torch.empty((NN, 3), dtype=torch.float32, device=dev)
    dn = torch.empty((NN, 3), dtype=torch.float32, device=dev)
    dw = torch.empty((NN,), dtype=torch.float32, device=dev)

    def step(k):
        o.evaluate_device([(k + 1) / 60.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
    parity = None
    h0 = h0c = None
    if rank == 0 and not a.no_parity:
        h0, h0c = o.get_spectrum()
        o.evaluate_device([1.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        o.synchronize()
        vd, nd, cd, hds = O.eval_matmul_f64(p, h0, h0c, 1.0, return_hds=True)
        workloads.assert_parity(dv.cpu().numpy(), dn.cpu().numpy(), dw.cpu().numpy()[:, None], vd, nd, cd[:, :1], O.rest_mesh(p)[0],
                                rel=2e-5, tag="bench parity gate (direct)", hds=hds)
        parity = "ok (vs oracle f64 matmul form, rel 2e-5)"
    barrier()
    preheat(lambda: [step(k) for k in range(4)], torch, a.preheat_ms)
    for k in range(a.warmup):
        step(k)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    barrier()
    if dist is not None:
        from mistral_water import parallel as par
        el = par.max_over_ranks(el, dist, dev if dist.get_backend() == "nccl" else torch.device("cpu"))
    kern = o.profile_kernels(nsteps=1, iters=50)
    gemm_ms = kern[0][1]
    flops = 60.0 * N ** 3
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if h0 is None:
            h0, h0c = o.get_spectrum()
        rng = np.random.default_rng(0)
        probe = rng.choice(NN, 4, replace=False).astype(np.int32)
        t1 = time.perf_counter()
        O.displacement_subset_f32(p, h0, h0c, 1.0, probe)
        per_vertex = (time.perf_counter() - t1) / probe.size
        count = int(max(8, min(NN, 12.0 / per_vertex)))
        idx = np.sort(rng.choice(NN, count, replace=False)).astype(np.int32)
        t1 = time.perf_counter()
        O.displacement_subset_f32(p, h0, h0c, 1.0, idx)
        elc = time.perf_counter() - t1
        t1 = time.perf_counter()
        O.eval_matmul_f64(p, h0, h0c, 1.0)
        elm = time.perf_counter() - t1
        cpu = {"value": count / elc, "unit": "grid-points/s", "cores": 1, "kind": "port",
               "sample": f"{count} of {NN} vertices of one {N}x{N} step through the literal O(N^4) FFTMesh.Displacement restatement "
                         f"(oracle/fftmesh_oracle.c), {elc:.1f} s; host has {os.cpu_count()} cores",
               "separable_f64_blas": {"value": NN / elm, "unit": "grid-points/s", "cores": host_cores(),
                                      "what": "the same step as two complex128 matrix products through numpy/BLAS (oracle.eval_matmul_f64), all host cores"}}
    out = None
    if rank == 0:
        v = world * a.steps * NN / el
        czt = "k_czt" in kern[0][0]
        step_ms = kern[0][1] + kern[1][1]
        if czt:
            # chirp-z form (csrc/czt_kernels.h; the default for N <= 2048): O(N^2 log N), memory-bound.  Algorithmic bytes per grid point by
            # SURVEY 8d's general formula 16 + 16 F + out with F = 3 Hermitian-packed complex planes crossing between the two axis passes
            # (round 4; five unpacked fields before): 16 (h0, h0conj) + 48 (3 planes written + read once) + 28 (vertex, normal, whitecap) = 92 B.
            bpp = 16 + 16 * 3 + 28
            roof = {"bound": "hbm", "kernel": "k_czt + assembly = one step (" + ("2 launches" if "rows_assemble" in kern[1][0] else "3 launches") + ")", "achieved": bpp * NN / (step_ms * 1e-3) / 1e9,
                    "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bpp * NN / (step_ms * 1e-3) / HBM_PEAK, "traffic": None,
                    "algorithmic_bytes_per_point": bpp, "bytes_per_launch_group": bpp * NN, "launch_group_us": step_ms * 1e3,
                    "kernels": [{"name": nm, "us_per_step": ms * 1e3} for nm, ms in kern],
                    "note": "achieved = 92 B x N^2 / the mean duration of one step's three launches (HIP events on the launch stream).  Small "
                            "grids are launch-latency-bound (N = 50: three launches of a few workgroups), and up to N ~ 1500 the whole working "
                            "set sits in the 256-MiB Infinity Cache: the fraction says how far one step is from streaming at HBM rate"}
            # What DOES bound k_czt: its transform work, priced against the VALU and the LDS.  Per line of one axis: two
            # Stockham transforms of size M (5 M log2 M flop each), the kernel product and the two chirps (6 flop per complex product);
            # every radix pass but the last-in-registers one crosses LDS once (8 B written + 8 B read per point) and reads (P - 1) / P
            # twiddles per point from the LDS table.  Lines: 3 packed planes x (N + 1 along j, then N along i).
            M = 64
            while M < 2 * N:
                M *= 2
            Pz = 16 if M >= 256 else 8
            S_full, m = 0, 1
            while m * Pz <= M:
                m *= Pz
                S_full += 1
            RLz = M // m
            n_exch = S_full if RLz > 1 else S_full - 1            # exchanges per transform (the last radix-P pass of M = P^S stays in registers)
            n_pass = S_full + (1 if RLz > 1 else 0)
            lines = 3 * (2 * N + 1)
            flop_line = 2 * 5.0 * M * np.log2(M) + 6.0 * M + 6.0 * (2 * N + 1)
            lds_line = 2 * (n_exch * 16.0 * M + (n_pass - 1) * 8.0 * M * (Pz - 1) / Pz)
            valu_peak, lds_peak = MFMA_F32_PEAK, 256 * 128 * 2.4e9   # f32 vector rate (FMA = 2 flop); 128 B/clk/CU x 256 CUs x 2.4 GHz
            b_valu, b_lds, b_hbm = lines * flop_line / valu_peak, lines * lds_line / lds_peak, bpp * NN / HBM_PEAK
            fused_small = "rows_assemble" in kern[1][0]       # N <= 128: the second axis runs inside the assembly launch
            k_czt_s = (kern[0][1] + (kern[1][1] if fused_small else 0.0)) * 1e-3
            binding = max((b_valu, "valu"), (b_lds, "lds"), (b_hbm, "hbm"))
            # What the counters say (round 6): k_czt issues ~2280 VALU instructions per wave at either size -- index
            # arithmetic, table reads and the zero half of the M = 2N padding included -- which is 4 issue cycles each on a SIMD: the INSTRUCTION
            # count, not the flop count and not the LDS, is the floor (bank conflicts 0, LDS waits 1-2 % of wave cycles, 44-46 % waiting in all).
            issue = None
            try:
                jp = json.load(open(os.path.join(REPO, "profiles", f"{PROFILE_ROUND}_direct{N}_b1_pmc.json")))
                if (jp.get("bench_line") or {}).get("build_id") == nat.build_id():
                    kz = [v for k, v in jp["pmc_mean_per_launch"].items() if "k_czt<" in k][0]
                    per_launch_s = kz["SQ_INSTS_VALU"] * 4.0 / (256 * 4) / 2.4e9        # wave-instructions x 4 cycles over 1024 SIMDs at 2.4 GHz
                    issue = {"valu_instructions_per_wave": kz["SQ_INSTS_VALU"] / kz["SQ_WAVES"], "waves_per_launch": kz["SQ_WAVES"],
                             "issue_bound_us": 2 * per_launch_s * 1e6,
                             "wait_any_over_wave_cycles": kz["SQ_WAIT_ANY"] / kz["SQ_WAVE_CYCLES"],
                             "lds_wait_over_wave_cycles": kz["SQ_WAIT_INST_LDS"] / kz["SQ_WAVE_CYCLES"],
                             "lds_bank_conflict_cycles": kz["SQ_LDS_BANK_CONFLICT"],
                             "source": f"profiles/{PROFILE_ROUND}_direct{N}_b1_pmc.json (same build)"}
            except Exception:
                pass
            roof["transform_bounds"] = {
                "transform_size": M, "points_per_thread": Pz, "lds_exchanges_per_transform": n_exch, "lines_per_step": lines,
                "flop_per_step": lines * flop_line, "lds_bytes_per_step": lines * lds_line,
                "valu_bound_us": b_valu * 1e6, "lds_bound_us": b_lds * 1e6, "hbm_bound_us": b_hbm * 1e6,
                "valu_peak_TFLOPs": valu_peak / 1e12, "lds_peak_TBps": lds_peak / 1e12,
                "binding": binding[1], "k_czt_us": k_czt_s * 1e6,
                "k_czt_frac_of_binding_bound": binding[0] / k_czt_s, "step_frac_of_binding_bound": binding[0] / (step_ms * 1e-3),
                "instruction_issue": issue,
                "k_czt_frac_of_issue_bound": (issue["issue_bound_us"] * 1e-6 / k_czt_s) if issue else None,
                "achieved_TFLOPs": lines * flop_line / k_czt_s / 1e12, "achieved_lds_TBps": lines * lds_line / k_czt_s / 1e12,
                "note": "lower bounds of one step from its transform work: flop at the f32 vector peak (every op priced as if fused), LDS bytes "
                        "(exchanges + table twiddles) at the aggregate LDS rate, 92 B per point at the HBM peak.  The achieved fractions say how "
                        "far the two k_czt launches are from the binding one; small grids are launch-bound (a few workgroups per launch)"}
            path = f"chirp-z: two Stockham transforms of size {M} per line and axis (k_czt x 2)"
        else:
            roof = {"bound": "mfma", "kernel": "k_gemm_f32_mfma", "achieved": flops / (gemm_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK / 1e12,
                    "unit": "TFLOP/s", "frac": flops / (gemm_ms * 1e-3) / MFMA_F32_PEAK, "traffic": None,
                    "flop_per_launch_group": flops, "algorithmic_flop_per_point": 60.0 * N, "launch_group_us": gemm_ms * 1e3,
                    "executed_flop_padded": 60.0 * ((N + 63) // 64 * 64) ** 3,
                    "kernels": [{"name": nm, "us_per_step": ms * 1e3} for nm, ms in kern],
                    "note": "achieved = 60 N^3 algorithmic flop of one step / the mean duration of that step's four GEMM launches "
                            "(HIP events on the launch stream); the GEMMs execute the zero-padded size"}
            path = f"direct sum as 4 MFMA GEMM launches (MW_DIRECT_CZT=0 or N > 2048), operands padded to {(N + 63) // 64 * 64}"
        out = {
            "metric": f"FFTMesh direct-sum grid-points/sec (non-FFT grid, spectrum -> separable sum -> disp -> Jacobian), {N}^2 grid",
            "value": v, "unit": "grid-points/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "build_id": nat.build_id(),
            "config": {"workload": f"FFTMesh-semantics grid {N}x{N}, unit_width {p.unit_width:g}, length {p.length:g}, wind ({p.wind_x:g}, {p.wind_y:g}), "
                                   f"amplitude {p.amplitude:.3g}, choppiness {p.choppiness:g}: not FFT-expressible, one step per call "
                                   f"(SURVEY 8f rank 2; N = 50 is the reference's Inspector default, S/FFTMesh.cs:13-19)",
                       "grid": N, "semantics": "MW_SEM_FFTMESH", "path": path},
            "roofline": roof, "parity": parity, "cpu_baseline": cpu}
    o.close()
    return out


def renderer(a, e, extra=False):
    """OceanRenderer semantics at the reference's shipped configuration (D/Ocean Demo.unity:296-302): 1024^2 textures.
    A "step" is one GenerateTexture() frame (S/OceanRenderer.cs:216-316).  --batch F > 1: F consecutive frames per enqueue
    (mw_ocean_generate_texture_steps_device: the phase chain of F/FFTCommon.cginc:101-104 walked in registers, bit-identical to F single
    calls); F = 1 or --tiles T > 1: one call per frame (of T independent oceans).
    Algorithmic bytes per texel and frame: spectrum 16 + phase in 4 + phase out 4 (once per ENQUEUE: 24 / F) + exchange 24 + 24
    + height, disp.rgb out 16 + re-read by the normal / whitecap pass 16 + normal, white out 16 = 96 + 24 / F   (120 at F = 1)."""
    import ctypes as C
    from mistral_water import _native as nat
    from oracle import oracle as O
    mw, torch, dev, stream, barrier, dist, rank, world = e.mw, e.torch, e.dev, e.stream, e.barrier, e.dist, e.rank, e.world
    from mistral_water import parallel as par
    T = max(1, a.tiles)
    F = 1 if T > 1 else max(1, min(a.batch, 32))
    kw = dict(resolution=128, length=434.48, wind=(14.45, 12.0), amplitude=0.41, choppiness=0.46, mult=1.5)
    # N > 1: one independent ocean per rank (weak), or --shard steps: ONE ocean, rank r renders the contiguous block [lo, hi) of the K frames.
    # In this semantics only the phase links the frames: a rank seeks to its block with the Dispersion pass alone (mw_ocean_advance_phase),
    # renders it, and walks the phase on to frame K so that every region starts from the same state on every rank.
    shard_steps = (a.shard == "steps" and world > 1 and T == 1)
    lo, hi = par.shard_steps(a.steps, world, rank) if shard_steps else (0, a.steps)
    o = mw.Ocean(seed=1 if shard_steps else 1 + 64 * rank, semantics=mw.MW_SEM_OCEANRENDERER, device=dev.index, ntiles=T, **kw)
    o.set_stream(stream.cuda_stream)
    M = o.N
    MM = M * M
    build_id = nat.build_id()
    DT = 1.0 / 60.0
    _lib = nat.lib()
    B, sizes = batch_plan(max(hi - lo, 1), F)
    seek = (np.full(max(lo, 1), DT, np.float32), lo, np.full(max(a.steps - hi, 1), DT, np.float32), a.steps - hi)
    dest = None
    if F > 1:       # caller-owned destinations, [F][...] like the FFTMesh enqueue's
        dest = (torch.empty((B, M, M), dtype=torch.float32, device=dev), torch.empty((B, M, M, 2), dtype=torch.float32, device=dev),
                torch.empty((B, M, M, 3), dtype=torch.float32, device=dev), torch.empty((B, M, M), dtype=torch.float32, device=dev))
    _calls = {}

    def call(nf):
        if nf not in _calls:
            if F == 1:
                _calls[nf] = (None, _lib.mw_ocean_generate_texture_device, (o.handle, C.c_float(DT), None, None, None, None))
            else:
                dts = np.full(nf, DT, np.float32)
                _calls[nf] = (dts, _lib.mw_ocean_generate_texture_steps_device,
                              (o.handle, dts.ctypes.data_as(C.c_void_p), nf) + tuple(C.c_void_p(t.data_ptr()) for t in dest))
        return _calls[nf][1:]

    def run(sz):
        for nf in sz:
            fn, args = call(nf)
            if F == 1:
                for _ in range(nf):
                    nat.check(fn(*args))
            else:
                nat.check(fn(*args))

    # ---- parity gate before any timing: the enqueue that is timed, frames 0 and last against the oracle, the phase bit for bit ----
    parity = None
    if not a.no_parity and rank == 0 and T == 1:
        import or_bounds
        rp = O.RendererParams(gravity=9.81, **{k: v for k, v in kw.items() if k != "wind"}, wind_x=kw["wind"][0], wind_y=kw["wind"][1])
        init4 = np.concatenate(o.get_spectrum(), -1)
        ph = o.get_phase().copy()
        if F > 1:
            run([B])
            torch.cuda.synchronize()
            got = {k: tuple(t[k].cpu().numpy() for t in dest) for k in sorted({0, B - 1})}
        else:       # one frame per call: the host form of the same call (same kernels, results copied out)
            got = {0: o.generate_texture(DT)}
        for k in range(B if F > 1 else 1):
            if k in got:
                oh, od, on, ow, og = O.renderer_step_f64(rp, init4, ph, DT, literal_passes=False)
                h, d, n, w = got[k]
                for x, y, nm in ((h, oh, "height"), (d, od, "disp")):
                    err, sc = float(np.abs(x - y).max()), max(float(np.abs(y).max()), 1e-6)
                    assert err <= 3e-6 * sc, f"bench parity gate, frame {k}, {nm}: {err:.3e} vs scale {sc:.3e}"
                or_bounds.assert_normal_white(n, w, on, ow, rp.length, od[..., 0], og, od[..., 1], oh, tag=f"bench parity gate frame {k}")
            else:
                O.renderer_advance_phase(rp, init4, ph, DT)
        assert (o.get_phase() == ph).all(), "bench parity gate: phase texture differs from the oracle's strict-float32 recurrence"
        parity = (f"ok (frames {sorted(got)} of a {B if F > 1 else 1}-frame enqueue vs oracle f64: height / disp 3e-6 of scale, normal / whitecap at "
                  "tests/or_bounds.py; phase texture bit for bit)")
        del got, init4

    barrier()
    preheat_ms = preheat(lambda: run([B]), torch, a.preheat_ms)
    run([B] * max(1, -(-a.warmup // B)) if a.warmup > 0 else [])
    torch.cuda.synchronize()

    def wall_region():
        barrier()
        t0 = time.perf_counter()
        if shard_steps and seek[1]:
            nat.check(_lib.mw_ocean_advance_phase(o.handle, seek[0].ctypes.data_as(C.c_void_p), seek[1]))
        run(sizes)
        if shard_steps and seek[3]:
            nat.check(_lib.mw_ocean_advance_phase(o.handle, seek[2].ctypes.data_as(C.c_void_p), seek[3]))
        while not stream.query():
            pass
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        return par.max_over_ranks(el, dist, e.red_dev)
    pilot = wall_region()
    R = scaled_repeats(pilot, a.repeats, a.min_timed_ms)
    regions = [wall_region() for _ in range(R)]
    el = float(np.median(regions))

    # ---- in situ per-launch durations (HIP events on the launch stream between the launches of the same enqueue), rank 0 ----
    kstats = frame = None
    if rank == 0:
        preheat(lambda: run([B]), torch, a.preheat_ms)
        kstats = o.profile_kernels_stats(nsteps=B if T == 1 else 1, iters=60 if B > 1 else 200)      # a batched handle: one call = T tile-frames
        if not a.no_latency and F > 1:      # what OceanRenderer.Update drives: one GenerateTexture() per call
            one = (o.handle, C.c_float(DT), None, None, None, None)
            for _ in range(50):
                nat.check(_lib.mw_ocean_generate_texture_device(*one))
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(400):
                nat.check(_lib.mw_ocean_generate_texture_device(*one))
            torch.cuda.synchronize()
            us = (time.perf_counter() - t2) / 400 * 1e6
            frame = {"device_us_per_frame": us, "device_texels_per_s": MM / (us * 1e-6),
                     "device_frac_of_hbm_roofline": MM * BYTES_RENDERER / (us * 1e-6) / HBM_PEAK,
                     "what": "one GenerateTexture() per call (mw_ocean_generate_texture_device), 400 calls back to back, 120 B per texel"}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # the oracle's restatement of the shader pipeline (oracle/ocean_renderer_oracle.c: dispersion, spectrum, normal and
        # whitecap passes in C, the 2 x 20 Stockham blits through numpy's fft2), one core, whole frames of the same texture
        rp = O.RendererParams(resolution=128, length=434.48, wind_x=14.45, wind_y=12.0, amplitude=0.41, choppiness=0.46, mult=1.5)
        init4 = O.renderer_initial_spectrum(rp, 1)
        ph = np.zeros((M, M), np.float32)
        O.renderer_step_f64(rp, init4, ph, DT, literal_passes=False)
        t1 = time.perf_counter()
        frames = 8 if extra else 40
        for _ in range(frames):
            O.renderer_step_f64(rp, init4, ph, DT, literal_passes=False)
        elc = (time.perf_counter() - t1) / frames
        cpu = {"value": MM / elc, "unit": "texels/s", "cores": 1, "kind": "port",
               "sample": f"{frames} whole GenerateTexture() frames of the same 1024^2 texture through oracle/ocean_renderer_oracle.c "
                         f"(+ numpy fft2 for the Stockham blits), {elc:.2f} s per frame; host has {os.cpu_count()} cores"}
    out = None
    if rank == 0:
        v = (a.steps if shard_steps else world * a.steps) * MM * T / el
        bytes_frame = 96.0 + 24.0 / F           # per texel and frame
        # HBM-side bytes of ONE ENQUEUE from the committed counter pass: a call per frame = its three launches; a steps call = one spectrum
        # launch + per chunk of OR_STEPS_CHUNK frames one pass-2 and one normal-pass launch + the copy of the last frame
        ktr = [None, None, None, None]
        nch = -(-B // OR_STEPS_CHUNK)
        if T == 1 and B > 1:
            parts = [pmc_traffic_ex("renderer1024", B, k, build_id) for k in ("k_or_pass1_steps", "k_or_pass2", "k_or_normal_white", "k_or_copy_frame")]
            if all(q["traffic"] is not None for q in parts):
                ktr = [parts[0]["traffic"], nch * parts[1]["traffic"], nch * parts[2]["traffic"], parts[3]["traffic"]]
                traffic, traffic_note = sum(ktr), parts[0]["note"] + f"; pass 2 and the normal pass: {nch} launches of {OR_STEPS_CHUNK} frames each per enqueue"
            else:
                traffic, traffic_note = None, next(q["note"] for q in parts if q["traffic"] is None)
        else:
            # the three kernels of one call (not k_or_init / k_or_omega / k_or_prep, which the counter file of the same process also holds)
            parts = [pmc_traffic_ex("renderer1024", B if T == 1 else T, k, build_id) for k in ("k_or_pass1", "k_or_pass2", "k_or_normal_white")]
            if all(q["traffic"] is not None for q in parts):
                ktr = [q["traffic"] for q in parts] + [None]
                traffic, traffic_note = sum(ktr[:3]), parts[0]["note"]
            else:
                traffic, traffic_note = None, next(q["note"] for q in parts if q["traffic"] is None)
        # per kernel, the bytes ITS PLAN must move per texel and frame.  Three-transform plan: pass 1 = exchange out 24 + (spectrum 16, omega 4, phase
        # in / out 8) once per enqueue; pass 2 = exchange in 24 + 16 out.  Packed plan (two transforms: height + i Dz share one; the default for
        # planar textures): exchange 16 each way; pass 1 reads (h0, h0c) or (P, Q) + omega + phase per field workgroup.  Normal / whitecap: 16 + 16.
        packed = (mw.get_switch("MW_OR_PACKED") != 0)
        # (packed pass 1 per enqueue: two field workgroups each read omega 4 + phase 4 + their coefficients 16, + (h0, h0c) of the Nyquist row, + the
        # phase out 4 = 52; the tile form reads both coefficient sets in one workgroup: 44)
        shares = ((16.0 + (44.0 if T > 1 else 52.0) / F, 32.0, 32.0) if packed else (24.0 + 28.0 / F, 40.0, 32.0))
        kernels, dom = None, None
        UL = B if T == 1 else T         # texel-frames per launch, in units of one texture
        # launches of each kernel per enqueue: a steps call runs pass 2 and the normal pass once per chunk of OR_STEPS_CHUNK frames
        nl = [1, nch if (T == 1 and B > 1) else 1, nch if (T == 1 and B > 1) else 1, 1]
        if kstats:
            kernels = [{"name": nm, "launches_per_enqueue": nl[i], "frames_per_launch": UL / nl[i],
                        "us_per_enqueue": st["mean"] * 1e3, "us_per_launch": st["mean"] * 1e3 / nl[i],
                        "us_per_launch_median": st["median"] * 1e3 / nl[i], "us_per_launch_p10": st["p10"] * 1e3 / nl[i],
                        "us_per_launch_p90": st["p90"] * 1e3 / nl[i],
                        "algorithmic_bytes_per_texel_frame": (shares[i] if i < 3 else None),
                        "physical_bytes_per_texel_frame": (ktr[i] / (MM * UL)) if ktr[i] else None,
                        "physical_GBps": (ktr[i] / (st["mean"] * 1e-3) / 1e9) if ktr[i] else None,
                        "frac": (shares[i] * MM * UL / (st["mean"] * 1e-3) / HBM_PEAK) if i < 3 else None} for i, (nm, st) in enumerate(kstats)]
            dom = max(range(3), key=lambda i: kstats[i][1]["mean"])
        rpc = pcts(regions)
        roof = {"bound": "hbm", "peak": HBM_PEAK / 1e9, "unit": "GB/s", "traffic": traffic, "traffic_note": traffic_note,
                "traffic_what": f"HBM-side bytes of one enqueue = {B if T == 1 else T} frame(s), all its kernels",
                "physical_bytes_per_texel_frame": (traffic / (MM * (B if T == 1 else T))) if traffic else None,
                "whole_frame": {"algorithmic_bytes_per_texel_frame": bytes_frame, "plan_bytes_per_texel_frame": sum(shares),
                                "plan": ("packed: two complex transforms per frame (exchange buffer 16 B per texel each way)" if packed
                                         else "three complex transforms per frame, as the shaders"),
                                "achieved": v / world * bytes_frame / 1e9,
                                "frac": v / world * bytes_frame / HBM_PEAK,
                                "real_frac": (v / world * traffic / (MM * (B if T == 1 else T)) / HBM_PEAK) if traffic else None},
                "kernels": kernels}
        if dom is not None:     # the contract's object: the dominant kernel's algorithmic bytes per launch / its mean launch duration
            k = kernels[dom]
            roof.update({"kernel": k["name"], "achieved": shares[dom] * MM * UL / (k["us_per_enqueue"] * 1e-6) / 1e9, "frac": k["frac"],
                         "launch_us": k["us_per_launch"], "frames_per_launch": k["frames_per_launch"], "launches_per_enqueue": k["launches_per_enqueue"],
                         "algorithmic_bytes_per_texel_frame": shares[dom], "bytes_per_launch": shares[dom] * MM * k["frames_per_launch"]})
        else:
            roof.update({"kernel": "whole frame (every kernel of a call)", "achieved": v / world * bytes_frame / 1e9, "frac": v / world * bytes_frame / HBM_PEAK})
        out = ({
            "metric": "OceanRenderer-semantics texels/sec (dispersion+spectrum -> 2-D Stockham -> normal -> whitecap), 1024^2",
            "value": v, "unit": "texels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if shard_steps else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "build_id": build_id, "preheat_ms": preheat_ms, "repeats": R, "timed_ms_total": sum(regions) * 1e3,
            "region_ms_stats": {k: (round(x * 1e3, 5) if k != "n" else x) for k, x in rpc.items()},
            "config": {"workload": "OceanRenderer GenerateTexture(), 1024x1024 textures, shipped demo parameters (D/Ocean Demo.unity:296-302: "
                                   "length 434.48, wind (14.45, 12), amplitude 0.41, choppiness 0.46, mult 1.5), deltaTime 1/60 s; "
                                   + (f"{T} independent ocean(s) (seed + k) per call, one frame of each per step" if F == 1 else
                                      f"one ocean, {B} consecutive frames per enqueue (mw_ocean_generate_texture_steps_device), "
                                      f"the timed region is {len(sizes)} enqueue(s)"),
                       "semantics": "MW_SEM_OCEANRENDERER", "tiles_per_call": T, "frames_per_enqueue": B if T == 1 else 1,
                       "parallelism": (f"steps{world}: ONE ocean, rank r renders frames [lo, hi) of the K after seeking there with mw_ocean_advance_phase"
                                       if shard_steps else f"tile{world}"),
                       "enqueue_sizes_timed": sizes if len(sizes) <= 4 else [sizes[0], "...", sizes[-1]],
                       "us_per_tile_frame": el / a.steps / T * 1e6},
            "frame_at_a_time": frame, "parity": parity,
            "roofline": roof, "cpu_baseline": cpu})
    o.close()
    del dest
    _calls.clear()
    torch.cuda.empty_cache()
    return out


def pond(a, e, extra=False):
    """BASELINE configs[4]: 1M-vertex, 8-wave Gerstner displacement; 24 B/vertex algorithmic (positions read once per launch of B time
    values, results written per step: (12 / B + 12) B per vertex-step)."""
    import ctypes as C
    import workloads
    from mistral_water import _native as nat
    from mistral_water import parallel as par
    mw, torch, dev, stream, barrier, dist, rank, world = e.mw, e.torch, e.dev, e.stream, e.barrier, e.dist, e.rank, e.world
    nv = 1000 * 1000
    g = torch.linspace(-50, 50, 1001, device=dev)[:-1]
    pos = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1)
    pos = torch.stack([pos[..., 0], torch.zeros_like(pos[..., 0]), pos[..., 1]], -1).reshape(-1, 3).contiguous()
    W = np.ascontiguousarray(workloads.pond_waves8(), np.float32)
    P = workloads.POND
    B = max(1, min(a.batch, nat.lib().mw_gerstner_max_steps(8)))   # time values per launch (steps are independent in t)
    out_t = torch.empty((B, nv, 3), dtype=torch.float32, device=dev)
    lib = nat.lib()
    _calls = {}

    def call(k, nb):      # the C-ABI call of the launch that covers steps k .. k + nb - 1, marshalled once
        if (k, nb) not in _calls:
            tt = np.array([(kk + 1) / 60.0 for kk in range(k, k + nb)], np.float32)
            if nb == 1:     # the single-step entry point (one sincos per wave and vertex)
                c = (lib.mw_gerstner_displace_device,
                     (C.c_void_p(pos.data_ptr()), nv, W.ctypes.data_as(C.c_void_p), 8, C.c_float(P["amplitude"]),
                      C.c_float(P["frequency"]), C.c_float(P["steepness"]), C.c_float(float(tt[0])), C.c_void_p(out_t.data_ptr()),
                      C.c_void_p(stream.cuda_stream)))
            else:
                c = (lib.mw_gerstner_displace_steps_device,
                     (C.c_void_p(pos.data_ptr()), nv, W.ctypes.data_as(C.c_void_p), 8, C.c_float(P["amplitude"]),
                      C.c_float(P["frequency"]), C.c_float(P["steepness"]), tt.ctypes.data_as(C.c_void_p), nb,
                      C.c_void_p(out_t.data_ptr()), C.c_void_p(stream.cuda_stream)))
            _calls[(k, nb)] = (tt, c)
        return _calls[(k, nb)][1]

    def run(nsteps, k0):
        k = k0
        while k < k0 + nsteps:
            nb = min(B, k0 + nsteps - k)
            fn, args = call(k, nb)
            nat.check(fn(*args))
            k += nb

    # parity gate on a sample of the lattice, first and last step of one launch
    parity = None
    from oracle import oracle as O
    if rank == 0 and not a.no_parity:
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
    run(a.steps, a.warmup)      # the timed plan's calls exist before the first region
    torch.cuda.synchronize()

    def wall_region():
        barrier()
        t0 = time.perf_counter()
        run(a.steps, a.warmup)
        while not stream.query():
            pass
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        return par.max_over_ranks(el, dist, e.red_dev)
    pilot = wall_region()
    R = scaled_repeats(pilot, a.repeats, a.min_timed_ms)
    regions = [wall_region() for _ in range(R)]
    while sum(regions) * 1e3 < a.min_timed_ms and len(regions) < 4000:
        regions.append(wall_region())
    R = len(regions)
    el = float(np.median(regions))
    # per-launch durations of the dominant kernel: HIP events between back-to-back launches on the launch stream
    nl = 64
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nl + 1)]
    fn, args = call(a.warmup, min(B, a.steps))
    Bl = min(B, a.steps)
    for _ in range(8):
        nat.check(fn(*args))
    evs[0].record(stream)
    for i in range(nl):
        nat.check(fn(*args))
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(nl)]
    lp = pcts(launch_ms)
    launch_mean_ms = float(np.mean(launch_ms))
    bytes_launch = (12.0 + 12.0 * Bl) * nv
    real = bytes_launch / (launch_mean_ms * 1e-3)
    step_us = el / a.steps * 1e6
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # CPU_GERSTNER: the shader's own float32 arithmetic (oracle/gerstner_oracle.c), whole
        # 1M-vertex steps, one thread and all host cores (the lattice split into contiguous ranges, GIL released in C)
        from concurrent.futures import ThreadPoolExecutor
        hp = np.ascontiguousarray(pos.cpu().numpy(), np.float32)
        ho = np.empty_like(hp)
        args_c = (W, P["amplitude"], P["frequency"], P["steepness"])

        def step_cpu(nthreads, t):
            if nthreads == 1:
                O.gerstner_f32_range(hp, 0, nv, *args_c, t, ho)
                return
            cuts = np.linspace(0, nv, nthreads + 1).astype(np.int64)
            with ThreadPoolExecutor(nthreads) as ex:
                list(ex.map(lambda i: O.gerstner_f32_range(hp, int(cuts[i]), int(cuts[i + 1]), *args_c, t, ho), range(nthreads)))

        def time_cpu(nthreads, reps):
            step_cpu(nthreads, 0.0)
            t1 = time.perf_counter()
            for r in range(reps):
                step_cpu(nthreads, (r + 1) / 60.0)
            return (time.perf_counter() - t1) / reps
        reps = 3 if extra else 5
        el1 = time_cpu(1, reps)
        cores = host_cores()
        cands = sorted({c for c in ((cores // 2, 32) if extra else (cores, cores // 2, cores // 4, 32, 16, 8)) if 1 < c <= cores}) or [1]
        eln, bestn = min((time_cpu(c, reps), c) for c in cands)
        cpu = {"value": nv / el1, "unit": "vertices/s", "cores": 1, "kind": "port",
               "sample": f"{reps} whole steps of the same 1M-vertex, 8-wave lattice through the float32 restatement of Gerstner() "
                         f"(oracle/gerstner_oracle.c), {el1 * 1e3:.0f} ms per step; host has {os.cpu_count()} cores",
               "all_cores": {"value": nv / eln, "unit": "vertices/s", "cores": bestn,
                             "what": f"the same split over host threads; best of {cands} on the {cores}-core host"}}
    out = None
    if rank == 0:
        build_id = nat.build_id()
        tr = pmc_traffic_ex("pond", Bl, "k_gerstner", build_id)
        traffic, traffic_note = tr["traffic"], tr["note"]
        rp = pcts(regions)
        out = {
            "metric": "pond Gerstner vertices/sec (1M vertices, 8 waves)", "value": world * a.steps * nv / el, "build_id": build_id,
            "unit": "vertices/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": R, "timed_ms_total": sum(regions) * 1e3,
            "region_ms_stats": {k: (round(v * 1e3, 5) if k != "n" else v) for k, v in rp.items()},
            "config": {"workload": "pond: 1000x1000 vertex lattice, 8 Gerstner waves (SURVEY.md 8d config 5), t_k = k/60 s",
                       "steps_per_launch": Bl},
            "roofline": {"bound": "hbm", "kernel": "k_gerstner_steps<8>" if Bl > 1 else "k_gerstner",
                         "achieved": real / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": real / HBM_PEAK, "traffic": traffic,
                         "traffic_note": traffic_note, "traffic_scaled_from_b": tr["scaled_from_b"],
                         "physical_bytes_per_vertex_step": (traffic / (nv * Bl)) if traffic else None,
                         "real_frac": (traffic / (launch_mean_ms * 1e-3) / HBM_PEAK) if traffic else None,
                         "bytes_per_launch": bytes_launch, "launch_us": launch_mean_ms * 1e3, "launches_timed": nl,
                         "launch_us_stats": {k: (v * 1e3 if k != "n" else v) for k, v in lp.items()},
                         "frac_at_median_launch": bytes_launch / (lp["median"] * 1e-3) / HBM_PEAK,
                         "us_per_step": step_us,
                         "note": f"one launch of {Bl} time values moves 12 B/vertex of positions once and 12 B/vertex per step of "
                                 f"results: (12/{Bl} + 12) B/vertex/step is what the kernel must move; achieved = that per launch / the mean "
                                 f"duration of {nl} back-to-back launches (HIP events on the launch stream)"},
            "parity": parity, "cpu_baseline": cpu}
    del out_t, pos
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
