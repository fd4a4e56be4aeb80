"""CPU tier: the parts of bench.py that decide WHAT is timed and WHICH counter file may be quoted -- no GPU needed.

VERDICT r2 item 2: the timed region is whole batches and the line prints the real enqueue sizes; a rocprofv3 counter summary
is quoted as `roofline.traffic` only next to numbers of the build it was measured on."""
import json
import os

import pytest

import bench


@pytest.mark.parametrize("K,maxb,B,sizes", [
    (20, 32, 20, [20]),                    # the driver's command line: ONE 20-step enqueue
    (32, 32, 32, [32]),
    (640, 32, 32, [32] * 20),
    (4000, 32, 32, [32] * 125),            # bench.py's default
    (1000, 32, 25, [25] * 40),             # largest divisor <= 32 (>= 16): whole batches, no remainder
    (100, 32, 25, [25] * 4),
    (101, 32, 32, [32, 32, 32, 5]),        # prime: full batches + one shorter enqueue, printed as such
    (7, 32, 7, [7]),
    (64, 20, 16, [16] * 4),                # --batch 20
    (1, 32, 1, [1]),
])
def test_batch_plan_times_exactly_k_steps_in_whole_batches(K, maxb, B, sizes):
    b, s = bench.batch_plan(K, maxb)
    assert (b, s) == (B, sizes)
    assert sum(s) == K and max(s) <= maxb


def test_counter_file_is_only_quoted_for_the_build_it_was_measured_on(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    path = tmp_path / "profiles" / f"{bench.PROFILE_ROUND}_ocean1024_b32_pmc.json"
    rec = {"bench_line": {"build_id": "0123456789abcdef default"},
           "pmc_mean_per_launch": {"void k_pass2_hs<1024, 16, 4, 2, false, 0>": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 3000.0},
                                   "void k_pass1<1024, 16, 1>": {"FETCH_SIZE": 10.0, "WRITE_SIZE": 30.0}}}
    path.write_text(json.dumps(rec))
    t, note = bench.pmc_traffic("ocean1024", 32, "k_pass2", "0123456789abcdef default")
    assert t == (2 * 1000.0 + 3000.0) * 1024.0 and "same build" in note     # FETCH_SIZE counts 128-B requests at 64 B: doubled
    t, note = bench.pmc_traffic("ocean1024", 32, "k_pass2", "fedcba9876543210 default")
    assert t is None and "not quoted" in note                                # another build: refused, and the note says why
    t, note = bench.pmc_traffic("ocean1024", 32, "k_pass", "0123456789abcdef default")
    assert t == (2 * 1010.0 + 3030.0) * 1024.0                               # several kernels of one call: their sum
    t, note = bench.pmc_traffic("ocean1024", 20, "k_pass2", "0123456789abcdef default")
    # another batch size without a pass of its own: the nearest pass of the SAME build, per step, and the note says so
    assert t == (2 * 1000.0 + 3000.0) * 1024.0 * 20 / 32 and "scaled per step" in note and "b32" in note
    t, note = bench.pmc_traffic("ocean1024", 20, "k_pass2", "fedcba9876543210 default")
    assert t is None and "not quoted" in note                                # ... never across builds
    t, note = bench.pmc_traffic("ocean2048", 32, "k_pass2", "0123456789abcdef default")
    assert t is None and "no committed PMC pass" in note                     # another workload: nothing to quote
    rec["bench_line"] = None                                                 # a round-2 style file without a build id
    path.write_text(json.dumps(rec))
    t, note = bench.pmc_traffic("ocean1024", 32, "k_pass2", "0123456789abcdef default")
    assert t is None and "not quoted" in note


def test_metric_and_bytes_are_baselines():
    base = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "BASELINE.json")))
    assert bench.BASELINE_METRIC == base["metric"]
    assert bench.BYTES_PER_POINT == 92 and bench.BYTES_PASS1 + bench.BYTES_PASS2 == 92      # SURVEY.md 8d canonical figure


def test_pass1_counters_are_only_scaled_across_equal_time_groups(tmp_path, monkeypatch):
    """a counter pass of another batch size is quoted PER STEP only for kernels whose workgroups are independent per time-step
    (pass 2).  Pass 1 reads the spectrum once per time GROUP (5 at 20 steps per launch, 8 at 32): its counters are quoted from another batch
    size only when the group size is the same, and the line says which pass was scaled (`scaled_from_b`)."""
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    rec = {"bench_line": {"build_id": "b", "config": {"pass1_time_group": 8}},
           "pmc_mean_per_launch": {"void k_pass2_hs<1024, 16, 4, 2, false, 0>": {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 3000.0},
                                   "void k_pass1<1024, 16, 1, false>": {"FETCH_SIZE": 10.0, "WRITE_SIZE": 30.0}}}
    (tmp_path / "profiles" / f"{bench.PROFILE_ROUND}_ocean1024_b32_pmc.json").write_text(json.dumps(rec))
    r2 = bench.pmc_traffic_ex("ocean1024", 20, "k_pass2", "b")
    assert r2["scaled_from_b"] == 32 and r2["traffic"] == (2 * 1000.0 + 3000.0) * 1024.0 * 20 / 32
    r1 = bench.pmc_traffic_ex("ocean1024", 20, "k_pass1", "b", tgroup=5)          # 20 steps group by 5, the pass was taken at 8
    assert r1["traffic"] is None and "time group" in r1["note"]
    r1 = bench.pmc_traffic_ex("ocean1024", 16, "k_pass1", "b", tgroup=8)          # 16 steps group by 8 like the pass: scaled
    assert r1["scaled_from_b"] == 32 and r1["traffic"] == (2 * 10.0 + 30.0) * 1024.0 * 16 / 32
    r1 = bench.pmc_traffic_ex("ocean1024", 32, "k_pass1", "b", tgroup=8)          # a pass at exactly this batch size: as measured
    assert r1["scaled_from_b"] is None and r1["traffic"] == (2 * 10.0 + 30.0) * 1024.0


def test_repeats_scale_to_a_minimum_of_timed_work():
    """five 0.27-ms regions were 1.4 ms of evidence.  R = max(--repeats, ceil(min_timed_ms / region))."""
    assert bench.scaled_repeats(0.27e-3, 5, 50.0) == 186
    assert bench.scaled_repeats(8.2e-3, 5, 50.0) == 7
    assert bench.scaled_repeats(0.5, 5, 50.0) == 5                  # long regions: --repeats decides
    assert bench.scaled_repeats(1e-9, 5, 50.0) == 2000              # capped
    st = bench.pcts([5.0, 1.0, 3.0, 2.0, 4.0])
    assert (st["median"], st["min"], st["max"], st["n"]) == (3.0, 1.0, 5.0, 5) and st["p10"] <= st["median"] <= st["p90"]


def test_phase_guard_prints_the_checkpointed_line_when_a_multi_gpu_phase_hangs_or_fails():
    """The N > 1 line's optional phases (the RCCL gather regions, the strong-scaling sub-run) have never run on real multi-GPU hardware: a
    hang or a failure there must still leave the driver ONE line with everything measured before it, from rank 0 only."""
    import time
    for rank in (0, 1):
        printed, exits = [], []
        g = bench.PhaseGuard(rank, printed.append, exits.append)
        g.checkpoint({"value": 1.0, "with_gather": {"error": "the gather regions did not finish"}})
        g.arm(0.05, "with_gather")
        g.disarm()                                   # the phase came back: nothing happens
        time.sleep(0.12)
        assert not printed and not exits
        g.arm(0.05, "with_gather")                   # the phase hangs
        time.sleep(0.3)
        assert exits == [0] and (len(printed) == 1) == (rank == 0)
        if rank == 0:
            assert printed[0]["value"] == 1.0 and printed[0]["aborted_phase"]["phase"] == "with_gather"
        printed.clear(); exits.clear()
        g.abandon("strong: RuntimeError('x')")       # the phase failed on this rank
        assert exits == [0] and (len(printed) == 1) == (rank == 0)
        if rank == 0:
            assert "RuntimeError" in printed[0]["aborted_phase"]["error"]
    g = bench.PhaseGuard(0, lambda o: None, exits.append)
    exits.clear()
    g.abandon("nothing measured yet")                # rank 0 without a line to print: a failure exit code
    assert exits == [3]
