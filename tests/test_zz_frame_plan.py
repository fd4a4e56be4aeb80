"""GPU tier, run last: the frame-at-a-time plan of the 1024^2 FFT path (FFTMesh.Update drives ONE step per call,
S/FFTMesh.cs:60-73) against the batched plan, bit for bit.

A single-step enqueue at 256^2 / 512^2 / 1024^2 is two launches of their own (csrc/mistral_water.hip, MW_LATENCY_PLAN): pass 1 with one FIELD
per workgroup over the list of active (column job, field) pairs, a column job's fields on one XCD (k_pass1<.., FS>,
p1_frame_jobs); pass 2 with the three fields of a row block and the halo row transformed side by side by 13 row groups that
meet through LDS (k_pass2_frame).  Neither changes the arithmetic of a row or a column, so every output of a step must be
the same bit pattern whichever plan produced it.  the switches MW_FRAME_KERNEL = 0 / MW_P1_FRAME_XCD = 0 (mw_debug_set_switch) select round 3's forms of the two
launches (run-time A/B switches): the same bits again."""
import os
import subprocess
import sys

import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames_against_batch(mw, p, seed, nframes, device_frames):
    import torch
    NN = p.N * p.N
    times = [0.25 + 0.37 * k for k in range(nframes)]
    with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                  choppiness=p.choppiness, gravity=p.gravity, seed=seed) as o:
        dv = torch.empty((nframes, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((nframes, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((nframes, NN), dtype=torch.float32, device="cuda")
        o.evaluate_device(times, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())      # batched plan (all steps in one enqueue)
        o.synchronize()
        bv, bn, bw = dv.cpu().numpy(), dn.cpu().numpy(), dw.cpu().numpy()
        assert np.abs(bw).max() > 0 and np.isfinite(bv).all()
        if device_frames:      # frame plan, device pointers, back to back on the handle's stream: no host round trip between frames
            fv, fn, fw = torch.zeros_like(dv), torch.zeros_like(dn), torch.zeros_like(dw)
            torch.cuda.synchronize()
            for k in range(nframes):
                o.evaluate_device([times[k]], fv[k].data_ptr(), fn[k].data_ptr(), fw[k].data_ptr())
            o.synchronize()
            assert (fv.cpu().numpy() == bv).all() and (fn.cpu().numpy() == bn).all() and (fw.cpu().numpy() == bw).all()
        for k in (0, nframes // 2, nframes - 1):
            v, n, c = o.evaluate(times[k])                                          # frame plan (one step per call, Color whitecap)
            assert (bv[k] == v).all(), k
            assert (bn[k] == n).all(), k
            assert (bw[k] == c[:, 0]).all() and (c[:, 0] == c[:, 3]).all(), k


def test_single_step_plan_equals_batched_plan_bit_for_bit_1024(mw):
    """Twelve consecutive frames enqueued back to back (a missing barrier or a stale LDS word shows up as a frame that differs) on
    the scaled sea and on BASELINE configs[1]'s literal sea (amplitude 0.41: saturated whitecaps, large slopes)."""
    _frames_against_batch(mw, workloads.fftmesh_params(1024), 3, 12, True)
    _frames_against_batch(mw, workloads.fftmesh_config2(1024), 1, 5, True)


@pytest.mark.parametrize("N", [512, 256])
def test_single_step_plan_equals_batched_plan_bit_for_bit_small_grids(mw, N):
    """The same at 512^2 (one wave per row at 8 points per thread) and 256^2 (BASELINE configs[0]; two rows of one field per wave):
    2-row workgroups of 7 row groups there."""
    _frames_against_batch(mw, workloads.fftmesh_params(N), 7, 12, True)
    _frames_against_batch(mw, workloads.fftmesh_config2(N), 2, 5, True)


_FRAME_CHILD = r'''
import sys
sys.path[:0] = [%(repo)r, %(repo)r + "/mistral-water_amd", %(repo)r + "/tests"]
import torch; torch.cuda.init()
import mistral_water as mw, workloads
import test_zz_frame_plan as T
mw.set_switch("MW_FRAME_KERNEL", %(frame_kernel)s); mw.set_switch("MW_P1_FRAME_XCD", %(p1_xcd)s)
T._frames_against_batch(mw, workloads.fftmesh_params(1024), 5, 4, True)
T._frames_against_batch(mw, workloads.fftmesh_params(512), 5, 4, True)
T._frames_against_batch(mw, workloads.fftmesh_params(256), 5, 4, True)
print("FRAME_OK")
'''


@pytest.mark.parametrize("frame_kernel,p1_xcd", [("0", "1"), ("1", "0"), ("0", "0")])
def test_single_step_plan_switches_select_the_same_bits(frame_kernel, p1_xcd):
    """The run-time A/B switches of the two single-step launches (csrc/mw_switches.h, set through the test hook), each combination in a
    child process of its own."""
    r = subprocess.run([sys.executable, "-c", _FRAME_CHILD % {"repo": REPO, "frame_kernel": frame_kernel, "p1_xcd": p1_xcd}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FRAME_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_cpp_host_mirror_runs_and_agrees_with_the_python_mirror(mw):
    """mistral-water_amd/host/FFTMesh.hpp -- the compiled host side above the C ABI (the reference's host language, C#, has no
    toolchain in the image): host_demo drives FFTMesh.Awake() / Update() x 3, an OceanRenderer frame with the RGBA targets and
    the mesh vertex stage, and the pond material.  Its FFTMesh numbers must be those of the Python mirror with the same
    Inspector fields (same seed rule, same float32 timer arithmetic, S/FFTMesh.cs:60-73)."""
    import re
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mistral-water_amd", "host")
    subprocess.run(["make", "-C", host, "-s"], check=True)
    r = subprocess.run([os.path.join(host, "host_demo")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"FFTMesh 256\^2: timer = (\S+), max\|height\| = (\S+), colour\[0\] = (\S+)", r.stdout)
    assert m and "OceanRenderer 128^2" in r.stdout and "pond Gerstner" in r.stdout, r.stdout
    f = mw.FFTMesh()
    f.resolution, f.unitWidth, f.length, f.amplitude, f.choppiness = 256, 1.0, 256.0, 2.4e-7, 0.46
    f.wind = mw.Vector2(14.45, 12.0)
    f.Awake()
    for _ in range(3):
        f.Update(1.0 / 60.0)
    hmax = float(np.abs(f.mesh.vertices[:, 1]).max())
    assert abs(float(m.group(1)) - f.timer) < 1e-7
    assert abs(float(m.group(2)) - hmax) <= 1e-6 * hmax and abs(float(m.group(3)) - float(f.mesh.colors[0, 0])) <= 1e-6


_CZT_CHILD = r'''
import sys
import numpy as np
sys.path[:0] = [%(repo)r, %(repo)r + "/mistral-water_amd", %(repo)r + "/tests"]
import torch; torch.cuda.is_available()
import mistral_water as mw, workloads
from oracle import oracle as O
mw.set_switch("MW_DIRECT_CZT", %(direct_czt)s)      # read when a handle is created
for (N, u, L, amp, rel) in ((12, 1.0, 12.39, 0.01, 2e-5), (50, 1.0, 1.0, 1.0, %(inspector_rel)s), (65, 0.5, 40.0, 1e-5, 2e-5), (200, 1.0, 212.5, 4e-7, 2e-5),
                            (1000, 1.0, 1000.0, 1.6e-8, 2e-5), (1500, 1.0, 1530.0, 7e-9, 2e-5)):      # 1500: M = 4096 = 16^3 (LastInRegs in k_czt)
    p = O.Params(N=N, unit_width=u, length=L, wind_x=5.0 if N < 100 else 14.45, wind_y=3.0 if N < 100 else 12.0, amplitude=amp, choppiness=0.8)
    h0, h0c = O.generate_spectrum(p, 4)
    rest = O.rest_mesh(p)[0]
    with mw.Ocean(resolution=N, unit_width=u, length=L, wind=(p.wind_x, p.wind_y), amplitude=amp, choppiness=0.8) as o:
        o.set_spectrum(h0, h0c)
        for t in (0.5, 16.0):
            v, n, c = o.evaluate(t)
            vd, nd, cd, hds = O.eval_matmul_f64(p, h0, h0c, t, return_hds=True)
            workloads.assert_parity(v, n, c, vd, nd, cd, rest, rel=rel, tag=f"chirp-z N={N} t={t}", hds=hds)
print("CZT_OK")
'''


@pytest.mark.parametrize("form", ["chirp-z", "gemm"])
def test_both_forms_of_the_direct_sum(form):
    """The two forms of the separable sum through the C ABI, each in a child process with its selector set: the chirp-z form
    (csrc/czt_kernels.h, the default for N <= 2048) and the MFMA GEMM form (csrc/direct_kernels.h, MW_DIRECT_CZT=0: larger grids and
    A/B) -- the shipped scene, the Inspector defaults, an odd grid, a non-commensurate grid and N = 1000 against the f64 oracle.
    Chirp-z: the FFT path's tolerance class everywhere (2e-5 stated; measured 2-5e-7), also on the Inspector-default grid, where a
    phase reaches 3900 rad and the float32 GEMM form needs 2e-4."""
    repo = REPO
    r = subprocess.run([sys.executable, "-c", _CZT_CHILD % {"repo": repo, "inspector_rel": "2e-5" if form == "chirp-z" else "2e-4",
                                                            "direct_czt": "1" if form == "chirp-z" else "0"}],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CZT_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.parametrize("N,u,L", [(2, 1.0, 2.0), (3, 1.0, 3.0), (7, 0.8, 9.1), (12, 1.0, 12.39), (20, 1.0, 25.0), (21, 1.0, 21.0), (32, 1.0, 32.0), (33, 1.0, 33.0),
                                   (50, 1.0, 1.0), (64, 1.3, 64.0), (65, 1.0, 65.0), (100, 1.0, 100.0), (128, 0.9, 128.0)])
def test_small_grid_fused_czt_equals_three_launches(mw, oracle, N, u, L):
    """Round 5: grids with a chirp-z transform size M <= 256 (N <= 128: the shipped 12-vertex scene, the Inspector default N = 50) run the
    second axis and the assembly in ONE launch (k_czt_rows_assemble: 3 RW + 2 lines side by side, vertices / normals / whitecap straight
    from LDS), and grids with N <= 20 (transform size 64) the WHOLE step in one workgroup and one launch (k_czt_one: the plane between the
    axes never leaves LDS).  the switch MW_CZT_FUSED = 0 selects the three-launch plan at run time, MW_CZT_ONE = 0 the two-launch one: the same bits,
    hds included; and the f64 oracle at 2e-5."""
    p = oracle.Params(N=N, unit_width=u, length=L, wind_x=3.0, wind_y=2.0, amplitude=2e-4 if N > 12 else 2e-3, choppiness=0.7)
    kw = dict(resolution=N, unit_width=u, length=L, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude, choppiness=p.choppiness, seed=5)
    res = {}
    plans = {"default": {}, "two": {"MW_CZT_ONE": 0}, "three": {"MW_CZT_FUSED": 0}}
    try:
        for name, env in plans.items():
            for k in ("MW_CZT_ONE", "MW_CZT_FUSED"):
                mw.set_switch(k, env.get(k, 1))
            with mw.Ocean(**kw) as o:
                assert o.max_batch == 1
                h0, h0c = o.get_spectrum()
                res[name] = (o.evaluate(0.75), o.debug_evaluate_hds(0.75), o.profile_kernels(nsteps=1, iters=3))
    finally:
        for k in ("MW_CZT_ONE", "MW_CZT_FUSED"):
            mw.set_switch(k, 1)
    names = {k: r[2][1][0] for k, r in res.items()}
    assert ("k_czt_one" in names["default"]) == (N <= 20), names                                  # the plans really differ
    assert "rows_assemble" in names["two"] and "rows_assemble" not in names["three"] and "k_czt_one" not in names["three"], names
    for other in ("two", "three"):
        for a, b in zip(res["default"][0] + res["default"][1], res[other][0] + res[other][1]):
            assert (a == b).all(), other
    v, n, c = res["default"][0]
    vd, nd, cd, hds = oracle.eval_matmul_f64(p, h0, h0c, 0.75, return_hds=True)
    workloads.assert_parity(v, n, c, vd, nd, cd, oracle.rest_mesh(p)[0], rel=2e-5, tag=f"fused czt N={N}", hds=hds)
