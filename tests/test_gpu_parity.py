"""GPU tier (-m gpu): the parity tests proper.  Everything goes through the C ABI of libmistral_water.so
(ctypes), with h0/h0conj injected identically into the oracle and the GPU path (mw_ocean_set_spectrum).

Tolerance (north_star: "within a stated float32 tolerance"): workloads.REL_TOL = 4e-6 of max |field| plus one
f32 ulp of the stored coordinate; index / sign-flip work (rest mesh, triangle indices, uvs, omega*t, the
whitecap edge rules) is compared bit for bit.
"""
import ctypes as C

import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def make(mw, p, seed=1):
    return mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y),
                    amplitude=p.amplitude, choppiness=p.choppiness, gravity=p.gravity, seed=seed)


def test_device_present(mw):
    assert mw.lib().mw_device_count() >= 1, "the -m gpu tier needs an MI355X"


@pytest.mark.parametrize("N", [64, 128, 256, 512, 1024])
def test_fftmesh_parity_vs_oracle_f64(mw, oracle, N):
    p = workloads.fftmesh_params(N)
    h0, h0c = oracle.generate_spectrum(p, 1)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        for t in (0.0, 1.0, 16.65):
            v, n, c = o.evaluate(t)
            vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
            workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"N={N} t={t}")
            assert (c[:, 0] == c[:, 3]).all() and (c[:, 1] == c[:, 2]).all()  # Color(xx,xx,xx,xx) :274


def test_fftmesh_256_vs_literal_f32_sample(mw, oracle):
    """BASELINE config 1 (256^2, one step): distance to the reference's literal float32 O(N^4) loop on a vertex
    sample (a full literal step is ~4 minutes of CPU).  The literal f32 sum is itself only ~1e-4 accurate."""
    p = workloads.fftmesh_params(256, choppiness=1.0)
    h0, h0c = oracle.generate_spectrum(p, 1)
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(256 * 256, 48, replace=False)).astype(np.int32)
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(1.0)
    hd, nor = oracle.displacement_subset_f32(p, h0, h0c, 1.0, idx)
    rest = oracle.rest_mesh(p)[0]
    scale = np.abs(hd).max()
    assert np.abs(v[idx, 1] - hd[:, 1]).max() < 3e-4 * scale                 # height
    assert np.abs((rest[idx, 0] - v[idx, 0]) - hd[:, 0]).max() < 3e-4 * scale  # d.x * choppiness(=1)
    assert np.abs((rest[idx, 2] - v[idx, 2]) - hd[:, 2]).max() < 3e-4 * scale
    assert np.abs(n[idx] - nor).max() < 3e-4


def test_fftmesh_1024_vs_literal_f32_sample(mw, oracle):
    """BASELINE's headline size: the distance to the reference's literal float32 O(N^4) loop (S/FFTMesh.cs:192-220),
    measured on 64 vertices of the 1024^2 bench workload (bench.py prints the same figure for its own sample)."""
    p = workloads.fftmesh_params(1024)
    h0, h0c = oracle.generate_spectrum(p, 1)
    idx = np.sort(np.random.default_rng(1).choice(1024 * 1024, 64, replace=False)).astype(np.int32)
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(1.0)
    hd, nor = oracle.displacement_subset_f32(p, h0, h0c, 1.0, idx)
    rest = oracle.rest_mesh(p)[0]
    scale = np.abs(hd).max()
    assert np.abs(v[idx, 1] - hd[:, 1]).max() < 3e-4 * scale
    assert np.abs((rest[idx, 0] - v[idx, 0]) - hd[:, 0] * p.choppiness).max() < 3e-4 * scale
    assert np.abs((rest[idx, 2] - v[idx, 2]) - hd[:, 2] * p.choppiness).max() < 3e-4 * scale
    assert np.abs(n[idx] - nor).max() < 3e-4


@pytest.mark.parametrize("N", [1024, 4096])
def test_fftmesh_survey_config2_literal_parameters(mw, oracle, N):
    """SURVEY.md 8d config 2 (1024^2) and config 4 (4096^2) with their literal parameters (amplitude 0.41, the shipped
    OceanRenderer value): waves of hundreds of metres, normals nearly horizontal, whitecap saturated -- the relative tolerance
    must hold there too.  This is the sea bench.py times (workloads.fftmesh_config2)."""
    p = workloads.fftmesh_config2(N)
    h0, h0c = oracle.generate_spectrum(p, 1)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        g0, g0c = o.get_spectrum()
        sc = np.abs(h0).max()
        assert np.abs(g0 - h0).max() < 4e-6 * sc and np.abs(g0c - h0c).max() < 4e-6 * sc
        o.set_spectrum(h0, h0c)
        for t in ((1.0 / 60.0, 1000.0 / 60.0) if N == 1024 else (100.0 / 60.0,)):   # first and last of the config's steps
            v, n, c = o.evaluate(t)
            vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
            assert np.abs(vf[:, 1]).max() > 50.0     # it really is the saturated regime
            workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"config literal {N}^2, t={t}", hds=hds)


@pytest.mark.parametrize("N,u,L,literal", [(64, 1.0, 64.0, False), (256, 0.5, 128.0, False), (512, 1.0, 512.0, False), (1024, 1.0, 1024.0, False),
                                           (2048, 1.0, 2048.0, False), (4096, 1.0, 4096.0, False), (12, 1.0, 12.39, False), (33, 0.9, 33.0, False),
                                           (1024, 1.0, 1024.0, True), (4096, 1.0, 4096.0, True)])
def test_whitecap_stage_bit_exact_on_device(mw, oracle, N, u, L, literal):
    """The Jacobian / whitecap stage (S/FFTMesh.cs:251-276) is index and sign work once hds and the normals exist: forward
    differences, the i = N-1 and j = N-1 edge rules, strict-float32 J, noise, SmoothStep.  With hds taken from the device
    (mw_debug_evaluate_hds) the device's colours must equal the oracle's float32 whitecap of the SAME hds and normals bit
    for bit -- in every kernel family (8 and 4 rows + halo group, sequential halo with virtual threads, direct sum), which
    also proves that a halo row handed to the previous workgroup is the very row its owner computed."""
    amp = 1.5e-8 * (1024.0 / N) ** 2 * u * u * 400.0          # steep enough that the mesh folds here and there
    p = oracle.Params(N=N, unit_width=u, length=L, wind_x=14.45, wind_y=12.0, amplitude=amp if N >= 64 else 0.01, choppiness=1.3)
    if literal:      # SURVEY 8d configs 2 / 4 literally (amplitude 0.41: what bench.py times) -- a saturated whitecap, so the
        p = workloads.fftmesh_config2(N)      # "stage is exercised" assertion below does not apply; bit equality does
    h0, h0c = oracle.generate_spectrum(p, 21)
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        for t in (0.0, 2.5):
            v, n, c, hds = o.debug_evaluate_hds(t)
            want = oracle.whitecap_f32(N, hds, n)
            assert (c == want).all(), f"N={N} t={t}: {(c != want).sum()} colours differ"
            cc = c[:, 0].reshape(N, N)
            assert cc.min() >= 0 and cc.max() <= 1 + 2.0 ** -22    # -2t^3 + 3t^2 left to right in float32 can pass 1 by an ulp (:273)
            if N >= 64 and not literal:
                assert cc.max() > 0.1 and cc.min() < 0.05       # not a saturated or empty field: the stage is exercised
            if literal:
                assert 0.2 < cc.mean() < 0.8 and ((cc == 0) | (cc == 1)).mean() > 0.99      # both saturated values occur, in bulk
            # the hook runs a second instantiation of the same kernel templates (with the hds store compiled in): the same
            # values up to the compiler's choice of fused multiply-adds in the butterflies
            v2, n2, c2 = o.evaluate(t)
            sc = np.abs(hds).max()
            assert np.abs(v - v2).max() < 2e-6 * sc and np.abs(n - n2).max() < 2e-6 and np.abs(c - c2).max() < 2e-5 * max(1.0, sc)
            # hds is what the vertices were displaced by (S/FFTMesh.cs:243-247), to the rounding of rest - d * choppiness
            rest = oracle.rest_mesh(p)[0]
            d = (rest[:, [0, 2]].astype(np.float64) - v[:, [0, 2]]) / p.choppiness
            assert np.abs(d - hds).max() <= 2.0 ** -22 * np.abs(rest).max() / p.choppiness + 1e-7 * np.abs(hds).max()


def test_fftmesh_random_inspector_settings(mw, oracle):
    """16 seeded random parameter sets (size, unit width, wind direction and speed, amplitude, choppiness, gravity, time,
    seed) with the library's OWN spectrum generation on both sides: device spectrum == oracle spectrum, then parity."""
    for p, seed, t in workloads.random_fftmesh_cases(16, seed=2024):
        h0, h0c = oracle.generate_spectrum(p, seed)
        rest = oracle.rest_mesh(p)[0]
        with make(mw, p, seed=seed) as o:
            g0, g0c = o.get_spectrum()
            sc = max(np.abs(h0).max(), 1e-30)
            assert np.abs(g0 - h0).max() < 4e-6 * sc and np.abs(g0c - h0c).max() < 4e-6 * sc, p
            o.set_spectrum(h0, h0c)
            v, n, c = o.evaluate(t)
        vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
        workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"{p} t={t}")


def test_fftmesh_parity_after_an_hour(mw, oracle):
    """timer = 3600 s: omega*t reaches ~2.5e4 rad in float32 (the reference's own product, reproduced bit for bit); the
    kernels' sine/cosine must still be accurate there."""
    p = workloads.fftmesh_params(256)
    h0, h0c = oracle.generate_spectrum(p, 2)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        for t in (3600.0, 86400.0):
            v, n, c = o.evaluate(t)
            vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
            workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"t={t}")


@pytest.mark.parametrize("case", ["calm", "negative_time", "no_chop", "inverted_chop", "moon_gravity", "tiny_patch", "huge_patch"])
def test_fftmesh_edge_parameters(mw, oracle, case):
    """The corners of FFTMesh's Inspector space on the FFT path (256^2 and the headline's 1024^2), device spectrum generation included:
    a calm sea (wind (0, 0): Phillips' l = w^2 / g = 0, exp(-1 / 0) = 0 -- S/FFTMesh.cs:149-166 -- the spectrum and every output are exactly
    flat), negative time (EvaluateWaves accepts any float, :224), choppiness 0 and negative (:244-245), another gravity (the quantised
    dispersion changes its floor, :141-147), a patch of 4 m and of 40 km (k spans 1e-4 .. 1e3 rad/m; length = N * unit_width stays commensurate)."""
    for N in (256, 1024):
        kw = dict(N=N, unit_width=1.0, length=float(N), wind_x=12.0, wind_y=-7.0, amplitude=2e-4, choppiness=0.8, gravity=9.81)
        t = 2.75
        if case == "calm":
            kw.update(wind_x=0.0, wind_y=0.0)
        elif case == "negative_time":
            t = -123.456
        elif case == "no_chop":
            kw.update(choppiness=0.0)
        elif case == "inverted_chop":
            kw.update(choppiness=-1.3)
        elif case == "moon_gravity":
            kw.update(gravity=1.62)
        elif case == "tiny_patch":
            kw.update(unit_width=4.0 / N, length=4.0, amplitude=2e-9)
        elif case == "huge_patch":
            kw.update(unit_width=40000.0 / N, length=40000.0, amplitude=2e-2, wind_x=30.0, wind_y=5.0)
        p = oracle.Params(**kw)
        assert p.commensurate
        h0, h0c = oracle.generate_spectrum(p, 17)
        rest = oracle.rest_mesh(p)[0]
        with make(mw, p, seed=17) as o:
            assert o.max_batch > 1, "FFT path expected"
            g0, g0c = o.get_spectrum()
            sc = max(float(np.abs(h0).max()), 1e-30)
            assert np.isfinite(g0).all() and np.isfinite(g0c).all()
            assert np.abs(g0 - h0).max() <= 4e-6 * sc and np.abs(g0c - h0c).max() <= 4e-6 * sc, (case, N)
            o.set_spectrum(h0, h0c)
            v, n, c = o.evaluate(t)
        assert np.isfinite(v).all() and np.isfinite(n).all() and np.isfinite(c).all(), (case, N)
        if case == "calm":
            assert (h0 == 0).all() and (v == rest).all() and (n == np.array([0, 1, 0], np.float32)).all() and (c == 0).all()
            continue
        vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, np.float32(t), return_hds=True)
        workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"{case} N={N}", hds=hds, min_decided=0.5)
        assert np.abs(v - rest).max() > 0, "a non-trivial sea"


@pytest.mark.parametrize("N", [2048, 4096])
def test_fftmesh_large_grids(mw, oracle, N):
    """BASELINE config 4 (4096^2): parity at full size against the numpy-FFT f64 oracle."""
    p = workloads.fftmesh_params(N)
    h0, h0c = oracle.generate_spectrum(p, 3)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(2.5)
    vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, 2.5, return_hds=True)
    workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"N={N}")


def test_omega_t_bit_exact_on_device(mw, oracle, emul):
    # the quantised dispersion floor() and the omega*t product are "index work": bit for bit over the WHOLE grid against the ORACLE's
    # restatement of S/FFTMesh.cs:141-147,183 (orc_dispersion_grid) -- not only against the host build of the kernels' own source
    # (emul.omega_t), which a shared transcription error in omega_f32 would pass.  Also at the bench's literal
    # config-2 / config-4 parameters (length = N) and with a gravity / length that is not a power of two.
    cases = [(workloads.fftmesh_params(N), t) for N, t in [(64, 1.0), (128, 0.37), (256, 16.65), (512, 3600.0), (1024, 7.3), (2048, 1.0 / 60.0), (4096, 2.5)]]
    cases += [(workloads.fftmesh_config2(1024), 1.0), (workloads.fftmesh_config2(4096), 20.0 / 60.0)]
    cases += [(oracle.Params(N=256, unit_width=1.7, length=435.2, wind_x=3, wind_y=4, amplitude=0.01, gravity=3.711), 11.5)]
    for p, t in cases:
        N = p.N
        with make(mw, p) as o:
            got = o.debug_omega_t(t)
        want = oracle.dispersion_grid(p, t)
        assert (got == want).all(), f"N={N}: {(got != want).sum()} of {N * N} omega*t values differ from the oracle"
        assert (emul.omega_t(p, t) == want).all()      # the emulation's copy is pinned by the same table
        for i in (0, N // 2, N - 1):                    # and the grid form by the scalar entry point the anchor tests use
            row = np.array([oracle.dispersion(p, i, j) for j in (0, 1, N // 2, N - 1)], np.float32) * np.float32(t)
            assert (want[i, [0, 1, N // 2, N - 1]] == row).all()


def test_wave_transpose4_instructions_match_the_index_map(mw, emul):
    """The in-wave exchange of the 1024-point transform (LastInWave): one wave runs v_permlane16_swap / v_permlane32_swap over 64 lanes
    x 16 complex slots of distinct values; every value must land where the index map says (the map the host emulation applies, which
    tests/test_emul.py proves equal to the LDS exchange bit for bit).  Index work: exact."""
    import ctypes as C
    a = np.arange(64 * 16 * 2, dtype=np.float32).reshape(64, 16, 2) + 0.5
    got = a.copy()
    mw.check(mw.lib().mw_debug_wave_transpose4(got.ctypes.data_as(C.c_void_p)))
    want = np.empty_like(a)
    for lane in range(64):
        for rho in range(16):
            sl, sr = C.c_int(), C.c_int()
            emul.L.emul_wave_transpose4_source(lane, rho, C.byref(sl), C.byref(sr))
            want[lane, rho] = a[sl.value, sr.value]
    assert (got == want).all(), f"{(got != want).any(axis=2).sum()} of 1024 slots differ"
    assert not (got == a).all()


def test_rest_mesh_bit_exact_on_device(mw, oracle):
    for N, u, L in [(64, 1.0, 64.0), (12, 1.0, 12.39), (7, 0.37, 3.0)]:
        p = oracle.Params(N=N, unit_width=u, length=L, amplitude=0.01, wind_x=5, wind_y=3)
        v, n, uv, idx = oracle.rest_mesh(p)
        with make(mw, p) as o:
            gv, gn, guv, gidx = o.rest_mesh()
        assert (gv == v).all() and (gn == n).all() and (guv == uv).all() and (gidx == idx).all()


def test_spectrum_generation_on_device(mw, oracle):
    p = workloads.fftmesh_params(256)
    h0, h0c = oracle.generate_spectrum(p, 9)
    with make(mw, p, seed=9) as o:
        g0, g0c = o.get_spectrum()
    sc = np.abs(h0).max()
    assert np.abs(g0 - h0).max() < 4e-6 * sc and np.abs(g0c - h0c).max() < 4e-6 * sc
    assert (g0[128, 128] == 0).all()  # k = 0 bin, S/FFTMesh.cs:153
    # round trip of an injected spectrum is exact
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        r0, r0c = o.get_spectrum()
    assert (r0 == h0).all() and (r0c == h0c).all()


def test_single_bins_and_nyquist_lines(mw, oracle):
    N = 64
    p = workloads.fftmesh_params(N, choppiness=1.0)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        for (i, j, conj) in [(0, 0, False), (0, 5, True), (7, 0, False), (63, 63, True), (32, 32, False), (33, 31, True)]:
            h0 = np.zeros((N, N, 2), np.float32)
            h0c = np.zeros((N, N, 2), np.float32)
            (h0c if conj else h0)[i, j] = (0.3, -0.2)
            o.set_spectrum(h0, h0c)
            v, n, c = o.evaluate(0.8)
            vf, nf, cf = oracle.eval_fft_f64(p, h0, h0c, 0.8)
            workloads.assert_parity(v, n, c, vf, nf, cf, rest, tag=str((i, j, conj)))


@pytest.mark.parametrize("N", [512, 1024, 2048, 4096])
def test_nyquist_lines_only_large_plans(mw, oracle, N):
    """Only the two self-mirrored lines of the spectrum (row i = 0, column j = 0) and a few single bins are non-zero: the outputs
    then consist of nothing but the terms a random Phillips spectrum makes negligible -- the i = 0 correction, the Nyquist-column
    job and its C terms added in pass 2 (deferred to the point of use where the rows are prefetched, 4096^2), and the rebuilt
    halves of the half-stored height and slope rows (mode 2 at 1024^2 / 2048^2, mode 1 at 4096^2).  One plan per size."""
    p = workloads.fftmesh_params(N, choppiness=1.0)
    rng = np.random.default_rng(N)
    h0 = np.zeros((N, N, 2), np.float32)
    h0c = np.zeros((N, N, 2), np.float32)
    sc = 0.05 / N
    for a in (h0, h0c):
        a[0, :] = rng.standard_normal((N, 2)) * sc
        a[:, 0] = rng.standard_normal((N, 2)) * sc
        for (i, j) in ((N // 2, N // 2), (N // 2 + 1, N // 2 - 2), (1, N - 1), (N - 1, 1), (N // 2, 3), (5, N // 2)):
            a[i, j] = rng.standard_normal(2) * sc * 8
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(1.75)
    vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, 1.75, return_hds=True)
    assert np.abs(vf - rest).max() > 1e-3      # the lines do produce a sea
    workloads.assert_parity(v, n, c, vf, nf, cf, rest, np.abs(hds).max(), tag=f"nyquist lines N={N}")


def test_zero_spectrum_flat_and_white_zero(mw, oracle):
    p = workloads.fftmesh_params(128)
    z = np.zeros((128, 128, 2), np.float32)
    with make(mw, p) as o:
        o.set_spectrum(z, z)
        v, n, c = o.evaluate(4.0)
    assert (v == oracle.rest_mesh(p)[0]).all() and (n == [0, 1, 0]).all() and (c == 0).all()


def test_linearity_and_time_reversal_properties_1024(mw, oracle):
    """Size-independent properties at BASELINE's full 1024^2: (i) the displacement/height fields are linear in
    (h0, h0conj); (ii) h~(k,-t) with h0 <-> h0conj swapped equals h~(k,t) (S/FFTMesh.cs:188), so outputs agree."""
    p = workloads.fftmesh_params(1024)
    rest = oracle.rest_mesh(p)[0]
    a0, a0c = oracle.generate_spectrum(p, 11)
    b0, b0c = oracle.generate_spectrum(p, 12)
    with make(mw, p) as o:
        o.set_spectrum(a0, a0c); va, _, _ = o.evaluate(3.0)
        o.set_spectrum(b0, b0c); vb, _, _ = o.evaluate(3.0)
        o.set_spectrum(a0 + b0, a0c + b0c); vs, _, _ = o.evaluate(3.0)
        lin = (vs - rest) - ((va - rest) + (vb - rest))
        assert np.abs(lin).max() < 2e-5 * np.abs(vs - rest).max() + 2.0 ** -21 * np.abs(rest).max()
        o.set_spectrum(a0c, a0); vr, nr, cr = o.evaluate(-3.0)
        o.set_spectrum(a0, a0c); v1, n1, c1 = o.evaluate(3.0)
        assert np.abs(vr - v1).max() < 1e-5 * np.abs(v1 - rest).max() + 2.0 ** -22 * np.abs(rest).max()
        assert np.abs(nr - n1).max() < 1e-5


def test_batched_device_steps_equal_single_steps(mw, oracle):
    """mw_ocean_evaluate_device with nsteps > 1 == the same steps one at a time, bit for bit."""
    import torch
    p = workloads.fftmesh_params(256)
    h0, h0c = oracle.generate_spectrum(p, 1)
    NN = 256 * 256
    times = [0.5 + k / 60.0 for k in range(5)]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        dv = torch.empty((5, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((5, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((5, NN), dtype=torch.float32, device="cuda")
        o.evaluate_device(times, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        o.synchronize()
        for k, t in enumerate(times):
            v, n, c = o.evaluate(t)
            assert (dv[k].cpu().numpy() == v).all() and (dn[k].cpu().numpy() == n).all()
            assert (dw[k].cpu().numpy() == c[:, 0]).all()
        # every pass-1 time-group size (p1_block_map): 24 -> groups of 8, 12 -> 4, 6 -> 2, 7 -> plain 2-D grid
        dv = torch.empty((24, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((24, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((24, NN), dtype=torch.float32, device="cuda")
        singles = {}
        for ns in (24, 12, 6, 7):
            tt = [0.25 + 0.37 * k for k in range(ns)]
            dv.zero_(); dn.zero_(); dw.zero_()
            torch.cuda.synchronize()      # the fills run on torch's stream, the handle enqueues on its own non-blocking stream
            o.evaluate_device(tt, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
            o.synchronize()
            hv, hw = dv.cpu().numpy(), dw.cpu().numpy()
            for k in (0, 1, ns // 2, ns - 1):
                if k not in singles:
                    singles[k] = o.evaluate(tt[k])
                assert (hv[k] == singles[k][0]).all() and (hw[k] == singles[k][2][:, 0]).all(), (ns, k)


@pytest.mark.parametrize("N", [1024, 2048])
def test_every_kind_of_batch_size_equals_single_steps(mw, oracle, N):
    """Enqueues of 2, 3, 5, 16, 31 and 32 steps (prime sizes take the plain 2-D grid of pass 1, the others time groups of 2 .. 8) at the
    headline's grid -- where a single step runs the frame plan's kernels and a batch the sequential-halo kernel with the in-wave exchange --
    and at 2048^2: first and last step of every batch == the same time evaluated alone, bit for bit."""
    import torch
    p = workloads.fftmesh_params(N)
    NN = N * N
    with make(mw, p, seed=6) as o:
        dv = torch.empty((32, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((32, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((32, NN), dtype=torch.float32, device="cuda")
        singles = {}
        for ns in (2, 3, 5, 16, 31, 32):
            tt = [0.125 + 0.61 * k for k in range(ns)]
            dv.zero_(); dn.zero_(); dw.zero_()
            torch.cuda.synchronize()      # the fills run on torch's stream, the handle enqueues on its own non-blocking stream
            o.evaluate_device(tt, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
            o.synchronize()
            for k in (0, ns - 1):
                if k not in singles:
                    singles[k] = o.evaluate(tt[k])
                v, n, c = singles[k]
                assert (dv[k].cpu().numpy() == v).all() and (dn[k].cpu().numpy() == n).all() and (dw[k].cpu().numpy() == c[:, 0]).all(), (ns, k)
        assert np.abs(singles[0][2]).max() > 0
    del dv, dn, dw
    torch.cuda.empty_cache()


@pytest.mark.parametrize("N,K", [(1024, 20), (4096, 32)])
def test_batched_enqueue_at_the_timed_shapes_vs_oracle(mw, oracle, N, K):
    """The shapes bench.py times -- 1024^2 in ONE 20-step enqueue (pass-1 time group 5), 4096^2 in 32-step enqueues (group 8) -- on the
    literal config-2 / config-4 sea: steps 0, mid and last of the batch against the f64 oracle (eval_fft_f64) at the stated tolerance,
    and against the same step evaluated alone, bit for bit (only step 0 went through the bench's gate)."""
    import torch
    p = workloads.fftmesh_config2(N)
    NN = N * N
    times = [(k + 1) / 60.0 for k in range(K)]
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p, seed=1) as o:
        h0, h0c = o.get_spectrum()
        dv = torch.empty((K, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((K, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((K, NN), dtype=torch.float32, device="cuda")
        o.evaluate_device(times, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        o.synchronize()
        for k in (0, K // 2, K - 1):
            v, n, w = dv[k].cpu().numpy(), dn[k].cpu().numpy(), dw[k].cpu().numpy()
            vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, np.float32(times[k]), return_hds=True)
            workloads.assert_parity(v, n, w[:, None], vf, nf, cf[:, :1], rest, np.abs(hds).max(), tag=f"N={N} step {k} of {K}", hds=hds)
            del vf, nf, cf, hds
            v1, n1, c1 = o.evaluate(times[k])
            assert (v == v1).all() and (n == n1).all() and (w == c1[:, 0]).all(), (N, k)
        del dv, dn, dw
    torch.cuda.empty_cache()


def test_update_lifecycle_matches_fftmesh_update(mw, oracle):
    """FFTMesh.Update: timer += deltaTime / tDivision (S/FFTMesh.cs:70), choppiness read live (:244)."""
    m = mw.FFTMesh()
    m.resolution, m.unitWidth, m.length, m.amplitude, m.tDivision = 64, 1.0, 64.0, 2e-6, 2.0
    m.wind = mw.Vector2(14.45, 12.0)
    m.Awake()
    p = oracle.Params(N=64, unit_width=1.0, length=64.0, wind_x=14.45, wind_y=12.0, amplitude=2e-6, choppiness=1.0)
    rv, rn, ruv, ridx = oracle.rest_mesh(p)
    assert (m.mesh.vertices == rv).all() and (m.mesh.indices == ridx).all() and (m.mesh.uv == ruv).all()
    h0, h0c = m.ocean.get_spectrum()
    for _ in range(3):
        m.Update(0.25)
    assert m.timer == pytest.approx(0.375)
    m.choppiness = 0.5
    m.Update(0.25)
    p.choppiness = 0.5
    vf, nf, cf = oracle.eval_fft_f64(p, h0, h0c, 0.5)
    workloads.assert_parity(m.mesh.vertices, m.mesh.normals, m.mesh.colors, vf, nf, cf, rv, tag="Update")
    m.generate = True
    m.Update(0.1)
    assert m.timer == pytest.approx(0.05)


def test_shipped_non_commensurate_scene_direct_path(mw, oracle):
    """The scene the reference ships (N=12, L=12.39, u=1) is not FFT-expressible: served by the direct-sum kernels."""
    p = workloads.shipped_fftmesh_scene()
    h0, h0c = oracle.generate_spectrum(p, 7)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        for t in (0.0, 2.0):
            v, n, c = o.evaluate(t)
            vd, nd, cd = oracle.eval_f64(p, h0, h0c, t)
            workloads.assert_parity(v, n, c, vd, nd, cd, rest, rel=4e-5, tag=f"shipped t={t}")
    # and the golden fixture of the literal f32 restatement (tests/golden/make_golden.py)
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "fftmesh_shipped_n12_t2.npz"))
    with make(mw, p) as o:
        o.set_spectrum(z["h0"], z["h0c"])
        v, n, c = o.evaluate(float(z["t"]))
    assert np.abs(v - z["vertices"]).max() < 2e-5 and np.abs(n - z["normals"]).max() < 2e-5
    assert np.abs(c - z["colors"]).max() < 2e-4


@pytest.mark.parametrize("N", [16, 33, 100])
def test_direct_path_other_grids(mw, oracle, N):
    p = oracle.Params(N=N, unit_width=0.9, length=float(N), wind_x=5, wind_y=3, amplitude=1e-3, choppiness=0.8)
    h0, h0c = oracle.generate_spectrum(p, 2)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(1.25)
    vd, nd, cd = oracle.eval_f64(p, h0, h0c, 1.25)
    workloads.assert_parity(v, n, c, vd, nd, cd, rest, rel=4e-5, tag=f"direct N={N}")


@pytest.mark.parametrize("N,u,L", [(2, 1.0, 2.0), (3, 1.0, 3.0), (5, 0.7, 4.0), (63, 1.0, 63.0), (64, 1.0, 70.0), (127, 1.0, 127.0),
                                   (128, 1.1, 128.0), (129, 1.0, 129.0)])
def test_direct_path_tiny_and_threshold_grids(mw, oracle, N, u, L):
    """Non-FFT grids at the edges of the chirp-z plans: the smallest meshes a scene can carry (2, 3, 5 vertices per side: transform size 64,
    most of it zero padding), power-of-two sizes that are NOT commensurate (64 with length 70, 128 with unit width 1.1: the FFT path must
    refuse them and the separable sum serve them), and the sizes either side of the two-launch / three-launch switch (127, 128 -> M = 256;
    129 -> M = 512); 2, 3, 5 run the one-launch plan (k_czt_one).  Against the f64 oracle's separable sum."""
    p = oracle.Params(N=N, unit_width=u, length=L, wind_x=4.0, wind_y=-2.5, amplitude=1e-3, choppiness=0.6)
    h0, h0c = oracle.generate_spectrum(p, 4)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        assert o.max_batch == 1, "the direct path (one step per enqueue) was expected"
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(0.75)
        kinds = [k for k, _ in o.profile_kernels(nsteps=1, iters=2)]
    # one launch up to N = 20 (k_czt_one), two up to M = 256, three beyond
    assert ("k_czt_one" in kinds[1]) == (N <= 20) and ("rows_assemble" in kinds[1]) == (20 < N <= 128), kinds
    vd, nd, cd, hds = oracle.eval_matmul_f64(p, h0, h0c, 0.75, return_hds=True)
    workloads.assert_parity(v, n, c, vd, nd, cd, rest, rel=2e-5, tag=f"direct N={N}", hds=hds, min_decided=0.0)


def test_direct_path_inspector_defaults(mw, oracle):
    """S/FFTMesh.cs:13-19: a fresh FFTMesh component has resolution 50, length 1, unitWidth 1, wind (1, 1), amplitude 1 -- not
    commensurate (50 * 1 != 1): the MFMA direct-sum path.  mw_params_default carries exactly these values."""
    import ctypes as C
    from mistral_water import _native
    d = _native.MwParams()
    mw.lib().mw_params_default(C.byref(d), mw.MW_SEM_FFTMESH)
    p = oracle.Params(N=d.resolution, unit_width=d.unit_width, length=d.length, wind_x=d.wind_x, wind_y=d.wind_y, amplitude=d.amplitude,
                      choppiness=d.choppiness, gravity=d.gravity)
    assert (p.N, p.length, p.unit_width) == (50, 1.0, 1.0)
    h0, h0c = oracle.generate_spectrum(p, 1)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        assert o.max_batch == 1                                   # direct path: one step per enqueue
        g0, g0c = o.get_spectrum()
        sc = np.abs(h0).max()
        assert np.abs(g0 - h0).max() < 4e-6 * sc
        o.set_spectrum(h0, h0c)
        # On this grid (k up to 157 rad/m on positions up to 25 m) a phase k.x reaches ~3900 rad: the float32 k of the reference
        # (S/FFTMesh.cs:201, PI = 3.1415926536f) and its float32 dot product (:206) are off by ~2^-24 * 3900 = 2e-4 rad per term.
        # Measured distances (float32 restatement of csrc/direct_kernels.h, relative to max |displacement|): device to the f64
        # oracle (exact pi, f64 k) 1.4-3.3e-5, device to the literal float32 loop 4-9e-5, literal to f64 4-9e-5.  Stated
        # tolerance: 2e-4 against the f64 oracle, 3e-4 against the literal loop (as test_fftmesh_1024_vs_literal_f32_sample).
        for t in (0.0, 1.0 / 60.0, 7.5):
            v, n, c = o.evaluate(t)
            vd, nd, cd, hds = oracle.eval_matmul_f64(p, h0, h0c, t, return_hds=True)
            workloads.assert_parity(v, n, c, vd, nd, cd, rest, rel=2e-4, tag=f"Inspector defaults t={t}", hds=hds)
            vl, nl, cl = oracle.eval_literal_f32(p, h0, h0c, t)
            sc = max(float(np.abs(vd - rest).max()), 1e-30)
            assert np.abs(v - vl).max() < 3e-4 * sc and np.abs(n - nl).max() < 3e-4, (t, np.abs(v - vl).max() / sc, np.abs(n - nl).max())


@pytest.mark.parametrize("N,u,L", [(1000, 1.0, 1000.0), (200, 1.0, 212.5), (65, 0.5, 40.0)])
def test_direct_path_large_and_odd_grids(mw, oracle, N, u, L):
    """Grids the FFT cannot express, through the GEMM form of the separable sum (zero-padded to multiples of 64): N = 1000 (not
    a power of two), a non-commensurate even grid, an odd grid one past a tile edge.  Checker: oracle.eval_matmul_f64."""
    p = oracle.Params(N=N, unit_width=u, length=L, wind_x=14.45, wind_y=12.0, amplitude=1.5e-8 * (1024.0 / N) ** 2 * (L / N) ** 2, choppiness=0.46)
    h0, h0c = oracle.generate_spectrum(p, 4)
    rest = oracle.rest_mesh(p)[0]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        for t in (0.5, 16.0):
            v, n, c = o.evaluate(t)
            vd, nd, cd, hds = oracle.eval_matmul_f64(p, h0, h0c, t, return_hds=True)
            # a float32 sum of 2N terms per output (an fmaf chain in k order on the matrix cores): ~sqrt(2N) * 2^-24 relative
            workloads.assert_parity(v, n, c, vd, nd, cd, rest, rel=2e-5, tag=f"direct N={N} t={t}", hds=hds)
        v2, n2, c2 = o.evaluate(0.5)                              # the fixed tables are reused, the result is reproducible
        v1, n1, c1 = o.evaluate(0.5)
        assert (v1 == v2).all() and (c1 == c2).all()


def test_config1_256_literal_sample_fixture(mw, oracle):
    """BASELINE configs[0] on the GPU: the 256 x 256 configuration of the reference's CPU path against the
    committed sample of the literal float32 O(N^4) loop and its f64 values (tests/golden/fftmesh_config1_256_literal_sample.npz)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fftmesh_config1_256_literal_sample.npz"))
    pr = z["params"]
    p = oracle.Params(N=int(pr[0]), unit_width=pr[1], length=pr[2], wind_x=pr[3], wind_y=pr[4], amplitude=pr[5], choppiness=pr[6], gravity=pr[7])
    h0, h0c = oracle.generate_spectrum(p, int(z["seed"]))
    idx, rest = z["vertex_idx"], z["rest"]
    with make(mw, p) as o:
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(float(z["t"]))
    sc = float(np.abs(z["literal_hd"]).max())
    dx, h, dz = (rest[:, 0] - v[idx, 0]) / p.choppiness, v[idx, 1], (rest[:, 2] - v[idx, 2]) / p.choppiness   # S/FFTMesh.cs:243-247
    for got, lit, f64 in ((dx, z["literal_hd"][:, 0], z["f64_disp_x"]), (h, z["literal_hd"][:, 1], z["f64_height"]), (dz, z["literal_hd"][:, 2], z["f64_disp_z"])):
        assert np.abs(got - f64).max() <= 4e-6 * sc + 2.0 ** -22 * np.abs(rest).max()      # the stated float32 tolerance (+ the stored coordinate's ulp)
        assert np.abs(got - lit).max() <= 3e-4 * sc                                        # the literal float32 sum is itself ~1e-5 off here
    assert np.abs(n[idx] - z["f64_normals"]).max() < 4e-6 * max(1.0, float(np.abs(z["f64_normals"][:, [0, 2]] / z["f64_normals"][:, [1]]).max()))


def test_golden_fixture_n16(mw):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fftmesh_n16_t1p5.npz"))
    pr = z["params"]
    with mw.Ocean(resolution=int(pr[0]), unit_width=pr[1], length=pr[2], wind=(pr[3], pr[4]), amplitude=pr[5],
                  choppiness=pr[6], gravity=pr[7]) as o:
        o.set_spectrum(z["h0"], z["h0c"])
        v, n, c = o.evaluate(float(z["t"]))
    sc = np.abs(z["vertices_f64"][:, 1]).max()
    assert np.abs(v - z["vertices_f64"]).max() < 1e-5 * max(sc, 1) + 1e-6
    assert np.abs(n - z["normals_f64"]).max() < 1e-5


def test_golden_pond_and_renderer_fixtures(mw):
    """The committed fixtures of the pond modes and of two OceanRenderer frames + mesh vertex stage, through the C ABI."""
    import os
    G = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(G, "pond_modes_t3p25.npz"))
    M = workloads.POND_MATERIAL
    for tag in ("wave", "gerstner", "level_one"):
        mode, smoothing, amp = z[tag + "_mode_smoothing_amplitude"]
        mat = mw.PondMaterial(mode=int(mode), **{**M, "_Amplitude": float(amp), "_Smoothing": float(smoothing)})
        v, n = mat.displace(z["pos"], float(z["t"]))
        assert np.abs(v - z[tag + "_vertices"]).max() < 6e-6 and np.abs(n - z[tag + "_normals"]).max() < 6e-6, tag
    z = np.load(os.path.join(G, "renderer_res8_frame2.npz"))
    pr = z["params"]
    with mw.Ocean(resolution=int(pr[0]), unit_width=float(z["unit_width"]), length=pr[1], wind=(pr[2], pr[3]), amplitude=pr[4],
                  choppiness=pr[5], gravity=pr[6], mult=pr[7], seed=5, semantics=mw.MW_SEM_OCEANRENDERER) as o:
        o.set_spectrum(z["init4"][..., :2], z["init4"][..., 2:])
        for dt in z["dts"]:
            H, D, Nn, W = o.generate_texture_rgba(float(dt))
        v, n, c = o.displace_mesh()
    for got, key in ((H, "height_rgba"), (D, "disp_rgba")):
        assert np.abs(got - z[key]).max() < 3e-6 * np.abs(z[key]).max(), key
    import or_bounds
    from oracle import oracle as O
    rpz = O.RendererParams(resolution=int(pr[0]), length=float(pr[1]), wind_x=float(pr[2]), wind_y=float(pr[3]), amplitude=float(pr[4]),
                           choppiness=float(pr[5]), gravity=float(pr[6]), mult=float(pr[7]))
    or_bounds.assert_normal_white_stage(O, rpz, H, D, Nn, W, tag="golden renderer frame")
    Dz, Hz = z["disp_rgba"].astype(np.float64), z["height_rgba"].astype(np.float64)
    _, bn, bw, delta = or_bounds.assert_normal_white(Nn[..., :3], W[..., 0], z["normal_rgba"][..., :3].astype(np.float64),
                                                     z["white_rgba"][..., 0].astype(np.float64), float(pr[1]), Dz[..., 0], Dz[..., 1], Dz[..., 2],
                                                     Hz[..., 0], got=(D[..., 0], D[..., 1], D[..., 2], H[..., 0]),
                                                     tag="golden renderer frame", return_bounds=True)
    # the vertex stage, every vertex against its own bound: alone (oracle's f64 stage on the device's textures), then end to end
    # against the committed mesh with the bounds of the texels each vertex samples
    res, uw = int(pr[0]), float(z["unit_width"]) if "unit_width" in z else 1.0
    stage_ref = O.renderer_mesh_vertex_stage_f64(rpz, uw, H[..., 0], D[..., [0, 2]], Nn[..., :3], W[..., 0])
    mag = np.maximum(np.abs(H[..., 0]), np.abs(D[..., [0, 2]]).max(-1))
    or_bounds.assert_mesh_stage_alone(stage_ref, (v, n, c), Nn[..., :3], res, mag, tag="golden renderer frame")
    or_bounds.assert_mesh_end_to_end((z["mesh_vertices"].astype(np.float64), z["mesh_normals"].astype(np.float64), z["mesh_colors"].astype(np.float64)),
                                     (v, n, c), z["normal_rgba"][..., :3].astype(np.float64), bn, bw, delta + 2.0 ** -22 * np.abs(Hz).max(), res, mag,
                                     tag="golden renderer frame")


def test_gerstner_pond(mw, oracle):
    """BASELINE config 5 (1M vertices, 8 waves) and ragged sizes, vs the f64 oracle (oracle/gerstner_oracle.c)."""
    rng = np.random.default_rng(0)
    W, P = workloads.pond_waves8(), workloads.POND
    for nv in (1000 * 1000, 1000003, 4096, 5, 1):
        pos = rng.uniform(-50, 50, (nv, 3)).astype(np.float32)
        out = mw.gerstner_displace(pos, W, P["amplitude"], P["frequency"], P["steepness"], 3.25)
        want = oracle.gerstner_f64(pos, W, P["amplitude"], P["frequency"], P["steepness"], 3.25)
        # the f32 phase frequency*dot(dir, x) reaches ~400 rad (1 ulp = 3e-5 rad) and the offsets are O(0.1)
        assert np.abs(out - want).max() < 2e-5 + 8e-6, nv
    # the 4 waves the reference actually ships (M/Pond Water Mat.mat:134-136)
    pos = rng.uniform(-5, 5, (1000, 3)).astype(np.float32)
    out = mw.gerstner_displace(pos, P["waves"], P["amplitude"], P["frequency"], P["steepness"], 0.7)
    assert np.abs(out - oracle.gerstner_f64(pos, P["waves"], P["amplitude"], P["frequency"], P["steepness"], 0.7)).max() < 3e-6


def test_gerstner_time_batched(mw, oracle):
    """BASELINE config 5 through the batched entry point: 1M vertices x 8 waves x 32 time values in one launch."""
    import torch
    P, W = workloads.POND, workloads.pond_waves8()
    assert mw.lib().mw_gerstner_max_steps(8) == 32 and mw.lib().mw_gerstner_max_steps(4) == 32
    assert mw.lib().mw_gerstner_max_steps(5) == 0
    for nv, nw in ((1000 * 1000, 8), (1003, 4), (3, 8)):
        pos = workloads.pond_lattice(1000, seed=2)[:nv]
        times = [(k + 1) / 60.0 for k in range(32)] if nv > 5000 else [0.0, 3.25, 600.0]
        dp = torch.from_numpy(pos).cuda()
        do = torch.empty((len(times), nv, 3), dtype=torch.float32, device="cuda")
        mw.gerstner_displace_steps_device(dp.data_ptr(), nv, W[:nw], P["amplitude"], P["frequency"], P["steepness"], times,
                                          do.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = do.cpu().numpy()
        for k in (0, len(times) // 2, len(times) - 1):
            want = oracle.gerstner_f64(pos, W[:nw], P["amplitude"], P["frequency"], P["steepness"], times[k])
            assert np.abs(got[k] - want).max() < 8e-6, (nv, nw, k)
    with pytest.raises(mw.MistralWaterError) as e:     # 33 steps of 8 waves exceed the phase table
        mw.gerstner_displace_steps_device(dp.data_ptr(), 3, W, 0.1, 2.58, 0.99, [0.0] * 33, do.data_ptr())
    assert e.value.status == mw.MW_EINVAL


def test_gerstner_edge_parameters(mw, oracle):
    """Corners of the pond's Gerstner() (W/MistralWaterLib.cginc:71-99): amplitude 0 (the lattice comes back bit for bit: every offset is a
    product with steepness * amplitude = 0), steepness 0 (only the vertical term moves), negative time and an hour of it (the phase
    t * speed reaches 1e4 rad: the hardware sine after an exact reduction must hold), one wave, and the two entry points agreeing with each
    other to the angle-addition rounding."""
    import torch
    rng = np.random.default_rng(3)
    W, P = workloads.pond_waves8(), workloads.POND
    pos = rng.uniform(-50, 50, (4099, 3)).astype(np.float32)
    out = mw.gerstner_displace(pos, W, 0.0, P["frequency"], P["steepness"], 1.25)
    assert (out == pos).all()
    out = mw.gerstner_displace(pos, W, P["amplitude"], P["frequency"], 0.0, 1.25)
    assert (out[:, 0] == pos[:, 0]).all() and (out[:, 2] == pos[:, 2]).all() and np.abs(out[:, 1] - pos[:, 1]).max() > 0
    def bound(wv, t):
        # the shader's phase theta = frequency * dot(dir, x) + t * speed is FLOAT32 (W/MistralWaterLib.cginc:80-84, `half` = f32 on desktop),
        # the oracle's f64: each of theta's roundings (the product t * speed, the sum) moves it by up to half an ulp of |theta|, and an
        # offset moves by at most its amplitude times that, per wave
        wv = np.asarray(wv, np.float64).reshape(-1, 3)
        th = np.abs(P["frequency"] * (wv[:, 0] * 50.0 * 1.0 + np.abs(wv[:, 1]) * 50.0)) + abs(t) * np.abs(wv[:, 2])
        ulp = float(np.spacing(np.float32(th.max())))
        return 2e-5 + 8e-6 + len(wv) * max(P["amplitude"], P["amplitude"] * P["steepness"]) * 1.5 * ulp
    for t in (-7.5, 3600.0, -3600.0):
        for wv in (W, W[:1]):
            got = mw.gerstner_displace(pos, wv, P["amplitude"], P["frequency"], P["steepness"], t)
            want = oracle.gerstner_f64(pos, wv, P["amplitude"], P["frequency"], P["steepness"], np.float32(t))
            assert np.abs(got - want).max() < bound(wv, t), (t, len(wv), np.abs(got - want).max(), bound(wv, t))
    dp = torch.from_numpy(pos).cuda()
    times = [-7.5, 0.0, 3600.0]
    do = torch.empty((len(times), pos.shape[0], 3), dtype=torch.float32, device="cuda")
    mw.gerstner_displace_steps_device(dp.data_ptr(), pos.shape[0], W, P["amplitude"], P["frequency"], P["steepness"], times, do.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k, t in enumerate(times):
        want = oracle.gerstner_f64(pos, W, P["amplitude"], P["frequency"], P["steepness"], np.float32(t))
        assert np.abs(do[k].cpu().numpy() - want).max() < bound(W, t), t      # (the batched form's time part is formed in f64 on the host: tighter in practice)


def test_pond_unaligned_device_views(mw, oracle):
    """Device pointers that are NOT 16-byte aligned (a torch view starting one vertex into a buffer, a 1003-vertex step
    stride): the kernels' float4 path must not be taken; results equal the aligned call bit for bit."""
    import torch
    P, W = workloads.POND, workloads.pond_waves8()
    nv = 1003
    pos = workloads.pond_lattice(40, seed=3)[:nv]
    base = torch.zeros((nv + 1) * 3 + 4, dtype=torch.float32, device="cuda")
    obuf = torch.zeros(3 * nv * 3 + 8, dtype=torch.float32, device="cuda")
    for off in (1, 3):                                            # 4- and 12-byte offsets from a 16-B aligned base
        dp = base[off:off + nv * 3]
        dp.copy_(torch.from_numpy(pos).reshape(-1).cuda())
        do = obuf[off:off + 3 * nv * 3]
        assert dp.data_ptr() % 16 != 0 and do.data_ptr() % 16 != 0
        times = [0.0, 3.25, 600.0]
        mw.gerstner_displace_steps_device(dp.data_ptr(), nv, W, P["amplitude"], P["frequency"], P["steepness"], times, do.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = do.cpu().numpy().reshape(3, nv, 3)
        for k, t in enumerate(times):
            want = oracle.gerstner_f64(pos, W, P["amplitude"], P["frequency"], P["steepness"], t)
            assert np.abs(got[k] - want).max() < 8e-6, (off, k)
        mat = mw.PondMaterial(mode=mw.MW_POND_WAVE, **workloads.POND_MATERIAL)
        nbuf = torch.zeros(nv * 3 + 8, dtype=torch.float32, device="cuda")
        dn = nbuf[off + 2:off + 2 + nv * 3]
        mat.displace_device(dp.data_ptr(), nv, 2.0, do.data_ptr(), dn.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ho, hn = mat.displace(pos, 2.0)
        assert (do[:nv * 3].cpu().numpy().reshape(nv, 3) == ho).all() and (dn.cpu().numpy().reshape(nv, 3) == hn).all()


def test_pond_material_displacement_modes(mw, oracle):
    """W/MistralWaterLib.cginc:154-180 Displacement() in its three modes (Wave with its finite-difference normal, Gerstner,
    GerstnerLevelOne) through mw_pond_displace, vs the f64 oracle (oracle/pond_oracle.c); 1M-vertex and ragged sizes."""
    import torch
    M = workloads.POND_MATERIAL
    common = dict(amplitude=M["_Amplitude"], frequency=M["_Frequency"], speed=M["_Speed"], steepness=M["_Steepness"],
                  wspeed=M["_WSpeed"], dir_ab=M["_WDirectionAB"], dir_cd=M["_WDirectionCD"])
    cases = [(mw.MW_POND_WAVE, 1.0, M["_Amplitude"]), (mw.MW_POND_WAVE, 0.35, M["_Amplitude"]),
             (mw.MW_POND_GERSTNER, 1.0, M["_Amplitude"]), (mw.MW_POND_GERSTNER_LEVEL_ONE, 1.0, 0.1)]
    for mode, smoothing, amp in cases:
        mat = mw.PondMaterial(mode=mode, _Amplitude=amp, _Frequency=M["_Frequency"], _Speed=M["_Speed"],
                              _Steepness=M["_Steepness"], _Smoothing=smoothing, _WSpeed=M["_WSpeed"],
                              _WDirectionAB=M["_WDirectionAB"], _WDirectionCD=M["_WDirectionCD"])
        op = oracle.pond_params(mode, smoothing=smoothing, **{**common, "amplitude": amp})
        for n, t in ((1000, 3.25), (37, 61.7), (1, 0.0)):
            pos = workloads.pond_lattice(n, y=0.25, seed=n)
            if n == 37:
                pos = pos[:-2]        # 1367 vertices: not a multiple of 4 (tail path)
            out, nrm = mat.displace(pos, t)
            want, wn = oracle.pond_displace_f64(op, pos, t)
            assert np.abs(out - want).max() < 6e-6, (mode, n)
            assert np.abs(nrm - wn).max() < 6e-6, (mode, n)
            out2, none = mat.displace(pos, t, normals=False)
            assert none is None and (out2 == out).all()
    # Gerstner mode == the 4-wave call of mw_gerstner_displace on the shipped material, bit for bit or within 1 ulp
    P = workloads.POND
    pos = workloads.pond_lattice(100, seed=1)
    a = mw.gerstner_displace(pos, P["waves"], P["amplitude"], P["frequency"], P["steepness"], 3.25)
    b, _ = mw.PondMaterial(mode=mw.MW_POND_GERSTNER, **M).displace(pos, 3.25)
    assert np.abs(a - b).max() < 4e-6
    # device-pointer form on a torch stream
    dp = torch.from_numpy(pos).cuda()
    do, dn = torch.empty_like(dp), torch.empty_like(dp)
    mat = mw.PondMaterial(mode=mw.MW_POND_WAVE, **M)
    mat.displace_device(dp.data_ptr(), pos.shape[0], 2.0, do.data_ptr(), dn.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ho, hn = mat.displace(pos, 2.0)
    assert (do.cpu().numpy() == ho).all() and (dn.cpu().numpy() == hn).all()
    # errors: unknown mode, empty input
    with pytest.raises(mw.MistralWaterError) as e:
        mw.PondMaterial(mode=7, **M).displace(pos, 0.0)
    assert e.value.status == mw.MW_EINVAL
    out, nrm = mw.PondMaterial(**M).displace(np.zeros((0, 3), np.float32), 0.0)
    assert out.shape == (0, 3)


def test_batch_limits_and_empty_inputs(mw, oracle):
    """Maximum batch (32 steps per enqueue) equals single steps; one more is MW_EINVAL; empty pond input is a no-op."""
    import torch
    p = workloads.fftmesh_params(64)
    NN = 64 * 64
    with make(mw, p) as o:
        assert o.max_batch == 32
        times = [0.1 * k for k in range(32)]
        dv = torch.empty((32, NN, 3), dtype=torch.float32, device="cuda")
        dn = torch.empty((32, NN, 3), dtype=torch.float32, device="cuda")
        dw = torch.empty((32, NN, 4), dtype=torch.float32, device="cuda")
        o.evaluate_device(times, dv.data_ptr(), dn.data_ptr(), dw.data_ptr(), rgba=True)
        o.synchronize()
        for k in (0, 17, 31):
            v, n, c = o.evaluate(times[k])
            assert (dv[k].cpu().numpy() == v).all() and (dw[k].cpu().numpy() == c).all()
        with pytest.raises(mw.MistralWaterError) as e:
            o.evaluate_device([0.0] * 33, dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        assert e.value.status == mw.MW_EINVAL
    out = mw.gerstner_displace(np.zeros((0, 3), np.float32), workloads.POND["waves"], 0.1, 2.58, 0.99, 1.0)
    assert out.shape == (0, 3)
    with pytest.raises(mw.MistralWaterError):
        mw.gerstner_displace(np.zeros((4, 3), np.float32), [(1, 0, 1)] * 17, 0.1, 1.0, 0.5, 0.0)   # > 16 waves


def test_registered_host_arrays(mw, oracle):
    """mw_host_register: results copied into page-locked caller arrays equal the pageable path bit for bit."""
    p = workloads.fftmesh_params(128)
    NN = 128 * 128
    with make(mw, p) as o:
        v0, n0, c0 = o.evaluate(2.5)
        v, n, c = np.empty((NN, 3), np.float32), np.empty((NN, 3), np.float32), np.empty((NN, 4), np.float32)
        for a in (v, n, c):
            mw.host_register(a)
        try:
            o.evaluate_into(2.5, v, n, c)
            assert (v == v0).all() and (n == n0).all() and (c == c0).all()
        finally:
            for a in (v, n, c):
                mw.host_unregister(a)
    assert mw.lib().mw_host_register(None, 16) == mw.MW_EINVAL


def test_errors_on_gpu(mw):
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Ocean(resolution=8192, length=8192.0)
    assert e.value.status == mw.MW_EINVAL
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Ocean(resolution=1, length=1.0)
    assert e.value.status == mw.MW_EINVAL
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Ocean(resolution=64, length=64.0, device=99)
    assert e.value.status == mw.MW_EINVAL
