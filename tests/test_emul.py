"""CPU tier: the kernels' own phase functions (mistral-water_amd/csrc/*.h), stepped in lock-step on the
host by tests/emul, against the oracle.  Verifies the Stockham passes, the Hermitian packing with its
Nyquist-line corrections, the exchange-buffer layout, the halo row and every index map before a GPU is
involved.  (The emulation is test infrastructure; the product has no CPU path.)"""
import numpy as np
import pytest

import workloads


@pytest.mark.parametrize("pts", [8, 16])
@pytest.mark.parametrize("N", [64, 128, 256, 512, 1024, 2048, 4096])
def test_stockham_passes_equal_unnormalised_inverse_dft(emul, N, pts):
    rng = np.random.default_rng(N)
    x = rng.standard_normal((N, 2)).astype(np.float32)
    y = emul.fft1d(x, pts)
    ref = np.fft.ifft(x[:, 0].astype(np.float64) + 1j * x[:, 1]) * N
    err = np.abs((y[:, 0] + 1j * y[:, 1]) - ref).max() / np.abs(ref).max()
    assert err < 4e-7


def test_stockham_impulse_and_linearity(emul):
    N = 1024
    for pos in (0, 1, 17, 1023):
        x = np.zeros((N, 2), np.float32)
        x[pos, 0] = 1.0
        y = emul.fft1d(x, 8)
        k = np.arange(N)
        want = np.exp(2j * np.pi * pos * k / N)
        assert np.abs((y[:, 0] + 1j * y[:, 1]) - want).max() < 3e-7
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((2, N, 2)).astype(np.float32)
    assert np.abs(emul.fft1d(a + b) - (emul.fft1d(a) + emul.fft1d(b))).max() < 2e-4


def test_omega_t_bit_exact(emul, oracle):
    for N, t in [(64, 1.0), (256, 16.65)]:
        p = workloads.fftmesh_params(N)
        got = emul.omega_t(p, t)
        want = np.array([[np.float32(oracle.dispersion(p, i, j)) * np.float32(t) for j in range(N)] for i in range(N)],
                        np.float32)
        assert (got == want).all()
        assert (oracle.dispersion_grid(p, t) == want).all()      # the grid entry point == the scalar one
    # the whole grid at the bench's sizes and literal parameters (the kernels' omega_f32 on the host vs the oracle's restatement)
    for p, t in [(workloads.fftmesh_config2(1024), 1.0), (workloads.fftmesh_params(2048), 3600.0), (workloads.fftmesh_config2(4096), 20.0 / 60.0)]:
        assert (emul.omega_t(p, t) == oracle.dispersion_grid(p, t)).all(), p.N


def test_rest_mesh_bit_exact(emul, oracle):
    for N, u in [(64, 1.0), (12, 1.0), (7, 0.37)]:
        p = oracle.Params(N=N, unit_width=u, length=1.0)
        v, n, uv, idx = oracle.rest_mesh(p)
        ev, en, euv, eidx = emul.rest_mesh(N, u)
        assert (ev == v).all() and (en == n).all() and (euv == uv).all() and (eidx == idx).all()


def test_spectrum_generation_matches_oracle(emul, oracle):
    p = workloads.fftmesh_params(64)
    h0, h0c = oracle.generate_spectrum(p, 9)
    e0, e0c = emul.spectrum(p, 9)
    sc = np.abs(h0).max()
    assert np.abs(e0 - h0).max() < 2e-6 * sc and np.abs(e0c - h0c).max() < 2e-6 * sc
    assert (e0[32, 32] == 0).all()


@pytest.mark.parametrize("pts", [8, 16])
@pytest.mark.parametrize("N", [64, 128, 256, 512])
def test_pipeline_vs_oracle_f64(emul, oracle, N, pts):
    p = workloads.fftmesh_params(N)
    h0, h0c = oracle.generate_spectrum(p, 1)
    times = [0.0, 1.0, 16.65]
    v, n, w = emul.evaluate(p, h0, h0c, times, pts=pts)
    rest = oracle.rest_mesh(p)[0]
    for k, t in enumerate(times):
        vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
        workloads.assert_parity(v[k], n[k], w[k], vf, nf, cf, rest, np.abs(hds).max(), tag=f"N={N} t={t}")


@pytest.mark.parametrize("layout", [0, 1])
def test_lds_exchange_layouts(oracle, layout):
    """MW_LDS_LAYOUT: 0 = one padded layout for every exchange, 1 = bank-exact layouts for the radix-16 plans only,
    2 (default) = also for radix 8 (the OceanRenderer passes and 512^2).  Same transforms, same pipeline results as the default build
    (bit for bit: the layouts only move data), 1-D at every size and the full pipeline at 512^2."""
    import emul_build
    e, e1 = emul_build.Emul(defs=(f"MW_LDS_LAYOUT={layout}",)), emul_build.load()
    rng = np.random.default_rng(layout)
    for N in (64, 256, 512, 1024, 2048, 4096):
        x = rng.standard_normal((N, 2)).astype(np.float32)
        for pts in (8, 16):
            assert (e.fft1d(x, pts=pts) == e1.fft1d(x, pts=pts)).all(), (N, pts)
    p = workloads.fftmesh_params(512)
    h0, h0c = oracle.generate_spectrum(p, 2)
    for pts in (8, 16):
        a, b = e.evaluate(p, h0, h0c, [3.25], pts=pts), e1.evaluate(p, h0, h0c, [3.25], pts=pts)
        assert all((u == v).all() for u, v in zip(a, b)), pts
    from oracle.oracle import RendererParams
    rp = RendererParams(resolution=64, length=434.48 / 2, wind_x=14.45, wind_y=12.0, amplitude=0.41)   # M = 512 texels
    ia, pa = e.or_init(rp, 4)
    ib, pb = e1.or_init(rp, 4)
    ra, rb = e.or_step(rp, ia, pa.copy(), 0.02), e1.or_step(rp, ib, pb.copy(), 0.02)
    assert rp.M == 512 and all((u == v).all() for u, v in zip(ra, rb))


def test_round4_shortcuts_change_no_bit(oracle):
    """Round-4 kernel changes that remove work without touching arithmetic, against builds with them switched off -- bit for bit:
    LastInRegs (N = P^S: the last radix-P pass stays in registers instead of an identity round trip through LDS; 512 = 8^3 here, the
    product's 4096 = 16^3 is covered on the GPU) and KeepT1 (the slope assembly takes the mirrored height-row values from the height
    fetch instead of loading them a second time; sequential-halo kernel)."""
    import emul_build
    base = emul_build.load()
    for defs in (("MW_LAST_IN_REGS=0",), ("MW_KEEP_T1=0",)):
        alt = emul_build.Emul(defs=defs)
        for N, pts, hs in ((512, 8, False), (512, 8, True), (256, 16, True)):
            p = workloads.fftmesh_params(N)
            h0, h0c = oracle.generate_spectrum(p, 11)
            base.set_variant(force_hs=hs); alt.set_variant(force_hs=hs)
            a, b = base.evaluate(p, h0, h0c, [2.75], pts=pts), alt.evaluate(p, h0, h0c, [2.75], pts=pts)
            assert all((u == v).all() for u, v in zip(a, b)), (defs, N, pts, hs)
        base.set_variant(force_hs=False)


def test_last_exchange_inside_the_wave_changes_no_bit(oracle):
    """Round 5, LastInWave (mw_math.h): at 1024 points and 16 per thread the exchange in front of the final radix-4 pass is a 4 x 4
    transposition between the wave's four 16-lane rows and the low two bits of the slot index (v_permlane16_swap / v_permlane32_swap on
    the device, the index map wave_transpose4_source here) + a slot renaming, instead of a round trip through LDS behind two barriers.
    Against a build with it switched off, bit for bit, in the three pass-2 kernels that run it: sequential halo (the shipped batched
    plan at 1024^2), the frame variant, and the halo-group form; the map itself is checked against the instructions on the GPU
    (test_wave_transpose4_instructions_match_the_index_map)."""
    import emul_build
    base = emul_build.load()
    alt = emul_build.Emul(defs=("MW_LAST_IN_WAVE=0",))
    p = workloads.fftmesh_params(1024)
    h0, h0c = oracle.generate_spectrum(p, 5)
    try:
        for kw in (dict(force_hs=True), dict(frame=True), dict()):
            base.set_variant(**kw); alt.set_variant(**kw)
            a = base.evaluate(p, h0, h0c, [1.75], pts=16, white_stride=1)
            b = alt.evaluate(p, h0, h0c, [1.75], pts=16, white_stride=1)
            assert all((u == v).all() for u, v in zip(a, b)), kw
            assert np.abs(a[0]).max() > 0 and np.abs(a[2]).max() > 0
    finally:
        base.set_variant()
    # the map is an involution that only touches lane bits 5:4 and slot bits 1:0
    L = base.L
    import ctypes as C
    for lane in range(64):
        for rho in range(16):
            sl, sr = C.c_int(), C.c_int()
            L.emul_wave_transpose4_source(lane, rho, C.byref(sl), C.byref(sr))
            assert (sl.value & 15) == (lane & 15) and (sr.value >> 2) == (rho >> 2)
            assert (sl.value >> 4) == (rho & 3) and (sr.value & 3) == (lane >> 4)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_slope_field_storage_modes(oracle, mode):
    """MW_SPLIT_SLOPES: the slope field crosses the exchange buffer whole (0), as the kx part G for j <= N/2 with
    T3 = G + kz T1 assembled in pass 2 (1, the 4096^2 plan), or as T3 for j <= N/2 with the mirrored half rebuilt as
    conj(T3 - 2 kz T1) (2, the default).  Every mode against the oracle, including the Nyquist lines; sequential-halo
    kernel (the two-part load) and halo-group kernel bit-identical."""
    import emul_build
    e = emul_build.Emul(defs=(f"MW_SPLIT_SLOPES={mode}", f"MW_SPLIT_SLOPES_4096={mode}"))
    for N, pts in ((128, 16), (256, 8)):
        p = workloads.fftmesh_params(N)
        h0, h0c = oracle.generate_spectrum(p, 5)
        times = [0.0, 7.3]
        v, n, w = e.evaluate(p, h0, h0c, times, pts=pts)
        rest = oracle.rest_mesh(p)[0]
        for k, t in enumerate(times):
            vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
            workloads.assert_parity(v[k], n[k], w[k], vf, nf, cf, rest, np.abs(hds).max(), tag=f"mode={mode} N={N} t={t}")
        try:
            e.set_variant(force_hs=True)
            v1, n1, w1 = e.evaluate(p, h0, h0c, times, pts=pts)
        finally:
            e.set_variant(force_hs=False)
        assert (v == v1).all() and (n == n1).all() and (w == w1).all()
    # Nyquist lines only: the j = 0 column (C3 term) and the i = 0 row through every mode
    p = workloads.fftmesh_params(64)
    h0, h0c = oracle.generate_spectrum(p, 3)
    m0 = np.zeros_like(h0); m0c = np.zeros_like(h0c)
    m0[0, :] = h0[0, :]; m0[:, 0] = h0[:, 0]; m0c[0, :] = h0c[0, :]; m0c[:, 0] = h0c[:, 0]
    v, n, w = e.evaluate(p, m0, m0c, [2.5], pts=16)
    vf, nf, cf, hds = oracle.eval_fft_f64(p, m0, m0c, 2.5, return_hds=True)
    workloads.assert_parity(v[0], n[0], w[0], vf, nf, cf, oracle.rest_mesh(p)[0], max(np.abs(hds).max(), 1e-30), tag=f"mode={mode} nyquist")


@pytest.mark.parametrize("N,pts", [(64, 8), (128, 16), (256, 8)])
def test_sequential_halo_pass2_equals_halo_group_pass2(emul, oracle, N, pts):
    """The N >= 4096 kernel (k_pass2_hs: no halo thread group, halo row transformed after the displacement field, whitecap
    finished in the slope field's final pass) forced onto small grids: bit-identical to the halo-group kernel, and
    within tolerance of the oracle."""
    p = workloads.fftmesh_params(N)
    h0, h0c = oracle.generate_spectrum(p, 11)
    times = [0.4, 9.75]
    try:
        emul.set_variant(force_hs=False)
        v0, n0, w0 = emul.evaluate(p, h0, h0c, times, pts=pts)
        emul.set_variant(force_hs=True)
        v1, n1, w1 = emul.evaluate(p, h0, h0c, times, pts=pts)
    finally:
        emul.set_variant(force_hs=False)
    assert (v0 == v1).all() and (n0 == n1).all() and (w0 == w1).all()


def test_pipeline_nyquist_lines_only(emul, oracle):
    """Spectrum supported ONLY on the Nyquist row i=0 and column j=0 (k index 0 mirrors onto itself with a
    sign flip): exercises the dPQ correction tables in isolation."""
    N = 64
    p = workloads.fftmesh_params(N, choppiness=1.0)
    rng = np.random.default_rng(5)
    h0 = np.zeros((N, N, 2), np.float32)
    h0c = np.zeros((N, N, 2), np.float32)
    h0[0, :, :] = rng.standard_normal((N, 2)) * 0.01
    h0[:, 0, :] = rng.standard_normal((N, 2)) * 0.01
    h0c[0, :, :] = rng.standard_normal((N, 2)) * 0.01
    h0c[:, 0, :] = rng.standard_normal((N, 2)) * 0.01
    v, n, w = emul.evaluate(p, h0, h0c, [2.25])
    vf, nf, cf = oracle.eval_fft_f64(p, h0, h0c, 2.25)
    rest = oracle.rest_mesh(p)[0]
    workloads.assert_parity(v[0], n[0], w[0], vf, nf, cf, rest, tag="nyquist")


def test_pipeline_single_bins_index_maps(emul, oracle):
    """One non-zero bin at a time (corners, Nyquist lines, interior): any wrong index / sign shows up at O(1)."""
    N = 64
    p = workloads.fftmesh_params(N, choppiness=1.0)
    for (i, j, conj) in [(0, 0, False), (0, 5, True), (7, 0, False), (63, 63, True), (32, 32, False), (33, 31, True),
                         (1, 62, False)]:
        h0 = np.zeros((N, N, 2), np.float32)
        h0c = np.zeros((N, N, 2), np.float32)
        (h0c if conj else h0)[i, j] = (0.3, -0.2)
        v, n, w = emul.evaluate(p, h0, h0c, [0.8])
        vf, nf, cf = oracle.eval_fft_f64(p, h0, h0c, 0.8)
        workloads.assert_parity(v[0], n[0], w[0], vf, nf, cf, oracle.rest_mesh(p)[0], tag=str((i, j, conj)))


def test_white_scalar_layout(emul, oracle):
    p = workloads.fftmesh_params(64, choppiness=1.5)
    h0, h0c = oracle.generate_spectrum(p, 2)
    _, _, w4 = emul.evaluate(p, h0, h0c, [1.0], white_stride=4)
    _, _, w1 = emul.evaluate(p, h0, h0c, [1.0], white_stride=1)
    assert (w4[..., 0] == w1[..., 0]).all() and (w4[..., 3] == w1[..., 0]).all()


def test_gerstner_vs_oracle(emul, oracle):
    rng = np.random.default_rng(0)
    pos = rng.uniform(-50, 50, (1000, 3)).astype(np.float32)
    W, P = workloads.pond_waves8(), workloads.POND
    out = emul.gerstner(pos, W, P["amplitude"], P["frequency"], P["steepness"], 3.25)
    want = oracle.gerstner_f64(pos, W, P["amplitude"], P["frequency"], P["steepness"], 3.25)
    assert np.abs(out - want).max() < 3e-5
    # numpy cross-check of the oracle itself (W/MistralWaterLib.cginc:77-88)
    x, y, z = pos.astype(np.float64).T
    ox, oy, oz = x.copy(), y.copy(), z.copy()
    for dx, dy, sp in W:
        th = np.float32(P["frequency"]) * (np.float32(dx) * x + np.float32(dy) * z) + np.float32(3.25) * np.float32(sp)
        ox += np.cos(th) * np.float32(P["steepness"]) * np.float32(P["amplitude"]) * np.float32(dx)
        oz += np.cos(th) * np.float32(P["steepness"]) * np.float32(P["amplitude"]) * np.float32(dy)
        oy += np.float32(P["amplitude"]) * np.sin(th)
    assert np.abs(want - np.stack([ox, oy, oz], 1)).max() < 1e-6


@pytest.mark.parametrize("gx,nsteps,tgroup", [(257, 32, 8), (257, 8, 4), (17, 4, 2), (65, 16, 8), (1025, 2, 2),
                                              (257, 20, 5), (257, 21, 7), (129, 18, 6), (17, 9, 3), (1025, 25, 5)])
def test_pass1_time_group_block_map_is_a_bijection(emul, gx, nsteps, tgroup):
    blocks = emul.p1_block_map(gx, nsteps, tgroup)
    live = [b for b in blocks if b is not None]
    assert sorted(live) == [(jb, s) for jb in range(gx) for s in range(nsteps)]
    assert len(blocks) - len(live) < 8 * tgroup
    # the tgroup steps of one column job sit in consecutive slots of one XCD
    for b, v in enumerate(blocks):
        if v is None:
            continue
        slot, xcd = b // 8, b % 8
        if slot % tgroup:
            prev = blocks[(slot - 1) * 8 + xcd]
            assert prev == (v[0], v[1] - 1)


def _pond_cases(oracle):
    M = workloads.POND_MATERIAL
    common = dict(amplitude=M["_Amplitude"], frequency=M["_Frequency"], speed=M["_Speed"], steepness=M["_Steepness"],
                  wspeed=M["_WSpeed"], dir_ab=M["_WDirectionAB"], dir_cd=M["_WDirectionCD"])
    return [("wave", oracle.pond_params(0, smoothing=1.0, **common)),
            ("wave_smooth", oracle.pond_params(0, smoothing=0.35, **common)),
            ("gerstner", oracle.pond_params(1, smoothing=1.0, **common)),
            ("level_one", oracle.pond_params(2, smoothing=1.0, **{**common, "amplitude": 0.1}))]


def test_pond_displacement_modes_vs_oracle(emul, oracle):
    """W/MistralWaterLib.cginc:154-180 Displacement() -- Wave, Gerstner, GerstnerLevelOne -- host-stepped kernel body vs f64."""
    pos = workloads.pond_lattice(64, y=0.25, seed=3)
    for name, p in _pond_cases(oracle):
        for t in (0.0, 3.25, 61.7):
            out, nrm = emul.pond(p, pos, t)
            want, wn = oracle.pond_displace_f64(p, pos, t)
            assert np.abs(out - want).max() < 6e-6, (name, t)     # coordinates up to 50: 1 ulp = 3.8e-6
            assert np.abs(nrm - wn).max() < 6e-6, (name, t)       # phase rounding at |x f| ~ 130 rad: 1.5e-5 rad
    # the Gerstner mode equals the round's first pond entry point (mw_gerstner_displace) on the shipped 4 waves
    P = workloads.POND
    a = emul.gerstner(pos, P["waves"], P["amplitude"], P["frequency"], P["steepness"], 3.25)
    b, _ = emul.pond(_pond_cases(oracle)[2][1], pos, 3.25)
    assert np.abs(a - b).max() < 4e-6


@pytest.mark.parametrize("nwaves", [4, 8])
def test_gerstner_time_batched_path_vs_oracle(emul, oracle, nwaves):
    """mw_gerstner_displace_steps_device's arithmetic (position part once, time part by angle addition), host-stepped."""
    P = workloads.POND
    W = workloads.pond_waves8()[:nwaves]
    pos = workloads.pond_lattice(40, y=0.1, seed=9)
    times = [0.0, 1.0 / 60, 3.25, 61.7, 600.0]
    out = emul.gerstner_steps(pos, W, P["amplitude"], P["frequency"], P["steepness"], times)
    for k, t in enumerate(times):
        want = oracle.gerstner_f64(pos, W, P["amplitude"], P["frequency"], P["steepness"], t)
        assert np.abs(out[k] - want).max() < 8e-6, (nwaves, t)    # 1 ulp of a coordinate near 50 = 3.8e-6


def test_pipeline_random_inspector_settings(emul, oracle):
    """Seeded random parameter sets through the host-stepped kernels (the GPU tier runs 16 of these on the device)."""
    for p, seed, t in workloads.random_fftmesh_cases(5, seed=7, sizes=(64, 128)):
        h0, h0c = oracle.generate_spectrum(p, seed)
        v, n, w = emul.evaluate(p, h0, h0c, [t])
        vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, t, return_hds=True)
        workloads.assert_parity(v[0], n[0], w[0], vf, nf, cf, oracle.rest_mesh(p)[0], np.abs(hds).max(), tag=f"{p} t={t}")


@pytest.mark.parametrize("N,u,L", [(12, 1.0, 12.39), (33, 0.9, 33.0), (50, 1.0, 1.0), (64, 1.0, 64.0), (65, 0.5, 40.0), (100, 0.9, 100.0),
                                   (200, 1.0, 212.5), (600, 1.0, 600.0)])
def test_chirp_z_form_equals_the_separable_sum(emul, oracle, N, u, L):
    """czt_kernels.h: the reference's basis is bilinear in the two indices on ANY grid, k_i x_a = theta (i - N/2)(a - (N-1)/2), so one
    axis of S/FFTMesh.cs:199-217 is a chirp-modulated convolution: two Stockham transforms of size M >= 2N - 1 (the FFT path's own
    passes) instead of an O(N^2) sum.  The kernel's phase functions stepped on the host, both launches with their transposed
    stores, against the f64 matrix-product form of the oracle: the shipped scene, an odd grid, the Inspector defaults (phases up
    to 3900 rad -- the chirps are reduced in f64), a commensurate grid, one past a power of two, M = 64 ... 2048."""
    p = oracle.Params(N=N, unit_width=u, length=L, wind_x=5.0, wind_y=3.0, amplitude=1.0 if L == 1.0 else 1e-3, choppiness=0.8)
    h0, h0c = oracle.generate_spectrum(p, 2)
    F = oracle.htilde_fields_f64(p, h0, h0c, 1.25)
    got, fin = emul.czt2d(N, u, L, F)
    want = oracle.transform_matmul_f64(p, fin)            # of the float32-rounded spectra the kernels see
    for f in range(5):
        assert np.abs(got[f] - want[f]).max() <= 2e-6 * np.abs(want[f]).max(), (f, np.abs(got[f] - want[f]).max() / np.abs(want[f]).max())
    # the product's form (round 4): five real outputs in THREE Hermitian-packed planes on the index set [0, N]^2 (czt_packed_value):
    # (H + i Dx, Sx + i Sz, Dz + i 0) against the real / imaginary parts of the five f64 sums (S/FFTMesh.cs:211-218)
    S = oracle.transform_matmul_f64(p, oracle.htilde_fields_f64(p, h0, h0c, 1.25))
    pk = emul.czt_packed(p, h0, h0c, 1.25)
    pairs = ((pk[0].real, S[0].real, "H"), (pk[0].imag, S[1].imag, "Dx"), (pk[1].real, S[3].imag, "Sx"), (pk[1].imag, S[4].imag, "Sz"),
             (pk[2].real, S[2].imag, "Dz"))
    for gotr, wantr, nm in pairs:      # each against the scale of ITS plane (fields of one scale share a plane)
        scale = max(np.abs(wantr).max(), np.abs(S[0].real).max() if nm in ("H", "Dx", "Dz") else np.abs(S[3].imag).max())
        assert np.abs(gotr - wantr).max() <= 3e-6 * scale, (nm, np.abs(gotr - wantr).max() / scale)
    assert np.abs(pk[2].imag).max() <= 3e-6 * np.abs(S[0].real).max()
    # round 5: omega(i, j) and the wave numbers come from tables formed once per handle (k_czt_tables) -- every element of the three planes is
    # the same bit pattern as with the dispersion and the wave numbers computed in place (czt_packed_value), at two times
    assert emul.czt_tables_vs_inline(p, h0, h0c, 1.25) == 0 and emul.czt_tables_vs_inline(p, h0, h0c, 3600.5) == 0


@pytest.mark.parametrize("N,pts", [(64, 8), (128, 16), (256, 8), (512, 8), (1024, 0)])
def test_frame_variant_of_pass2_equals_the_shipped_plan(emul, oracle, N, pts):
    """k_pass2_frame (the single-step plan at 1024^2: 3 R2 + 1 row groups transform the three fields of a row block and the halo
    row side by side and meet through LDS -- hds published, noise term parked, vertices stored by the height groups, whitecap by the
    displacement groups) stepped phase by phase on the host, at small grids too: the bit patterns of the plan Plan<N> ships
    (halo-group kernel up to 512^2, sequential-halo kernel at 1024^2 = pts 0, the product's points per thread)."""
    p = workloads.fftmesh_params(N)
    h0, h0c = oracle.generate_spectrum(p, 11)
    times = [0.4, 9.75] if N < 1024 else [3.25]
    try:
        emul.set_variant()
        v0, n0, w0 = emul.evaluate(p, h0, h0c, times, pts=pts)
        emul.set_variant(frame=True)
        v1, n1, w1 = emul.evaluate(p, h0, h0c, times, pts=pts, white_stride=1)
    finally:
        emul.set_variant()
    assert (v0 == v1).all() and (n0 == n1).all() and (w0[..., 0] == w1[..., 0]).all() and np.abs(w0).max() > 0


def test_frame_plan_pass1_job_list(emul):
    """p1_frame_jobs (the 1-D grid of the single-step plan's pass 1): every ACTIVE (column job, field) pair exactly once, nothing
    else; workgroup b runs on XCD b % 8, and the fields of one column job sit in consecutive slots of one XCD."""
    import ctypes as C
    N = 1024
    buf = (C.c_int * 4096)()
    n = emul.L.emul_p1_frame_jobs(N, buf, 4096)
    jobs = list(buf[:n])
    assert n % 8 == 0 and n <= 4096
    seen = {}
    for b, j in enumerate(jobs):
        if j < 0:
            continue
        f, jb = j >> 16, j & 0xffff
        assert emul.L.emul_p1_field_active(N, jb, f) == 1 and (f, jb) not in seen
        seen[(f, jb)] = b
        assert b % 8 == jb % 8
    want = {(f, jb) for jb in range(N // 4 + 1) for f in range(3) if emul.L.emul_p1_field_active(N, jb, f)}
    assert set(seen) == want and len(want) == 516
    for (f, jb), b in seen.items():      # the other active fields of the same column job: the neighbouring slots of the same XCD
        for g in range(3):
            if (g, jb) in seen:
                assert abs(seen[(g, jb)] - b) <= 16
    per_xcd = [sum(1 for b, j in enumerate(jobs) if j >= 0 and b % 8 == x) for x in range(8)]
    assert max(per_xcd) - min(per_xcd) <= 4
