"""CPU tier: pins the oracle (oracle/) as far as the reference allows.

The reference holds no tests or golden vectors, so the oracle is anchored on
 (i) the closed-form anchor table of SURVEY.md 8a, (ii) analytic cases, (iii) agreement between its
three independent evaluators, (iv) the committed fixtures under tests/golden/ (regression pins).
"""
import os

import numpy as np
import pytest

import workloads

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_anchor_table_phillips_dispersion(oracle):
    # SURVEY.md 8a "Known-answer anchors": A Phillips / Dispersion, N=12, L=12.39, wind=(5,3), A=0.01
    p = workloads.shipped_fftmesh_scene()
    ph = {(6, 6): 0.0, (7, 6): 8.043365213e-2, (6, 7): 2.895611477e-2, (8, 9): 7.127415871e-4,
          (0, 0): 2.732263740e-5, (11, 11): 5.654809393e-5}
    for (n, m), want in ph.items():
        got = oracle.phillips(p, n, m)
        assert got == pytest.approx(want, rel=2e-6, abs=1e-12), (n, m)
    om = {(6, 6): 0.0, (7, 6): 2.0284698, (8, 9): 4.0569397, (0, 0): 6.0854095, (11, 11): 5.5782920}
    for (n, m), want in om.items():
        assert oracle.dispersion(p, n, m) == pytest.approx(want, rel=1e-6, abs=1e-9), (n, m)


def test_dispersion_is_quantised_and_even(oracle):
    p = workloads.fftmesh_params(64)
    w0 = np.float32(2) * np.float32(3.1415926536) / np.float32(p.length)
    for (n, m) in [(1, 2), (10, 33), (63, 5), (0, 0), (32, 32)]:
        w = np.float32(oracle.dispersion(p, n, m))
        q = w / w0
        assert abs(q - round(float(q))) < 1e-3
        if n and m:  # omega(k) == omega(-k) bit for bit: the Hermitian packing relies on it
            assert oracle.dispersion(p, 64 - n, 64 - m) == float(w)


def test_rng_range_and_determinism(oracle):
    u = np.array([oracle.uniform(7, c) for c in range(20000)], np.float32)
    assert u.min() > 0.0 and u.max() <= 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    assert oracle.uniform(7, 123) == oracle.uniform(7, 123) != oracle.uniform(8, 123)


def test_spectrum_statistics(oracle):
    # E|h0|^2 = Phillips (two components of variance Phillips/2 each), S/FFTMesh.cs:168-176
    p = workloads.fftmesh_params(64)
    acc = np.zeros((64, 64))
    K = 200
    for seed in range(K):
        h0, _ = oracle.generate_spectrum(p, seed)
        acc += (h0.astype(np.float64) ** 2).sum(-1)
    acc /= K
    ph = np.array([[oracle.phillips(p, i, j) for j in range(64)] for i in range(64)])
    mask = ph > ph.max() * 1e-3
    ratio = acc[mask] / ph[mask]
    assert abs(ratio.mean() - 1.0) < 0.05
    assert acc[32, 32] == 0.0  # k = 0 -> Phillips returns 0 (:153)


def test_conj_draw_is_independent_not_mirrored(oracle):
    # S/FFTMesh.cs:115-116: vertConj is a fresh draw at the mirrored k, NOT conj(verttilde[mirror])
    p = workloads.fftmesh_params(64)
    h0, h0c = oracle.generate_spectrum(p, 3)
    mir = h0[(64 - np.arange(64)) % 64][:, (64 - np.arange(64)) % 64]
    assert not np.allclose(h0c[..., 0], mir[..., 0])


def test_rest_mesh_layout(oracle):
    p = workloads.fftmesh_params(64)
    v, n, uv, idx = oracle.rest_mesh(p)
    assert v[0].tolist() == [-31.5, 0.0, -31.5] and v[64 * 64 - 1].tolist() == [31.5, 0.0, 31.5]
    assert v[1].tolist() == [-31.5, 0.0, -30.5]          # j is the fast (z) axis, :110
    assert (n == [0, 1, 0]).all()
    assert uv[-1].tolist() == [1.0, 1.0]
    assert idx.size == 63 * 63 * 6 and idx.min() == 0 and idx.max() == 64 * 64 - 1
    assert idx[:3].tolist() == [0, 1, 64]                 # first triangle, :122-124
    # odd N has no half-cell offset (:112)
    v2 = oracle.rest_mesh(oracle.Params(N=5, unit_width=2.0, length=10.0))[0]
    assert v2[0].tolist() == [-4.0, 0.0, -4.0]


@pytest.mark.parametrize("N", [16, 32])
def test_literal_f32_vs_f64_vs_fft(oracle, N):
    # the FFT restatement is exact iff unit_width == length/N (second probe)
    p = workloads.fftmesh_params(N, choppiness=1.0)
    h0, h0c = oracle.generate_spectrum(p, 11)
    for t in (0.0, 1.0, 12.75):
        vl, nl, cl = oracle.eval_literal_f32(p, h0, h0c, t)
        vd, nd, cd = oracle.eval_f64(p, h0, h0c, t)
        vf, nf, cf = oracle.eval_fft_f64(p, h0, h0c, t)
        scale = np.abs(vd[:, 1]).max()
        assert np.abs(vf - vd).max() < 1e-11 * max(scale, 1)
        assert np.abs(nf - nd).max() < 1e-12 and np.abs(cf - cd).max() < 1e-11
        assert np.abs(vl - vd).max() < 2e-5 * max(scale, 1)
        assert np.abs(nl - nd).max() < 2e-5 and np.abs(cl - cd).max() < 5e-4


def test_non_commensurate_is_not_fft_expressible(oracle):
    # the SHIPPED scene (N=12, L=12.39, u=1) is outside the FFT precondition
    p = workloads.shipped_fftmesh_scene()
    assert not p.commensurate
    h0, h0c = oracle.generate_spectrum(p, 5)
    vl, nl, cl = oracle.eval_literal_f32(p, h0, h0c, 2.5)
    vd, nd, cd = oracle.eval_f64(p, h0, h0c, 2.5)
    assert np.abs(vl - vd).max() < 1e-4 and np.abs(nl - nd).max() < 1e-5


def test_zero_spectrum_is_flat(oracle):
    p = workloads.fftmesh_params(16)
    z = np.zeros((16, 16, 2), np.float32)
    v, n, c = oracle.eval_literal_f32(p, z, z, 3.0)
    rest = oracle.rest_mesh(p)[0]
    assert (v == rest).all() and (n == [0, 1, 0]).all() and (c == 0).all()  # J = 1 -> colour 0


def test_single_mode_is_a_travelling_cosine(oracle):
    # one non-zero h0 bin -> height = Re(h0 e^{i(w t + k.x)}); phase speed from Dispersion()
    N = 32
    p = oracle.Params(N=N, unit_width=1.0, length=float(N), wind_x=1, wind_y=0, amplitude=1, choppiness=0.0)
    h0 = np.zeros((N, N, 2), np.float32)
    hc = np.zeros((N, N, 2), np.float32)
    i0, j0, amp = 19, 14, 0.25
    h0[i0, j0, 0] = amp
    t = 1.7
    v, n, c = oracle.eval_f64(p, h0, hc, t)
    kx, kz = 2 * np.pi * (i0 - N / 2) / N, 2 * np.pi * (j0 - N / 2) / N
    w = oracle.dispersion(p, i0, j0)
    rest = oracle.rest_mesh(p)[0].astype(np.float64)
    want = amp * np.cos(np.float64(np.float32(w) * np.float32(t)) + kx * rest[:, 0] + kz * rest[:, 2])
    assert np.abs(v[:, 1] - want).max() < 1e-12
    assert np.abs(v[:, 0] - rest[:, 0]).max() == 0  # choppiness 0


def test_whitecap_edge_rows_bit_exact(oracle):
    # S/FFTMesh.cs:260-267: forward differences, zero on the last row / column
    N = 8
    rng = np.random.default_rng(0)
    hds = rng.standard_normal((N * N, 2)).astype(np.float32)
    nor = np.tile(np.array([0, 1, 0], np.float32), (N * N, 1))
    c = oracle.whitecap_f32(N, hds, nor)[:, 0].reshape(N, N)
    # bottom-right corner has both differences zeroed: J = 1 -> turb = 0 -> 0
    assert c[N - 1, N - 1] == 0.0
    # last row uses only the j-difference: J = 1 * (1 + dDdy.y) - 0
    h = hds.reshape(N, N, 2)
    dy = np.float32(0.5) * (h[N - 1, 2, 1] - h[N - 1, 3, 1])
    turb = max(np.float32(1) - (np.float32(1) * (np.float32(1) + dy)), np.float32(0))
    tt = np.float32(min(max(turb, 0), 1))
    want = np.float32(-2.0) * tt * tt * tt + np.float32(3.0) * tt * tt
    assert c[N - 1, 2] == np.float32(want)


@pytest.mark.parametrize("name", ["fftmesh_n16_t1p5", "fftmesh_shipped_n12_t2"])
def test_golden_fixtures(oracle, name):
    """Regression pins generated by tests/golden/make_golden.py from the oracle itself (the reference
    cannot run; see the fixture README).  Literal f32 outputs are compared bit for bit."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    p = oracle.Params(**{k: (int(z["params"][i]) if k == "N" else float(z["params"][i]))
                         for i, k in enumerate(["N", "unit_width", "length", "wind_x", "wind_y", "amplitude",
                                                "choppiness", "gravity"])})
    h0, h0c = oracle.generate_spectrum(p, int(z["seed"]))
    assert (h0 == z["h0"]).all() and (h0c == z["h0c"]).all()
    v, n, c = oracle.eval_literal_f32(p, h0, h0c, float(z["t"]))
    assert (v == z["vertices"]).all() and (n == z["normals"]).all() and (c == z["colors"]).all()
    vd, nd, cd = oracle.eval_f64(p, h0, h0c, float(z["t"]))
    assert np.abs(vd - z["vertices_f64"]).max() < 1e-12


def _golden_pond_cases(z):
    M = workloads.POND_MATERIAL
    common = dict(frequency=M["_Frequency"], speed=M["_Speed"], steepness=M["_Steepness"], wspeed=M["_WSpeed"],
                  dir_ab=M["_WDirectionAB"], dir_cd=M["_WDirectionCD"])
    for tag in ("wave", "gerstner", "level_one"):
        mode, smoothing, amp = z[tag + "_mode_smoothing_amplitude"]
        yield tag, int(mode), float(smoothing), float(amp), common


def test_golden_pond_modes(oracle, emul):
    """tests/golden/pond_modes_t3p25.npz: the oracle re-derives it exactly; the kernel body (host-stepped) matches it."""
    z = np.load(os.path.join(GOLDEN, "pond_modes_t3p25.npz"))
    for tag, mode, smoothing, amp, common in _golden_pond_cases(z):
        p = oracle.pond_params(mode, amplitude=amp, smoothing=smoothing, **common)
        v, n = oracle.pond_displace_f64(p, z["pos"], float(z["t"]))
        assert np.abs(v - z[tag + "_vertices"]).max() < 1e-12 and np.abs(n - z[tag + "_normals"]).max() < 1e-12
        ev, en = emul.pond(p, z["pos"], float(z["t"]))
        assert np.abs(ev - z[tag + "_vertices"]).max() < 6e-6 and np.abs(en - z[tag + "_normals"]).max() < 6e-6


def test_golden_renderer_frame(oracle):
    """tests/golden/renderer_res8_frame2.npz: two GenerateTexture() frames at 64^2 and the mesh vertex stage."""
    z = np.load(os.path.join(GOLDEN, "renderer_res8_frame2.npz"))
    pr = z["params"]
    rp = oracle.RendererParams(resolution=int(pr[0]), length=pr[1], wind_x=pr[2], wind_y=pr[3], amplitude=pr[4],
                               choppiness=pr[5], gravity=pr[6], mult=pr[7])
    init4 = oracle.renderer_initial_spectrum(rp, 5)
    assert (init4 == z["init4"]).all()
    ph = np.zeros((rp.M, rp.M), np.float32)
    for dt in z["dts"]:
        H, D, Nn, W = oracle.renderer_textures_f64(rp, init4, ph, float(dt))
    assert (ph == z["phase"]).all()          # the float32 phase state, bit for bit
    for got, key in ((H, "height_rgba"), (D, "disp_rgba"), (Nn, "normal_rgba"), (W, "white_rgba")):
        assert (got.astype(np.float32) == z[key]).all()
    v, n, c = oracle.renderer_mesh_vertex_stage_f64(rp, float(z["unit_width"]), H[..., 0], D[..., [0, 2]], Nn[..., :3], W[..., 0])
    assert np.abs(v - z["mesh_vertices"]).max() < 1e-12 and np.abs(n - z["mesh_normals"]).max() < 1e-12
    assert np.abs(c - z["mesh_colors"]).max() < 1e-12


@pytest.mark.parametrize("N,threads", [(64, 1), (256, 3), (512, 8)])
def test_cpu_fft_baseline_matches_oracle(oracle, N, threads):
    """oracle/cpu_fft_baseline.c (bench.py's all-cores CPU baseline) against the f64 oracle; thread count is immaterial."""
    p = workloads.fftmesh_params(N)
    h0, h0c = oracle.generate_spectrum(p, 4)
    v, n, c = oracle.cpu_fft_step_f32(p, h0, h0c, 2.5, threads)
    vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, 2.5, return_hds=True)
    rest = oracle.rest_mesh(p)[0]
    sc = max(np.abs(hds).max(), np.abs(vf[:, 1]).max())
    assert np.abs(v - vf).max() < 2e-5 * sc + 2.0 ** -22 * np.abs(rest).max()      # f32 radix-2, log2 N stages
    assert np.abs(n - nf).max() < 2e-5 and np.abs(c - cf).max() < 2e-4
    v1, n1, c1 = oracle.cpu_fft_step_f32(p, h0, h0c, 2.5, 1)
    assert (v1 == v).all() and (n1 == n).all() and (c1 == c).all()
    with pytest.raises(ValueError):
        oracle.cpu_fft_step_f32(workloads.shipped_fftmesh_scene(), *oracle.generate_spectrum(workloads.shipped_fftmesh_scene(), 1), 1.0)


@pytest.mark.parametrize("N,u,L", [(12, 1.0, 12.39), (33, 0.9, 33.0), (50, 1.0, 1.0), (64, 1.0, 64.0)])
def test_matmul_form_equals_the_direct_f64_sum(oracle, N, u, L):
    """oracle.eval_matmul_f64 (BLAS form of S/FFTMesh.cs:199-217 for any grid: the checker of the large non-FFT cases) against
    orc_eval_f64, the statement-by-statement f64 restatement: the shipped scene, an odd grid, the Inspector defaults
    (S/FFTMesh.cs:13-19: resolution 50, length 1, unitWidth 1) and a commensurate grid."""
    p = oracle.Params(N=N, unit_width=u, length=L, wind_x=1.0, wind_y=1.0, amplitude=1.0 if L == 1.0 else 1e-3, choppiness=1.0)
    h0, h0c = oracle.generate_spectrum(p, 9)
    a = oracle.eval_f64(p, h0, h0c, 1.25)
    b = oracle.eval_matmul_f64(p, h0, h0c, 1.25)
    for x, y in zip(a, b):
        assert np.abs(x - y).max() <= 1e-11 * max(1.0, float(np.abs(x).max()))


def test_config1_literal_sample_fixture(oracle):
    """tests/golden/fftmesh_config1_256_literal_sample.npz -- BASELINE configs[0] (256 x 256, the reference's CPU-runnable case):
    the spectrum regenerates from the seed, the literal float32 loop reproduces the committed sample bit for bit, and the f64
    FFT-form evaluator stands where the fixture says (1e-5 of the field: the literal sum of 65536 float32 terms)."""
    z = np.load(os.path.join(GOLDEN, "fftmesh_config1_256_literal_sample.npz"))
    pr = z["params"]
    p = oracle.Params(N=int(pr[0]), unit_width=pr[1], length=pr[2], wind_x=pr[3], wind_y=pr[4], amplitude=pr[5], choppiness=pr[6], gravity=pr[7])
    h0, h0c = oracle.generate_spectrum(p, int(z["seed"]))
    assert np.allclose([h0.astype(np.float64).sum(), np.abs(h0).astype(np.float64).sum()], z["h0_checksum"], rtol=1e-12, atol=0)
    idx = z["vertex_idx"]
    hd, nor = oracle.displacement_subset_f32(p, h0, h0c, float(z["t"]), idx[:24])          # a slice: 24 x 65536 terms
    assert (hd == z["literal_hd"][:24]).all() and (nor == z["literal_normals"][:24]).all()
    vf, nf, cf, hds = oracle.eval_fft_f64(p, h0, h0c, float(z["t"]), return_hds=True)
    assert np.allclose(vf[idx, 1], z["f64_height"], rtol=0, atol=1e-9) and np.allclose(hds[idx, 0], z["f64_disp_x"], rtol=0, atol=1e-9)
    sc = np.abs(z["literal_hd"]).max()
    assert np.abs(z["literal_hd"][:, 1] - z["f64_height"]).max() < 3e-5 * sc
