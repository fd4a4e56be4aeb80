"""OceanRenderer semantics.  CPU tier: oracle self-consistency + the kernels' phase functions
under host emulation; GPU tier (-m gpu): the C ABI (mw_ocean_generate_texture) against the oracle."""
import numpy as np
import pytest

import or_bounds
from oracle.oracle import RendererParams


def shipped(resolution=8):
    # D/Ocean Demo.unity:296-302 (resolution 128 -> 1024^2 textures there; smaller here), length scaled with M
    M = 8 * resolution
    return RendererParams(resolution=resolution, length=434.48 * M / 1024.0, wind_x=14.45, wind_y=12.0, amplitude=0.41,
                          choppiness=0.46, gravity=9.81, mult=1.5)


@pytest.fixture(params=["packed", "three"])
def or_plan(request):
    """The two plans of a planar-texture call: two transforms per frame (height + i Dz share one; csrc/ocean_renderer_kernels.h, "the packed
    plan") -- the default wherever the phase texture is mirror-symmetric -- and the shaders' three (switch MW_OR_PACKED = 0)."""
    import mistral_water
    mistral_water.set_switch("MW_OR_PACKED", 1 if request.param == "packed" else 0)
    yield request.param
    mistral_water.set_switch("MW_OR_PACKED", 1)


def tol_check(got, want, rel, name):
    sc = max(float(np.abs(want).max()), 1e-6)
    err = float(np.abs(got - want).max())
    assert err <= rel * sc, f"{name}: {err:.3e} vs scale {sc:.3e}"


def test_anchor_table_b_phillips(oracle):
    # SURVEY.md 8a anchors: B Phillips phi1, N=1024, L=434.48, wind=(14.45,12), A=0.41e-4 at texel px
    rp = RendererParams(resolution=128, length=434.48, wind_x=14.45, wind_y=12.0, amplitude=0.41)
    L = oracle.lib()
    L.orr_phillips.restype = __import__("ctypes").c_float
    import ctypes as C
    for (px, py), want in {(1, 0): 13.75721, (0, 1): 9.487616, (3, 2): 4.126636, (1023, 1023): 36.58870, (1000, 5): 9.844989e-4}.items():
        got = L.orr_phillips(C.byref(rp.c()), C.c_float(px + 0.5), C.c_float(py + 0.5), 1024)
        assert got == pytest.approx(want, rel=3e-6)


def test_anchor_table_b_dispersion(oracle):
    # B omega (dt = 1, phase from 0): px=(1,0) / (3,2) / (512,512) / (1000,5)
    import ctypes as C
    rp = RendererParams(resolution=128, length=434.48)
    L = oracle.lib()
    L.orr_phase_advance.restype = C.c_float
    for (px, py), want in {(1, 0): 0.3766514, (3, 2): 0.7151965, (1000, 5): 1.8649121}.items():
        got = L.orr_phase_advance(C.byref(rp.c()), 1024, px, py, C.c_float(0.0), C.c_float(1.0))
        assert got == pytest.approx(want, rel=2e-6)
    got = L.orr_phase_advance(C.byref(rp.c()), 1024, 512, 512, C.c_float(0.0), C.c_float(1.0))
    assert got == pytest.approx(10.1392509 % (2 * np.float32(3.1415926536)), rel=2e-6)   # wrapped by fmod


def test_literal_stockham_schedule_is_forward_dft(oracle):
    """The 2*log2(M) gather passes of S/OceanRenderer.cs:229-262 == numpy fft2 (forward, unnormalised, natural order)."""
    rp = shipped(8)
    init4 = oracle.renderer_initial_spectrum(rp, 3)
    pa = np.zeros((rp.M, rp.M), np.float32)
    pb = np.zeros((rp.M, rp.M), np.float32)
    a = oracle.renderer_step_f64(rp, init4, pa, 0.02, literal_passes=True)
    b = oracle.renderer_step_f64(rp, init4, pb, 0.02, literal_passes=False)
    assert (pa == pb).all()
    for x, y, nm in zip(a, b, ("height", "disp", "normal", "white", "disp_g")):
        tol_check(x, y, 1e-10, nm)


def test_initial_spectrum_off_by_one_mirror(oracle):
    # b5: at texel (0,0) phi1 = 0 (k = 0) but phi2 = Phillips at texel (M-1, M-1) != 0
    rp = shipped(8)
    init4 = oracle.renderer_initial_spectrum(rp, 1)
    assert (init4[0, 0, :2] == 0).all() and np.abs(init4[0, 0, 2:]).max() > 0


def test_phase_is_stateful_and_wrapped(oracle):
    rp = shipped(8)
    init4 = oracle.renderer_initial_spectrum(rp, 1)
    ph = np.zeros((rp.M, rp.M), np.float32)
    for _ in range(40):
        oracle.renderer_step_f64(rp, init4, ph, 0.3, literal_passes=False)
    assert ph.min() >= 0 and ph.max() < 2 * np.float32(3.1415926536) and ph[0, 0] == 0  # k = 0 never advances


@pytest.mark.parametrize("resolution", [8, 16, 32])
@pytest.mark.parametrize("packed", [False, True], ids=["three_transforms", "packed"])
def test_emulated_kernels_vs_oracle(emul, oracle, resolution, packed):
    """The kernels' phase functions stepped on the host, both plans of a planar-texture frame: the shaders' three transforms, and the packed
    two (height + i Dz from one transform of the Hermitian parts; csrc/ocean_renderer_kernels.h)."""
    rp = shipped(resolution)
    M = rp.M
    init4 = oracle.renderer_initial_spectrum(rp, 5)
    initT, phaseT = emul.or_init(rp, 5)
    sc = np.abs(init4).max()
    assert np.abs(initT.transpose(1, 0, 2) - init4).max() < 3e-6 * sc   # generation: libm differences only
    initT = np.ascontiguousarray(init4.transpose(1, 0, 2))              # then inject identical spectra
    ph = np.zeros((M, M), np.float32)
    for frame, dt in enumerate((0.016, 0.033, 0.3, 25.0)):     # 25 s: phase steps beyond 4 pi (library fmod path)
        h, d, n, w, g = emul.or_step(rp, initT, phaseT, dt, packed=packed)
        H, D, Nn, W, G = oracle.renderer_step_f64(rp, init4, ph, dt, literal_passes=(M <= 128))
        assert (phaseT.T == ph).all(), "stateful f32 phase must match bit for bit"
        tol_check(h, H, 3e-6, "height"); tol_check(d, D, 3e-6, "disp"); tol_check(g, G, 3e-6, "disp.g")
        assert np.abs(n - Nn).max() < 2e-5
        assert np.abs(w - W).max() < 5e-5


def _rgba_checks(tex, want, rp, oracle):
    (Ht, Dt, Nt, Wt), (HT, DT, NT, WT) = tex, want
    tol_check(Ht, HT, 3e-6, "height rgba"); tol_check(Dt, DT, 3e-6, "disp rgba")
    assert (Ht[..., 0] == Ht[..., 2]).all() and (Ht[..., 1] == Ht[..., 3]).all()     # float4(h, h)
    assert (Nt[..., 3] == 1).all() and (Wt[..., 3] == 1).all()
    assert (Wt[..., 0] == Wt[..., 1]).all() and (Wt[..., 0] == Wt[..., 2]).all()
    # (1) the normal / whitecap passes alone, on the device's own textures: float32 rounding of the pass only, per texel
    rn, rw, mn, mw_ = or_bounds.assert_normal_white_stage(oracle, rp, Ht, Dt, Nt, Wt, tag="rgba")
    M = Ht.shape[0]                                      # the bound is not vacuous: ~1e-3 at the shipped 1024^2 scale, where
    assert mn < 2e-3 * max(1.0, M / 1024.0) ** 2, mn     # float32 differences of a ~10 m swell on a 0.42 m texel lose that much
    # (2) end to end against the oracle's textures: the measured texture error times each texel's condition
    _, bn, bw, delta = or_bounds.assert_normal_white(Nt[..., :3], Wt[..., 0], NT[..., :3], WT[..., 0], rp.length, DT[..., 0], DT[..., 1],
                                                     DT[..., 2], HT[..., 0], got=(Dt[..., 0], Dt[..., 1], Dt[..., 2], Ht[..., 0]),
                                                     tag="rgba", return_bounds=True)
    return bn, bw, delta


def _mesh_checks(oracle, rp, uw, got, dev_tex, want_tex, bounds):
    """The vertex stage (W/TestOcean.shader:61-79), every vertex against its own bound (tests/or_bounds.py): (1) the stage alone
    -- the oracle's f64 stage on the DEVICE's textures, float32 rounding only; (2) end to end against the oracle's mesh from the
    oracle's textures, each vertex with the bound of the texels it samples."""
    (h, d_rb, n, w), (HT, DT, NT, WT), (bn, bw, delta) = dev_tex, want_tex, bounds
    res = rp.resolution
    stage_ref = oracle.renderer_mesh_vertex_stage_f64(rp, uw, h, d_rb, n, w)
    mag = np.maximum(np.abs(h), np.abs(d_rb).max(-1))
    or_bounds.assert_mesh_stage_alone(stage_ref, got, n, res, mag, tag="mesh")
    ref = oracle.renderer_mesh_vertex_stage_f64(rp, uw, HT[..., 0], DT[..., [0, 2]], NT[..., :3], WT[..., 0])
    rn, rc, mn, mc = or_bounds.assert_mesh_end_to_end(ref, got, NT[..., :3], bn, bw, delta, res, mag, tag="mesh")
    # the bounds are not vacuous: the median vertex is held well below the range of a unit normal / a [0,1] colour (emulated
    # kernels: 9e-5 / 3e-5 on 128^2 textures, 5.6e-2 / 1.0e-2 at the shipped 1024^2 scale, where a texel's own bound is ~1e-3
    # -- float32 differences of a ~10 m swell on a 0.42 m texel -- and a vertex takes the largest of its dilated taps)
    assert mn < 0.2 and mc < 0.05, (mn, mc)


def test_emulated_rgba_textures_and_mesh_vertex_stage(emul, oracle):
    """Consumer-side packing: the four ARGBFloat targets in the shaders' channel layout and
    W/TestOcean.shader:61-79 on the S/OceanRenderer.cs:172-207 mesh, host-stepped kernel bodies vs the f64 oracle."""
    rp = shipped(16)
    M = rp.M
    init4 = oracle.renderer_initial_spectrum(rp, 5)
    initT = np.ascontiguousarray(init4.transpose(1, 0, 2))
    phaseT = np.zeros((M, M), np.float32)
    ph = np.zeros((M, M), np.float32)
    for dt in (0.016, 0.3):
        h, d, n, w, g, hg, da = emul.or_step(rp, initT, phaseT, dt, imag=True)
        want = oracle.renderer_textures_f64(rp, init4, ph, dt)
        bounds = _rgba_checks(emul.or_pack_rgba(h, hg, d, g, da, n, w), want, rp, oracle)
    for uw in (1.0, 0.37):
        got = emul.or_displace_mesh(M, rp.resolution, uw, h, d, n, w)
        _mesh_checks(oracle, rp, uw, got, (h, d, n, w), want, bounds)
    # corner vertices sample the clamped corner texels: uv = 0 -> texel 0, uv = 1 -> texel M-1
    v = got[0]
    res = rp.resolution
    assert abs(v[0, 1] - h[0, 0] / 8) < 1e-6 and abs(v[res * res - 1, 1] - h[M - 1, M - 1] / 8) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("resolution", [8, 128])
def test_gpu_rgba_textures_and_mesh_vertex_stage(mw, oracle, resolution):
    rp = shipped(resolution)
    M = rp.M
    uw = 0.75
    o = mw.Ocean(resolution=resolution, unit_width=uw, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude,
                 choppiness=rp.choppiness, gravity=rp.gravity, mult=rp.mult, seed=5, semantics=mw.MW_SEM_OCEANRENDERER)
    with pytest.raises(mw.MistralWaterError) as e:
        o.displace_mesh()                       # no frame yet
    assert e.value.status == mw.MW_ESTATE
    init4 = oracle.renderer_initial_spectrum(rp, 5)
    o.set_spectrum(init4[..., :2], init4[..., 2:])
    ph = np.zeros((M, M), np.float32)
    h0, d0, n0, w0 = o.generate_texture(0.016)                    # compact frame first, RGBA frames afterwards
    oracle.renderer_textures_f64(rp, init4, ph, 0.016)
    for dt in (0.033, 0.3):
        tex = o.generate_texture_rgba(dt)
        want = oracle.renderer_textures_f64(rp, init4, ph, dt)
        bounds = _rgba_checks(tex, want, rp, oracle)
    HT, DT, NT, WT = want
    got = o.displace_mesh()
    Ht, Dt, Nt, Wt = tex
    _mesh_checks(oracle, rp, uw, got, (Ht[..., 0], Dt[..., [0, 2]], Nt[..., :3], Wt[..., 0]), want, bounds)
    rest = o.rest_mesh()[0]
    assert np.abs(got[0][:, [0, 2]] - rest[:, [0, 2]]).max() <= np.abs(DT[..., [0, 2]]).max() / 8 * 1.001
    o.close()
    with mw.Ocean(resolution=64, length=64.0) as f:               # FFTMesh handle: wrong semantics
        with pytest.raises(mw.MistralWaterError) as e:
            f.generate_texture_rgba(0.1)
        assert e.value.status == mw.MW_ESTATE


@pytest.mark.gpu
@pytest.mark.parametrize("resolution", [8, 32, 128, 256, 512])
def test_gpu_generate_texture_vs_oracle(mw, oracle, resolution, or_plan):
    """BASELINE's shipped OceanRenderer configuration (1024^2 textures at resolution 128), smaller ones, the Inspector
    default resolution 256 (2048^2) and the largest supported texture (4096^2, P = 16 kernels)."""
    rp = shipped(resolution)
    M = rp.M
    o = mw.Ocean(resolution=resolution, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude,
                 choppiness=rp.choppiness, gravity=rp.gravity, mult=rp.mult, seed=5, semantics=mw.MW_SEM_OCEANRENDERER)
    assert o.N == M
    init4 = oracle.renderer_initial_spectrum(rp, 5)
    # (1) the device-generated initial spectrum equals the oracle's up to device libm (exp/log/sqrt)
    g0, g0c = o.get_spectrum()
    sc = np.abs(init4).max()
    assert np.abs(g0 - init4[..., :2]).max() < 5e-6 * sc and np.abs(g0c - init4[..., 2:]).max() < 5e-6 * sc
    # (2) with the SAME spectrum injected, every frame matches to float32 transform accuracy
    o.set_spectrum(init4[..., :2], init4[..., 2:])
    ph = np.zeros((M, M), np.float32)
    for dt in (0.016, 0.033, 0.3, 25.0):
        h, d, n, w = o.generate_texture(dt)
        H, D, Nn, W, G = oracle.renderer_step_f64(rp, init4, ph, dt, literal_passes=False)
        tol_check(h, H, 3e-6, "height"); tol_check(d, D, 3e-6, "disp")
        # F/OceanNormal.shader normalises a sum of four cross products that (with its `center = D.rgb` quirk) can nearly
        # cancel; 1/|n| then amplifies float32 rounding.  Each texel is held to its own condition-number-scaled bound.
        or_bounds.assert_normal_white(n, w, Nn, W, rp.length, D[..., 0], G, D[..., 1], H, tag=f"M={M} dt={dt}")
    # the two passes alone, on the device's own textures (the RGBA form hands all four channels out): float32 rounding only
    tex = o.generate_texture_rgba(0.016)
    rn, rw, mn, mw_ = or_bounds.assert_normal_white_stage(oracle, rp, *tex, tag=f"M={M}")
    assert mn < 2e-3 * max(1.0, M / 1024.0) ** 2, mn
    o.close()


@pytest.mark.gpu
def test_gpu_generate_texture_edge_time_steps(mw, oracle, or_plan):
    """deltaTime corners of GenerateTexture() (S/OceanRenderer.cs:216-223; F/FFTCommon.cginc:101-104, the stateful fmod phase): a frame with
    deltaTime = 0 (a paused game: the phase must not move, the textures repeat), a negative step (time scale < 0: fmod of a negative
    argument keeps its sign in HLSL and in C), a step of ten minutes (omega dt ~ 1e4 rad before the fmod), and mult = 0 (the Inspector's
    time multiplier, :223) -- the phase texture bit for bit against the oracle's strict-float32 recurrence, the textures at the usual bounds."""
    for mult in (1.5, 0.0):
        import dataclasses
        rp = dataclasses.replace(shipped(32), mult=mult)
        M = rp.M
        o = mw.Ocean(resolution=32, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
                     gravity=rp.gravity, mult=rp.mult, seed=5, semantics=mw.MW_SEM_OCEANRENDERER)
        init4 = oracle.renderer_initial_spectrum(rp, 5)
        o.set_spectrum(init4[..., :2], init4[..., 2:])
        ph = np.zeros((M, M), np.float32)
        prev = None
        for dt in (0.02, 0.0, -0.25, 600.0, 0.0):
            before = ph.copy()
            h, d, n, w = o.generate_texture(dt)
            H, D, Nn, W, G = oracle.renderer_step_f64(rp, init4, ph, dt, literal_passes=False)      # advances ph in place
            assert (o.get_phase() == ph).all(), (mult, dt)                                         # the recurrence: index-like work, exact
            if dt == 0.0 or mult == 0.0:
                assert (ph == before).all()
                if prev is not None:
                    assert all((a == b).all() for a, b in zip(prev, (h, d, n, w))), "a frame without time must repeat the previous one"
            tol_check(h, H, 3e-6, "height"); tol_check(d, D, 3e-6, "disp")
            or_bounds.assert_normal_white(n, w, Nn, W, rp.length, D[..., 0], G, D[..., 1], H, tag=f"mult={mult} dt={dt}")
            prev = (h, d, n, w)
        o.close()


def _frame_dts(n):
    # an uneven frame clock (Time.deltaTime is never constant): a hitch, a paused frame, a negative step and a ten-minute jump (omega dt ~ 1e4
    # rad: the library fmod path of the phase step, also inside the chain a later frame group re-walks) among ordinary ones
    base = [0.016, 0.0171, 0.033, 0.0, 0.25, 600.0, -0.02, 0.0169]
    return np.array([base[k % len(base)] * (1.0 + 0.01 * (k // len(base))) for k in range(n)], np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("resolution,nframes", [(8, 5), (32, 32), (128, 1), (128, 7), (128, 32), (256, 6), (512, 3)])
def test_gpu_generate_texture_steps_equal_single_calls(mw, oracle, resolution, nframes, or_plan):
    """mw_ocean_generate_texture_steps_device: n consecutive GenerateTexture() calls in one enqueue (S/OceanRenderer.cs:216-307 per
    frame; the phase chain of F/Dispersion.shader:32-41 / F/FFTCommon.cginc:101-104 walked in registers) must be, bit for bit, n
    single calls -- every texture of every frame and the phase texture after the last one -- at the shipped 1024^2 configuration
    (n = 1, 7, 32), the P = 16 kernels (2048^2) and the per-frame spectrum fallback (4096^2); frames 0 / mid / last also against
    the oracle at the usual float32 bounds."""
    rp = shipped(resolution)
    M = rp.M
    kw = dict(resolution=resolution, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
              gravity=rp.gravity, mult=rp.mult, seed=5, semantics=mw.MW_SEM_OCEANRENDERER)
    dts = _frame_dts(nframes)
    with mw.Ocean(**kw) as a, mw.Ocean(**kw) as b:
        assert a.max_frames() == 32
        a.generate_texture(0.05); b.generate_texture(0.05)          # not from a zero phase
        init4 = np.concatenate(a.get_spectrum(), -1)
        ph = a.get_phase().copy()
        H, D, Nn, W = a.generate_texture_steps(dts)
        assert H.shape == (nframes, M, M) and D.shape == (nframes, M, M, 2) and Nn.shape == (nframes, M, M, 3)
        check = sorted({0, nframes // 2, nframes - 1})
        for k in range(nframes):
            h, d, n, w = b.generate_texture(float(dts[k]))
            assert (H[k] == h).all() and (D[k] == d).all() and (Nn[k] == n).all() and (W[k] == w).all(), (resolution, nframes, k)
            if k in check and M <= 1024:
                oh, od, on, ow, og = oracle.renderer_step_f64(rp, init4, ph, float(dts[k]), literal_passes=False)
                tol_check(h, oh, 3e-6, "height"); tol_check(d, od, 3e-6, "disp")
                or_bounds.assert_normal_white(n, w, on, ow, rp.length, od[..., 0], og, od[..., 1], oh, tag=f"M={M} frame {k}")
            elif M <= 1024:
                oracle.renderer_advance_phase(rp, init4, ph, float(dts[k]))
        assert (a.get_phase() == b.get_phase()).all()
        if M <= 1024:
            assert (a.get_phase() == ph).all()                        # the oracle's strict-float32 recurrence, exact
        # the handle's latest frame is frame n-1: the vertex stage and the next single frame continue from it
        va, vb = a.displace_mesh(), b.displace_mesh()
        assert all((x == y).all() for x, y in zip(va, vb))
        ta, tb = a.generate_texture(0.02), b.generate_texture(0.02)
        assert all((x == y).all() for x, y in zip(ta, tb))


@pytest.mark.gpu
def test_gpu_advance_phase_seeks_like_rendered_frames(mw, oracle):
    """mw_ocean_advance_phase: the Dispersion pass alone (F/Dispersion.shader:32-41).  A handle advanced over 75 frames (three launches'
    worth, a ten-minute jump among them) holds, bit for bit, the phase texture of a handle that rendered them and of the oracle's recurrence --
    and renders the next frames identically: what lets rank r of a job start at frame lo of a sequence (time-steps shard)."""
    rp = shipped(16)
    M = rp.M
    kw = dict(resolution=16, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
              gravity=rp.gravity, mult=rp.mult, seed=4, semantics=mw.MW_SEM_OCEANRENDERER)
    dts = _frame_dts(75)
    with mw.Ocean(**kw) as a, mw.Ocean(**kw) as b, mw.Ocean(ntiles=2, **kw) as t:
        init4 = np.concatenate(a.get_spectrum(), -1)
        ph = np.zeros((M, M), np.float32)
        for dt in dts:
            a.generate_texture(float(dt))
            oracle.renderer_advance_phase(rp, init4, ph, float(dt))
        b.advance_phase(dts)
        b.advance_phase([])                                       # zero frames: nothing happens
        assert (a.get_phase() == ph).all() and (b.get_phase() == ph).all()
        fa, fb = a.generate_texture_steps(dts[:5]), b.generate_texture_steps(dts[:5])
        assert all((x == y).all() for x, y in zip(fa, fb))
        t.advance_phase(dts[:40])                                 # a batched handle: every tile's phase
        pt = t.get_phase()
        with mw.Ocean(**kw) as c:
            c.advance_phase(dts[:40])
            assert (pt[0] == c.get_phase()).all() and (pt[1] == c.get_phase()).all()     # omega is the tiles' common table
        with pytest.raises(mw.MistralWaterError) as e:
            mw.check(mw.lib().mw_ocean_advance_phase(a.handle, None, 3))
        assert e.value.status == mw.MW_EINVAL
    with mw.Ocean(resolution=64, length=64.0) as f:
        with pytest.raises(mw.MistralWaterError) as e:
            f.advance_phase([0.1])
        assert e.value.status == mw.MW_ESTATE


@pytest.mark.gpu
def test_gpu_generate_texture_steps_destinations_rgba_and_errors(mw):
    """Caller-owned device destinations ([n][M*M*...]), the ARGBFloat form, a second call that grows the frame buffers, and the
    argument errors of the steps entry points."""
    import ctypes as C
    import torch
    rp = shipped(16)
    M = rp.M
    kw = dict(resolution=16, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
              gravity=rp.gravity, mult=rp.mult, seed=9, semantics=mw.MW_SEM_OCEANRENDERER)
    with mw.Ocean(**kw) as a, mw.Ocean(**kw) as b:
        dts = _frame_dts(3)
        want = a.generate_texture_steps(dts)
        dh = torch.empty((3, M, M), device="cuda"); dd = torch.empty((3, M, M, 2), device="cuda")
        dn = torch.empty((3, M, M, 3), device="cuda"); dw = torch.empty((3, M, M), device="cuda")
        b.generate_texture_steps_device(dts, dh.data_ptr(), dd.data_ptr(), None, dw.data_ptr())
        b.synchronize()
        assert b.frame_textures(1)[0] is None and b.frame_textures(1)[2] is not None      # only the normal stayed in the handle
        assert (dh.cpu().numpy() == want[0]).all() and (dd.cpu().numpy() == want[1]).all() and (dw.cpu().numpy() == want[3]).all()
        # ARGBFloat targets of 9 more frames (grows the frame buffers) == 9 single RGBA frames
        dts = _frame_dts(9)
        tex = [torch.empty((9, M, M, 4), device="cuda") for _ in range(4)]
        a.generate_texture_steps_device(dts, *[t.data_ptr() for t in tex], rgba=True)
        a.synchronize()
        for k in range(9):
            one = b.generate_texture_rgba(float(dts[k]))
            for t, o in zip(tex, one):
                assert (t[k].cpu().numpy() == o).all(), k
        assert (a.get_phase() == b.get_phase()).all()
        # the host forms (what a C# host with managed arrays calls): the same frames, planar and ARGBFloat; a NULL array is skipped
        hp = a.generate_texture_steps(dts[:4]); hb = [b.generate_texture(float(dt)) for dt in dts[:4]]
        assert all((hp[c][k] == hb[k][c]).all() for k in range(4) for c in range(4))
        hr = a.generate_texture_steps(dts[:2], rgba=True); rb = [b.generate_texture_rgba(float(dt)) for dt in dts[:2]]
        assert all((hr[c][k] == rb[k][c]).all() for k in range(2) for c in range(4))
        only_h = np.empty((2, M, M), np.float32)
        mw.check(mw.lib().mw_ocean_generate_texture_steps(a.handle, _frame_dts(2).ctypes.data_as(C.c_void_p), 2, only_h.ctypes.data_as(C.c_void_p), None, None, None))
        assert (only_h[1] == [b.generate_texture(float(dt)) for dt in _frame_dts(2)][1][0]).all()
        for bad in (0, 33):
            with pytest.raises(mw.MistralWaterError) as e:
                a.generate_texture_steps_device(np.zeros(bad, np.float32))
            assert e.value.status == mw.MW_EINVAL
        with pytest.raises(mw.MistralWaterError) as e:
            a.frame_textures(9)
        assert e.value.status == mw.MW_EINVAL
    with mw.Ocean(ntiles=2, **kw) as t:
        assert t.max_frames() == 0
        with pytest.raises(mw.MistralWaterError) as e:
            t.generate_texture_steps_device(_frame_dts(2))
        assert e.value.status == mw.MW_ESTATE
    with mw.Ocean(resolution=64, length=64.0) as f:
        assert f.max_frames() == 0
        with pytest.raises(mw.MistralWaterError) as e:
            f.generate_texture_steps_device(_frame_dts(2))
        assert e.value.status == mw.MW_ESTATE


@pytest.mark.gpu
@pytest.mark.parametrize("resolution", [8, 128])
def test_gpu_packed_plan_against_the_three_transform_plan(mw, oracle, resolution):
    """Round 6: a planar-texture call runs TWO complex transforms per frame -- F(G) = height + i Dz with G built from the Hermitian parts
    (P, Q) of the initial spectrum; hx keeps its own -- where the shaders run three (F/Spectrum.shader:47-50, S/OceanRenderer.cs:229-262).
    (a) Both plans agree to float32 transform accuracy and (b) sit inside the oracle's bounds (also through the Nyquist row py = M/2, where
    the odd multiplier is even: a single-bin spectrum there); (c) the identity needs a mirror-symmetric phase texture: a phase injected
    with mw_ocean_set_phase that is NOT symmetric silently selects the three-transform plan -- bit for bit the result with the switch off --
    and the library's own phase (get_phase -> set_phase) keeps the packed one; (d) the RGBA form always runs three."""
    rp = shipped(resolution)
    M = rp.M
    kw = dict(resolution=resolution, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
              gravity=rp.gravity, mult=rp.mult, seed=5, semantics=mw.MW_SEM_OCEANRENDERER)
    rng = np.random.default_rng(3)

    def frames(o, dts):
        return [o.generate_texture(dt) for dt in dts]
    dts = (0.016, 0.3, 0.033)
    try:
        with mw.Ocean(**kw) as a, mw.Ocean(**kw) as b:
            init4 = np.concatenate(a.get_spectrum(), -1)
            mw.set_switch("MW_OR_PACKED", 1); fa = frames(a, dts)
            mw.set_switch("MW_OR_PACKED", 0); fb = frames(b, dts)
            assert (a.get_phase() == b.get_phase()).all()
            ph = np.zeros((M, M), np.float32)
            for k, dt in enumerate(dts):
                H, D, Nn, W, G = oracle.renderer_step_f64(rp, init4, ph, dt, literal_passes=False)
                for f in (fa[k], fb[k]):
                    tol_check(f[0], H, 3e-6, "height"); tol_check(f[1], D, 3e-6, "disp")
                    or_bounds.assert_normal_white(f[2], f[3], Nn, W, rp.length, D[..., 0], G, D[..., 1], H, tag=f"M={M} frame {k}")
                sc = max(np.abs(H).max(), np.abs(D).max())
                assert np.abs(fa[k][0] - fb[k][0]).max() < 2e-6 * sc and np.abs(fa[k][1] - fb[k][1]).max() < 2e-6 * sc
                assert not (fa[k][0] == fb[k][0]).all()                       # the plans really differ (last bits)
            # single bins on the Nyquist row / column and their neighbours: the even-multiplier special case of G
            for (py, px) in ((M // 2, 3), (M // 2, M // 2), (M // 2 - 1, 5), (7, M // 2), (M // 2, 0), (0, M // 2)):
                h0 = np.zeros((M, M, 2), np.float32); h0c = np.zeros((M, M, 2), np.float32)
                h0[py, px] = (0.7, -0.2); h0c[(M - py) % M, px] = (0.1, 0.4); h0c[py, (M - px) % M] = (-0.3, 0.25)
                i4 = np.concatenate([h0, h0c], -1)
                mw.set_switch("MW_OR_PACKED", 1)
                a.set_spectrum(h0, h0c)
                h, d, n, w = a.generate_texture(0.21)
                ph = np.zeros((M, M), np.float32)
                H, D, Nn, W, G = oracle.renderer_step_f64(rp, i4, ph, 0.21, literal_passes=False)
                tol_check(h, H, 3e-6, f"height bin {py},{px}"); tol_check(d, D, 3e-6, f"disp bin {py},{px}")
            # (c) an asymmetric phase: the three-transform plan whatever the switch says
            a.set_spectrum(init4[..., :2], init4[..., 2:]); b.set_spectrum(init4[..., :2], init4[..., 2:])
            bad = rng.uniform(0, 6.2, (M, M)).astype(np.float32)
            mw.set_switch("MW_OR_PACKED", 1); a.set_phase(bad); ga = a.generate_texture(0.05)
            mw.set_switch("MW_OR_PACKED", 0); b.set_phase(bad); gb = b.generate_texture(0.05)
            assert all((x == y).all() for x, y in zip(ga, gb))
            ph = bad.copy()
            H, D, Nn, W, G = oracle.renderer_step_f64(rp, init4, ph, 0.05, literal_passes=False)
            tol_check(ga[0], H, 3e-6, "height, injected phase"); tol_check(ga[1], D, 3e-6, "disp, injected phase")
            assert (a.get_phase() == ph).all()
            # ... and the library's own phase texture, saved and restored, is symmetric: the packed plan again (its bits, not the other plan's)
            mw.set_switch("MW_OR_PACKED", 1)
            a.set_spectrum(init4[..., :2], init4[..., 2:]); b.set_spectrum(init4[..., :2], init4[..., 2:])
            a.generate_texture(0.4); b.generate_texture(0.4)
            saved = a.get_phase()
            assert (saved == saved[(-np.arange(M)) % M][:, (-np.arange(M)) % M]).all()
            a.set_phase(saved)
            ra, rb = a.generate_texture(0.02), b.generate_texture(0.02)
            assert all((x == y).all() for x, y in zip(ra, rb))
            # (d) the RGBA targets carry Im h and Im Dz: three transforms, and the planar call after it is packed again
            ta = a.generate_texture_rgba(0.02)
            mw.set_switch("MW_OR_PACKED", 0)
            tb = b.generate_texture_rgba(0.02)
            assert all((x == y).all() for x, y in zip(ta, tb))
    finally:
        mw.set_switch("MW_OR_PACKED", 1)


@pytest.mark.gpu
def test_gpu_oceanrenderer_lifecycle(mw):
    r = mw.OceanRenderer()
    r.resolution, r.length, r.amplitude, r.choppiness, r.mult = 16, 60.0, 0.41, 0.46, 1.5
    r.wind = mw.Vector2(14.45, 12.0)
    r.Awake()
    assert r.mesh.vertices.shape == (256, 3) and r.mesh.indices.size == 15 * 15 * 6
    r.Update(0.016)
    h1 = r.heightTexture.copy()
    r.Update(0.016)
    assert r.heightTexture.shape == (128, 128) and not np.array_equal(h1, r.heightTexture)
    assert np.isfinite(r.normalTexture).all() and (r.whiteTexture >= 0).all() and (r.whiteTexture <= 1).all()
    q = mw.OceanRenderer()                 # the same component driven three frames at a time: the per-frame textures, bit for bit
    q.resolution, q.length, q.amplitude, q.choppiness, q.mult = 16, 60.0, 0.41, 0.46, 1.5
    q.wind = mw.Vector2(14.45, 12.0)
    q.Awake()
    H, D, Nn, W = q.GenerateTextures([0.016, 0.016, 0.02])
    assert (H[1] == r.heightTexture).all() and (Nn[1] == r.normalTexture).all()
    r.Update(0.02)
    assert (q.heightTexture == r.heightTexture).all() and (q.whiteTexture == r.whiteTexture).all() and (H[2] == r.heightTexture).all()
    r.wind = mw.Vector2(3.0, 1.0)          # param change -> RenderInitial again (S/OceanRenderer.cs:98-109)
    r.Update(0.016)
    r.Update(0.016)
    assert np.isfinite(r.heightTexture).all()
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Ocean(resolution=12, length=10.0, semantics=mw.MW_SEM_OCEANRENDERER)   # 96 is not a power of two
    assert e.value.status == mw.MW_ENOTPOW2
