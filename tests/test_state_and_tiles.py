"""GPU tier: the handle's state entry points (stream choice, in-place spectrum regeneration that keeps the OceanRenderer
phase, phase/timer save + restore) and the multi-device tile API with its RCCL gather, all through the C ABI.

Reference behaviour pinned here: S/OceanRenderer.cs:91-110 (Update: the parameter-change branch re-runs RenderInitial() and
leaves the ping/pong phase textures alone; normalMat keeps the _Length of SetParams), S/FFTMesh.cs:60-73 (`generate` draws a
fresh spectrum)."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

import or_bounds
import workloads

pytestmark = pytest.mark.gpu


def shipped(resolution=16, length=60.0):
    from oracle.oracle import RendererParams
    return RendererParams(resolution=resolution, length=length, wind_x=14.45, wind_y=12.0, amplitude=0.41, choppiness=0.46, mult=1.5)


def make_or(mw, rp, seed=5):
    return mw.Ocean(resolution=rp.resolution, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude,
                    choppiness=rp.choppiness, gravity=rp.gravity, mult=rp.mult, seed=seed, semantics=mw.MW_SEM_OCEANRENDERER)


def tol(a, b, rel):
    return np.abs(np.asarray(a, np.float64) - b).max() <= rel * max(np.abs(b).max(), 1e-30)


def test_reinit_spectrum_keeps_the_oceanrenderer_phase(mw, oracle):
    """OceanRenderer.Update's parameter-change branch (S/OceanRenderer.cs:98-109)."""
    old = shipped()
    new = dataclasses.replace(old, length=83.5, wind_x=3.0, wind_y=-7.0, amplitude=0.9)
    M = old.M
    init_old = oracle.renderer_initial_spectrum(old, 5)
    init_new = oracle.renderer_initial_spectrum(new, 5)         # same seeds: the reference keeps _RandomSeed1/2
    ph = np.zeros((M, M), np.float32)
    with make_or(mw, old) as o:
        o.set_spectrum(init_old[..., :2], init_old[..., 2:])
        for dt in (0.016, 0.3):
            o.generate_texture(dt)
            oracle.renderer_step_f64(old, init_old, ph, dt, literal_passes=False)
        before = o.get_phase()
        assert (before == ph).all()                             # the strict-f32 phase recurrence, bit for bit
        o.reinit_spectrum(length=new.length, wind=(new.wind_x, new.wind_y), amplitude=new.amplitude)
        assert (o.get_phase() == before).all(), "RenderInitial() must not touch the phase textures"
        g0, g0c = o.get_spectrum()
        sc = np.abs(init_new).max()
        assert np.abs(g0 - init_new[..., :2]).max() < 5e-6 * sc and np.abs(g0c - init_new[..., 2:]).max() < 5e-6 * sc
        o.set_spectrum(init_new[..., :2], init_new[..., 2:])    # identical inputs on both sides from here on ...
        o.set_phase(before)                                     # ... (set_spectrum restarts the phase; put it back)
        for dt in (0.033, 0.5):
            h, d, n, w = o.generate_texture(dt)
            # dispersion + spectrum on the NEW length, normal pass on the OLD one (normalMat._Length is set once, :163)
            H, D, Nn, W, G = oracle.renderer_step_f64(new, init_new, ph, dt, literal_passes=False, normal_params=old)
            assert tol(h, H, 3e-6) and tol(d, D, 3e-6)
            or_bounds.assert_normal_white(n, w, Nn, W, old.length, D[..., 0], G, D[..., 1], H, tag=f"after reinit, dt={dt}")
            assert (o.get_phase() == ph).all()
        # the normal pass really is on the old length: the new one gives visibly different normals
        _, _, Nwrong, _, _ = oracle.renderer_step_f64(new, init_new, ph.copy(), 0.0, literal_passes=False)
        h, d, n, w = o.generate_texture(0.0)
        assert np.abs(n - Nwrong).max() > 1e-3


def test_phase_save_and_restore_resumes_bit_for_bit(mw, oracle):
    rp = shipped()
    with make_or(mw, rp) as a:
        for dt in (0.016, 0.25, 0.033):
            a.generate_texture(dt)
        h0, h0c, ph = *a.get_spectrum(), a.get_phase()
        want = a.generate_texture(0.05)
    with make_or(mw, rp, seed=99) as b:                         # a different sea, overwritten by the checkpoint
        b.set_spectrum(h0, h0c)
        b.set_phase(ph)
        got = b.generate_texture(0.05)
    for x, y in zip(got, want):
        assert (x == y).all()
    with mw.Ocean(resolution=64, length=64.0) as f:             # FFTMesh: the state is the timer
        with pytest.raises(mw.MistralWaterError) as e:
            f.get_phase()
        assert e.value.status == mw.MW_ESTATE
        f.update(0.25)
        f.set_timer(7.5)
        assert f.timer == 7.5
        v, n, c = f.update(0.5)
        v2, n2, c2 = f.evaluate(8.0)
        assert (v == v2).all() and (c == c2).all()


def test_reinit_spectrum_fftmesh(mw, oracle):
    p = workloads.fftmesh_params(128)
    with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                  choppiness=p.choppiness, seed=3) as o:
        o.update(0.5)
        o.reinit_spectrum(wind=(5.0, 3.0), amplitude=2 * p.amplitude, seed=11)
        assert o.timer == 0.5                                   # the timer is not the spectrum's business
        q = dataclasses.replace(p, wind_x=5.0, wind_y=3.0, amplitude=2 * p.amplitude)
        h0, h0c = oracle.generate_spectrum(q, 11)
        g0, g0c = o.get_spectrum()
        sc = np.abs(h0).max()
        assert np.abs(g0 - h0).max() < 4e-6 * sc and np.abs(g0c - h0c).max() < 4e-6 * sc
        o.set_spectrum(h0, h0c)
        v, n, c = o.evaluate(1.25)
        vf, nf, cf, hds = oracle.eval_fft_f64(q, h0, h0c, 1.25, return_hds=True)
        workloads.assert_parity(v, n, c, vf, nf, cf, oracle.rest_mesh(q)[0], np.abs(hds).max(), tag="after reinit")
        with pytest.raises(mw.MistralWaterError) as e:          # 128 * 1.0 != 130: would leave the FFT path
            o.reinit_spectrum(length=130.0)
        assert e.value.status == mw.MW_ESTATE


def test_mirrors_regenerate_and_render_initial(mw):
    m = mw.FFTMesh(seed=4)
    m.resolution, m.unitWidth, m.length, m.amplitude = 64, 1.0, 64.0, 2e-6
    m.wind = mw.Vector2(14.45, 12.0)
    m.Awake()
    a = m.ocean.get_spectrum()[0].copy()
    m.generate = True
    m.Update(0.1)
    b = m.ocean.get_spectrum()[0].copy()
    assert not np.array_equal(a, b), "GenerateMesh draws fresh random values on every regeneration (S/FFTMesh.cs:114-116)"
    m.fixedSeed = True
    m.generate = True
    m.Update(0.1)
    c = m.ocean.get_spectrum()[0].copy()
    m.generate = True
    m.Update(0.1)
    assert np.array_equal(c, m.ocean.get_spectrum()[0])
    r = mw.OceanRenderer()
    r.resolution, r.length, r.amplitude, r.choppiness, r.mult = 16, 60.0, 0.41, 0.46, 1.5
    r.wind = mw.Vector2(14.45, 12.0)
    r.Awake()
    r.Update(0.016)
    r.Update(0.016)
    ph = r.ocean.get_phase()
    assert ph.max() > 0
    r.wind = mw.Vector2(3.0, 1.0)                               # parameter change: RenderInitial again, phase keeps running
    r.Update(0.016)
    ph2 = r.ocean.get_phase()
    assert (ph2 != ph).any() and ph2.max() > 0 and np.isfinite(r.heightTexture).all()
    assert not np.array_equal(ph2, np.zeros_like(ph2))


def test_stream_argument_means_what_it_says(mw, oracle):
    """NULL is HIP's legacy default stream (torch's default current stream); None returns to the private stream."""
    import torch
    p = workloads.fftmesh_params(256)
    NN = 256 * 256
    with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                  choppiness=p.choppiness) as o:
        own = mw.lib().mw_ocean_get_stream(o.handle)
        assert own
        v0, n0, c0 = o.evaluate(1.0)
        o.set_stream(torch.cuda.current_stream().cuda_stream)   # 0 on the default stream
        assert not mw.lib().mw_ocean_get_stream(o.handle)
        dv = torch.zeros((1, NN, 3), device="cuda")
        dn = torch.zeros((1, NN, 3), device="cuda")
        dw = torch.zeros((1, NN), device="cuda")
        o.evaluate_device([1.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        s = (dv.sum(), dw.sum())                                 # torch work on the same stream: ordered, no explicit sync
        assert float(s[0]) == float(torch.from_numpy(v0).cuda().sum())   # the same reduction on bit-identical data
        assert (dv[0].cpu().numpy() == v0).all() and (dw[0].cpu().numpy() == c0[:, 0]).all()
        side = torch.cuda.Stream()
        o.set_stream(side.cuda_stream)
        assert mw.lib().mw_ocean_get_stream(o.handle) == side.cuda_stream
        o.evaluate_device([2.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
        side.synchronize()
        v2, _, _ = o.evaluate(2.0)
        assert (dv[0].cpu().numpy() == v2).all()
        o.set_stream(None)
        assert mw.lib().mw_ocean_get_stream(o.handle) == own


def _d2h(ptr, nfloats):
    hip = C.CDLL("libamdhip64.so")
    out = np.empty(nfloats, np.float32)
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nfloats * 4), 2) == 0
    return out


@pytest.mark.parametrize("ntiles", [1, 3])
def test_tiles_evaluate_and_rccl_gather(mw, ntiles):
    """Tiles on device 0 (a 1-GPU box): the gather runs over a one-rank RCCL communicator (self send/receive) on the side
    stream; the root buffer must equal every tile's own outputs bit for bit, and tile k must be the ocean of seed + k."""
    p = workloads.fftmesh_params(128)
    NN = 128 * 128
    times = [0.5, 1.0, 2.5]
    kw = dict(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
              choppiness=p.choppiness)
    with mw.Tiles(ntiles=ntiles, devices=[0] * ntiles, max_steps=3, seed=7, **kw) as t:
        assert t.count == ntiles and t.local_count == ntiles
        t.evaluate(times)
        t.gather(step=2, root=0)
        t.synchronize()
        ptr, fpt = t.gathered()
        assert ptr and fpt == NN * 7
        got = _d2h(ptr, ntiles * fpt).reshape(ntiles, fpt)
        for k in range(ntiles):
            with mw.Ocean(seed=7 + k, **kw) as o:
                v, n, c = o.evaluate(times[2])
            assert (got[k, :NN * 3].reshape(NN, 3) == v).all(), k
            assert (got[k, NN * 3:NN * 6].reshape(NN, 3) == n).all(), k
            assert (got[k, NN * 6:] == c[:, 0]).all(), k
            dv, dn, dw = t.outputs(k)
            assert (_d2h(dv, 3 * NN * 3).reshape(3, NN, 3)[2] == v).all()
        t.evaluate(times[:1], rgba=True)                        # Color output: 4 floats per vertex travel
        t.gather(step=0, root=ntiles - 1)
        t.synchronize()
        ptr, fpt = t.gathered()
        assert fpt == NN * 10
        got = _d2h(ptr, ntiles * fpt).reshape(ntiles, fpt)
        with mw.Ocean(seed=7, **kw) as o:
            v, n, c = o.evaluate(times[0])
        assert (got[0, NN * 6:].reshape(NN, 4) == c).all()
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Tiles(ntiles=2, devices=[0, 57], max_steps=1, **kw)
    assert e.value.status == mw.MW_EINVAL


def test_tiles_per_process_form_with_one_rank(mw):
    """mw_tiles_create_rank (the torch.distributed.run launch): unique id -> ncclCommInitRank; world of one rank here."""
    p = workloads.fftmesh_params(64)
    NN = 64 * 64
    kw = dict(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
              choppiness=p.choppiness)
    uid = mw.Tiles.unique_id()
    assert len(uid) == 128
    with mw.Tiles(max_steps=2, seed=3, comm_id=uid, rank=0, nranks=1, device=0, **kw) as t:
        assert t.count == 1 and t.local_count == 1
        t.evaluate([0.25, 0.75])
        t.gather(step=1, root=0)
        t.synchronize()
        ptr, fpt = t.gathered()
        got = _d2h(ptr, fpt)
    with mw.Ocean(seed=3, **kw) as o:
        v, n, c = o.evaluate(0.75)
    assert (got[:NN * 3].reshape(NN, 3) == v).all() and (got[NN * 6:] == c[:, 0]).all()


def test_batched_oceanrenderer_handle_equals_single_handles(mw, oracle):
    """mw_ocean_create_batch: ntiles independent oceans advanced by the same three launches per GenerateTexture(); every tile
    must be, bit for bit, the single handle of seed + k (S/OceanRenderer.cs:216-316 per tile)."""
    rp = shipped(resolution=16, length=60.0)
    T = 3
    kw = dict(resolution=rp.resolution, unit_width=0.75, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude,
              choppiness=rp.choppiness, gravity=rp.gravity, mult=rp.mult, semantics=mw.MW_SEM_OCEANRENDERER)
    with mw.Ocean(seed=5, ntiles=T, **kw) as b:
        assert mw.lib().mw_ocean_batch_size(b.handle) == T
        singles = [mw.Ocean(seed=5 + k, **kw) for k in range(T)]
        g0, g0c = b.get_spectrum()
        assert g0.shape == (T, rp.M, rp.M, 2)
        for k, o in enumerate(singles):
            s0, s0c = o.get_spectrum()
            assert (g0[k] == s0).all() and (g0c[k] == s0c).all()
        assert not (g0[0] == g0[1]).all()
        for dt in (0.016, 0.3, 0.033):
            H, D, Nn, W = b.generate_texture(dt)
            for k, o in enumerate(singles):
                h, d, n, w = o.generate_texture(dt)
                assert (H[k] == h).all() and (D[k] == d).all() and (Nn[k] == n).all() and (W[k] == w).all(), (dt, k)
        tex = b.generate_texture_rgba(0.05)
        V, Nv, Cv = b.displace_mesh()
        ph = b.get_phase()
        for k, o in enumerate(singles):
            t1 = o.generate_texture_rgba(0.05)
            for a, c in zip(tex, t1):
                assert (a[k] == c).all()
            v, nv_, cv = o.displace_mesh()
            assert (V[k] == v).all() and (Nv[k] == nv_).all() and (Cv[k] == cv).all()
            assert (ph[k] == o.get_phase()).all()
        # checkpoint of the whole batch into a fresh batched handle (tile-major arrays)
        g0, g0c = b.get_spectrum()
        want = b.generate_texture(0.02)
        with mw.Ocean(seed=77, ntiles=T, **kw) as c:
            c.set_spectrum(g0, g0c)
            c.set_phase(ph)
            got = c.generate_texture(0.02)
        for x, y in zip(got, want):
            assert (x == y).all()
        for o in singles:
            o.close()
    with pytest.raises(mw.MistralWaterError) as e:              # FFTMesh batches in time, not in tiles
        mw.Ocean(resolution=64, length=64.0, ntiles=2)
    assert e.value.status == mw.MW_EINVAL
