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


@pytest.mark.parametrize("N", [50, 200])
def test_reinit_spectrum_on_a_non_fft_grid_rebuilds_the_chirp_tables(mw, oracle, N):
    """Round 5: the chirp-z tables of a non-FFT grid depend on (unit_width, length) and are uploaded by mw_ocean_create and by
    mw_ocean_reinit_spectrum -- never lazily inside an enqueue.  After a reinit with a NEW length (two launches at N = 50, three at N = 200) the
    handle must be the handle a fresh create with those parameters gives, bit for bit, and match the oracle."""
    old = oracle.Params(N=N, unit_width=1.0, length=float(N) * 0.9, wind_x=4.0, wind_y=2.0, amplitude=5e-4, choppiness=0.7)
    new = dataclasses.replace(old, length=float(N) * 1.37, wind_x=-3.0, wind_y=6.0, amplitude=8e-4)
    kw = lambda q, seed: dict(resolution=q.N, unit_width=q.unit_width, length=q.length, wind=(q.wind_x, q.wind_y), amplitude=q.amplitude,
                              choppiness=q.choppiness, seed=seed)
    with mw.Ocean(**kw(old, 3)) as o:
        assert o.max_batch == 1
        o.evaluate(0.5)                                            # tables of the old length in use
        o.reinit_spectrum(length=new.length, wind=(new.wind_x, new.wind_y), amplitude=new.amplitude, seed=9)
        a = o.evaluate(1.25)
        h0, h0c = o.get_spectrum()
    with mw.Ocean(**kw(new, 9)) as f:
        b = f.evaluate(1.25)
        g0, g0c = f.get_spectrum()
    assert (h0 == g0).all() and (h0c == g0c).all()
    assert all((x == y).all() for x, y in zip(a, b)), "reinit + evaluate must equal create + evaluate"
    vd, nd, cd, hds = oracle.eval_matmul_f64(new, h0, h0c, 1.25, return_hds=True)
    workloads.assert_parity(a[0], a[1], a[2], vd, nd, cd, oracle.rest_mesh(new)[0], rel=2e-5, tag=f"non-FFT reinit N={N}", hds=hds, min_decided=0.0)


def test_handles_give_their_device_memory_back(mw):
    """Create / use / destroy, twenty times over every kind of handle (FFT path with a batched and a single-step enqueue, the chirp-z path,
    OceanRenderer with RGBA targets, tiles with a gather = both output sets + the root buffer): the free device memory after the last destroy
    is the free memory before the first create (hipMemGetInfo; torch's caching allocator is not involved -- the library allocates with hipMalloc)."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()

    def free_bytes():
        torch.cuda.synchronize()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    p = workloads.fftmesh_params(256)
    kw = dict(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude, choppiness=p.choppiness)
    NN = 256 * 256
    dv = torch.empty((4, NN, 3), dtype=torch.float32, device="cuda"); dn = torch.empty_like(dv); dw = torch.empty((4, NN), dtype=torch.float32, device="cuda")

    def cycle():
        with mw.Ocean(seed=1, **kw) as o:
            o.evaluate_device([0.1, 0.2, 0.3, 0.4], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
            o.evaluate(0.5)
            o.reinit_spectrum(seed=2)
        with mw.Ocean(resolution=50, unit_width=1.0, length=1.0, wind=(1.0, 1.0), amplitude=1.0, choppiness=1.0, seed=1) as o:
            o.evaluate(0.5)
        with mw.Ocean(resolution=16, length=54.0, wind=(14.45, 12.0), amplitude=0.41, choppiness=0.46, mult=1.5, seed=1,
                      semantics=mw.MW_SEM_OCEANRENDERER) as o:
            o.generate_texture(0.02)
            o.generate_texture_rgba(0.02)
        with mw.Tiles(ntiles=2, devices=[0, 0], max_steps=2, seed=3, **kw) as t:
            t.evaluate([0.5, 1.0])
            t.gather(step=1, root=0)
            t.evaluate([1.5, 2.0])
            t.gather(step=0, root=1)
            t.synchronize()
    cycle()                                  # first use: one-time allocations of the runtime / RCCL itself
    before = free_bytes()
    for _ in range(20):
        cycle()
    after = free_bytes()
    assert before - after < (8 << 20), f"{(before - after) / 2**20:.1f} MiB did not come back after 20 create / destroy cycles"


def test_two_host_threads_drive_two_handles_at_once(mw, oracle):
    """The library keeps no unguarded global state: two host threads (ctypes releases the GIL inside every call), each with its own handle on
    its own stream -- one alternating single steps and 5-step batches on a 512^2 FFT grid, the other stepping a non-FFT grid and the pond's
    host entry point -- produce exactly what each produces alone."""
    import threading
    p = workloads.fftmesh_params(512)
    kwa = dict(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude, choppiness=p.choppiness, seed=4)
    kwb = dict(resolution=65, unit_width=0.5, length=40.0, wind=(3.0, 2.0), amplitude=1e-3, choppiness=0.8, seed=5)
    W, P = workloads.pond_waves8(), workloads.POND
    pos = workloads.pond_lattice(200, seed=1)

    def work_a(out):
        with mw.Ocean(**kwa) as o:
            for k in range(12):
                out.append(o.evaluate(0.1 * (k + 1)))

    def work_b(out):
        with mw.Ocean(**kwb) as o:
            for k in range(12):
                out.append(o.evaluate(0.07 * (k + 1)))
                out.append((mw.gerstner_displace(pos, W, P["amplitude"], P["frequency"], P["steepness"], 0.3 * k),))
    ref_a, ref_b, got_a, got_b = [], [], [], []
    work_a(ref_a); work_b(ref_b)
    errs = []

    def guarded(fn, out):
        try:
            fn(out)
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ta, tb = threading.Thread(target=guarded, args=(work_a, got_a)), threading.Thread(target=guarded, args=(work_b, got_b))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    for ref, got in ((ref_a, got_a), (ref_b, got_b)):
        assert len(ref) == len(got)
        for r, g in zip(ref, got):
            assert all((x == y).all() for x, y in zip(r, g))


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


@pytest.fixture(params=["1", "0"], ids=["rccl_self_send", "device_local_copy"])
def gather_path(request):
    """Tiles that live on the root's device are copied, not sent (round 5); the switch MW_TILES_FORCE_RCCL = 1 sends them through the
    one-rank communicator as round 4 did, so that ncclSend / ncclRecv stay exercised on the 1-GPU box.  Both must deliver the same bytes."""
    import mistral_water
    old = mistral_water.get_switch("MW_TILES_FORCE_RCCL")
    mistral_water.set_switch("MW_TILES_FORCE_RCCL", int(request.param))
    yield request.param
    mistral_water.set_switch("MW_TILES_FORCE_RCCL", old)


@pytest.mark.parametrize("ntiles", [1, 3])
def test_tiles_evaluate_and_rccl_gather(mw, ntiles, gather_path):
    """Tiles on device 0 (a 1-GPU box): the gather runs over a one-rank RCCL communicator (self send/receive) on the side
    stream, or as a device-local copy; the root buffer must equal every tile's own outputs bit for bit, and tile k must be the ocean of seed + k."""
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


def test_gather_sends_from_alternating_output_sets_without_a_snapshot(mw, gather_path):
    """Round 5: a gathered FFTMesh tile owns two output sets.  The gather reads the set the latest evaluate wrote (no copy on the compute
    stream), the next evaluate fills the other one, and an evaluate that comes back to a set waits for the sends that read it: six
    batches with a gather each, never synchronised in between -- every gathered step must be its own batch's, bit for bit."""
    p = workloads.fftmesh_params(256)
    NN = 256 * 256
    kw = dict(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
              choppiness=p.choppiness)
    ntiles, B = 2, 4
    with mw.Tiles(ntiles=ntiles, devices=[0] * ntiles, max_steps=B, seed=3, **kw) as t:
        seen = []
        for b in range(6):
            times = [0.25 * (b * B + k + 1) for k in range(B)]
            t.evaluate(times)
            seen.append(t.outputs(0)[0])
            t.gather(step=B - 1, root=0)
            t.gather(step=B - 1, root=0)      # a second gather of the same batch reads the same set
        assert seen[0] == seen[2] == seen[4] and seen[1] == seen[3] == seen[5] and seen[0] != seen[1]
        t.synchronize()
        ptr, fpt = t.gathered()
        got = _d2h(ptr, ntiles * fpt).reshape(ntiles, fpt)
        t_last = 0.25 * (5 * B + B)
        for k in range(ntiles):
            with mw.Ocean(seed=3 + k, **kw) as o:
                v, n, c = o.evaluate(t_last)
            assert (got[k, :NN * 3].reshape(NN, 3) == v).all() and (got[k, NN * 6:] == c[:, 0]).all(), k
            dv, dn, dw = t.outputs(k)
            assert (_d2h(dv, B * NN * 3).reshape(B, NN, 3)[B - 1] == v).all()
        # a gather between two evaluates that is NOT synchronised must still deliver the FIRST batch: gather, overwrite-attempt, check
        t.evaluate([1.0] * B)
        t.gather(step=0, root=0)
        t.evaluate([2.0] * B)                  # goes to the other set while the sends read the first one
        t.evaluate([3.0] * B)                  # ... and this one to the same other set again (no gather in between)
        t.synchronize()
        got = _d2h(t.gathered()[0], ntiles * fpt).reshape(ntiles, fpt)
        with mw.Ocean(seed=3, **kw) as o:
            v1 = o.evaluate(1.0)[0]
            v3 = o.evaluate(3.0)[0]
        assert (got[0, :NN * 3].reshape(NN, 3) == v1).all()
        assert (_d2h(t.outputs(0)[0], NN * 3).reshape(NN, 3) == v3).all()


def test_baseline_config3_eight_1024_tiles_gathered_on_one_device(mw):
    """BASELINE.json configs[2] as far as ONE device allows: 8 independent 1024^2 tiles (config 2's literal
    parameters, seeds 1..8) created through mw_tiles_create -- all on device 0 here, one per GPU on the 8-GPU node -- a batch of
    time-steps each, then the RCCL gather of the last step to the root (self send/receive over a one-rank communicator on the
    side stream).  gathered[k] must be the single handle of seed k + 1, bit for bit, for every tile."""
    p = workloads.fftmesh_config2(1024)
    N, NN, ntiles = 1024, 1024 * 1024, 8
    times = [(k + 1) / 60.0 for k in range(4)]
    kw = dict(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
              choppiness=p.choppiness, gravity=p.gravity)
    with mw.Tiles(ntiles=ntiles, devices=[0] * ntiles, max_steps=len(times), seed=1, **kw) as t:
        assert t.count == ntiles and t.local_count == ntiles
        t.evaluate(times)
        t.gather(step=len(times) - 1, root=0)
        t.synchronize()
        ptr, fpt = t.gathered()
        assert ptr and fpt == NN * 7                              # 28 B per grid point per tile travel (vertex 12 + normal 12 + whitecap 4)
        got = _d2h(ptr, ntiles * fpt).reshape(ntiles, fpt)
    for k in range(ntiles):
        with mw.Ocean(seed=1 + k, **kw) as o:
            v, n, c = o.evaluate(times[-1])
        assert (got[k, :NN * 3].reshape(NN, 3) == v).all(), f"tile {k}: vertices"
        assert (got[k, NN * 3:NN * 6].reshape(NN, 3) == n).all(), f"tile {k}: normals"
        assert (got[k, NN * 6:] == c[:, 0]).all(), f"tile {k}: whitecap"
        if k:
            assert not (got[k, :NN * 3] == got[0, :NN * 3]).all()    # the tiles ARE different oceans


def test_tiles_per_process_form_with_one_rank(mw, gather_path):
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


def test_normal_length_is_part_of_the_oceanrenderer_checkpoint(mw):
    """After a length change (S/OceanRenderer.cs:98-109) the normal pass keeps the length of SetParams (:163): a checkpoint
    restored into a fresh handle created with the CURRENT length resumes bit for bit only with mw_ocean_set_normal_length."""
    old = shipped()
    with make_or(mw, old) as a:
        a.generate_texture(0.1)
        assert a.normal_length == np.float32(old.length)
        a.reinit_spectrum(length=83.5)
        assert a.normal_length == np.float32(old.length)        # not updated by the parameter-change branch
        a.generate_texture(0.05)
        h0, h0c, ph, nl = *a.get_spectrum(), a.get_phase(), a.normal_length
        want = a.generate_texture(0.02)
    new = dataclasses.replace(old, length=83.5)
    with make_or(mw, new, seed=99) as b:
        b.set_spectrum(h0, h0c)
        b.set_phase(ph)
        wrong = b.generate_texture(0.02)
        assert (wrong[0] == want[0]).all() and not (wrong[2] == want[2]).all()   # textures resume, normals do not ...
        b.set_phase(ph)
        b.normal_length = nl
        got = b.generate_texture(0.02)
    for x, y in zip(got, want):                                  # ... until the third piece of the checkpoint is restored
        assert (x == y).all()
    with mw.Ocean(resolution=64, length=64.0) as f:
        assert f.normal_length == 64.0
        with pytest.raises(mw.MistralWaterError) as e:
            f.normal_length = 3.0
        assert e.value.status == mw.MW_ESTATE


def test_handle_leaves_a_caller_stream_before_it_is_destroyed(mw):
    """The lifetime contract of mw_ocean_set_stream (include/mistral_water.h): the caller's stream outlives its use by the handle;
    the handle leaves it (mw_ocean_use_own_stream) BEFORE the caller destroys it, and carries on on its own stream."""
    hip = C.CDLL("libamdhip64.so")
    s = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(s)) == 0
    p = workloads.fftmesh_params(64)
    with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude) as o:
        v0 = o.evaluate(1.0)[0]
        o.set_stream(s.value)
        v1 = o.evaluate(1.0)[0]
        o.set_stream(None)                                       # back to the handle's own stream (drains the caller's first)
        assert hip.hipStreamDestroy(s) == 0
        v2 = o.evaluate(1.0)[0]
        assert (v0 == v1).all() and (v0 == v2).all()


def test_reinit_spectrum_fftmesh_regenerates_in_place(mw, oracle):
    p = workloads.fftmesh_params(128)
    with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                  choppiness=p.choppiness, seed=3) as o:
        a = o.evaluate(0.5)
        o.reinit_spectrum(seed=4)
        b = o.evaluate(0.5)
        with mw.Ocean(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                      choppiness=p.choppiness, seed=4) as f:
            c = f.evaluate(0.5)
            g = f.get_spectrum()
        assert not (a[0] == b[0]).all()
        for x, y in zip(b, c):
            assert (x == y).all()
        for x, y in zip(o.get_spectrum(), g):
            assert (x == y).all()
        with pytest.raises(mw.MistralWaterError) as e:           # would move the grid off the FFT path: refused, handle intact
            o.reinit_spectrum(length=p.length * 1.01)
        assert e.value.status == mw.MW_ESTATE
        for x, y in zip(o.evaluate(0.5), c):
            assert (x == y).all()


@pytest.mark.parametrize("ntiles", [1, 2])
def test_oceanrenderer_tiles_and_rccl_gather(mw, ntiles, gather_path):
    """OceanRenderer semantics behind mw_tiles_* (tile axis only: the phase recurrence F/FFTCommon.cginc:101-104 serialises
    time): tile k is, bit for bit, the single handle of seed + k, frame after frame, and the gathered buffer carries the four
    result textures of every tile."""
    rp = shipped(resolution=16, length=60.0)
    MM = rp.M * rp.M
    kw = dict(resolution=rp.resolution, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
              gravity=rp.gravity, mult=rp.mult, semantics=mw.MW_SEM_OCEANRENDERER)
    with pytest.raises(mw.MistralWaterError) as e:
        mw.Tiles(ntiles=1, devices=[0], max_steps=33, seed=5, **kw)            # at most 32 frames per enqueue
    assert e.value.status == mw.MW_EINVAL
    singles = [mw.Ocean(seed=5 + k, **kw) for k in range(ntiles)]
    with mw.Tiles(ntiles=ntiles, devices=[0] * ntiles, max_steps=1, seed=5, **kw) as t:
        assert t.count == ntiles and t.N == rp.M
        with pytest.raises(mw.MistralWaterError) as e:
            t.gather()                                                          # no frame yet
        assert e.value.status == mw.MW_ESTATE
        with pytest.raises(mw.MistralWaterError) as e:
            t.evaluate([1.0])                                                   # the FFTMesh entry point refuses OceanRenderer tiles
        assert e.value.status == mw.MW_ESTATE
        for dt in (0.016, 0.3):
            t.generate_texture(dt)
            want = [o.generate_texture(dt) for o in singles]
        t.gather(root=ntiles - 1)
        t.synchronize()
        ptr, fpt = t.gathered()
        assert ptr and fpt == MM * 7
        got = _d2h(ptr, ntiles * fpt).reshape(ntiles, fpt)
        for k in range(ntiles):
            h, d, n, w = want[k]
            assert (got[k, :MM] == h.ravel()).all() and (got[k, MM:3 * MM] == d.ravel()).all(), k
            assert (got[k, 3 * MM:6 * MM] == n.ravel()).all() and (got[k, 6 * MM:] == w.ravel()).all(), k
            ph, pd, pn, pw = t.textures(k)
            assert (_d2h(pn, 3 * MM) == n.ravel()).all()
    for o in singles:
        o.close()


@pytest.mark.parametrize("ntiles", [1, 2])
def test_oceanrenderer_tiles_frames_per_enqueue_and_gather(mw, ntiles, gather_path):
    """OceanRenderer tiles with max_steps > 1 (mw_tiles_generate_texture_steps): every tile runs its frames as ONE enqueue
    (mw_ocean_generate_texture_steps_device: the phase chain of F/FFTCommon.cginc:101-104 in registers) and is, frame by frame and bit
    for bit, the single handle of seed + k called once per frame; mw_tiles_gather(step) collects FRAME `step` of every tile."""
    rp = shipped(resolution=16, length=60.0)
    MM = rp.M * rp.M
    kw = dict(resolution=rp.resolution, length=rp.length, wind=(rp.wind_x, rp.wind_y), amplitude=rp.amplitude, choppiness=rp.choppiness,
              gravity=rp.gravity, mult=rp.mult, semantics=mw.MW_SEM_OCEANRENDERER)
    singles = [mw.Ocean(seed=5 + k, **kw) for k in range(ntiles)]
    F = 5
    dts = [0.016, 0.3, 0.0, 0.02, 0.017]
    with mw.Tiles(ntiles=ntiles, devices=[0] * ntiles, max_steps=F, seed=5, **kw) as t:
        with pytest.raises(mw.MistralWaterError) as e:
            t.generate_texture_steps(dts + [0.1])                                # more frames than max_steps
        assert e.value.status == mw.MW_EINVAL
        want = None
        for rnd in range(2):                                                     # the second enqueue continues the first one's phase
            t.generate_texture_steps(dts)
            want = [[o.generate_texture(dt) for dt in dts] for o in singles]
        t.gather(step=3, root=0)
        t.synchronize()
        ptr, fpt = t.gathered()
        assert ptr and fpt == MM * 7
        got = _d2h(ptr, ntiles * fpt).reshape(ntiles, fpt)
        for k in range(ntiles):
            h, d, n, w = want[k][3]
            assert (got[k, :MM] == h.ravel()).all() and (got[k, MM:3 * MM] == d.ravel()).all(), k
            assert (got[k, 3 * MM:6 * MM] == n.ravel()).all() and (got[k, 6 * MM:] == w.ravel()).all(), k
            ph, pd, pn, pw = t.frames(k)
            fh, fn = _d2h(ph, F * MM).reshape(F, MM), _d2h(pn, F * 3 * MM).reshape(F, 3 * MM)
            for f in range(F):
                assert (fh[f] == want[k][f][0].ravel()).all() and (fn[f] == want[k][f][2].ravel()).all(), (k, f)
            lh = t.textures(k)[0]                                                # the handle's latest frame = the last one
            assert (_d2h(lh, MM) == want[k][F - 1][0].ravel()).all()
        t.generate_texture_steps(dts[:2])                                        # a shorter enqueue: frame 3 is no longer there
        with pytest.raises(mw.MistralWaterError) as e:
            t.gather(step=3, root=0)
        assert e.value.status == mw.MW_EINVAL
        t.generate_texture(0.05)                                                 # one frame through the same tiles
        t.gather(step=0, root=0)
        t.synchronize()
        got = _d2h(t.gathered()[0], ntiles * fpt).reshape(ntiles, fpt)
        for k, o in enumerate(singles):
            for dt in dts[:2]:
                o.generate_texture(dt)
            h = o.generate_texture(0.05)[0]
            assert (got[k, :MM] == h.ravel()).all(), k
    with mw.Tiles(ntiles=1, devices=[0], max_steps=1, seed=5, **kw) as t1:
        with pytest.raises(mw.MistralWaterError) as e:
            t1.frames(0)
        assert e.value.status == mw.MW_ESTATE
    for o in singles:
        o.close()


@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_tiles_on_two_devices_restore_the_callers_device(mw, ndev):
    """Real multi-device gather (skipped where fewer devices are visible: the 1-GPU box skips all three): one tile per device on the
    first `ndev` devices, root = the LAST device; every tile equals Ocean(seed + k) on its own device bit for bit, two batches with
    a gather each (the alternating output sets across xGMI), and the entry points put the caller's current HIP device back."""
    if mw.lib().mw_device_count() < ndev:
        pytest.skip(f"needs {ndev} devices")
    hip = C.CDLL("libamdhip64.so")
    p = workloads.fftmesh_params(128)
    NN = 128 * 128
    kw = dict(resolution=p.N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
              choppiness=p.choppiness)
    dev = C.c_int(-1)
    assert hip.hipSetDevice(0) == 0
    root = ndev - 1
    with mw.Tiles(ntiles=ndev, devices=list(range(ndev)), max_steps=2, seed=11, **kw) as t:
        t.evaluate([0.25, 0.75])
        t.gather(step=0, root=root)
        t.evaluate([0.5, 1.5])                 # the other output set of every tile, while the first gather travels
        t.gather(step=1, root=root)
        t.synchronize()
        assert hip.hipGetDevice(C.byref(dev)) == 0 and dev.value == 0
        ptr, fpt = t.gathered()
        assert hip.hipSetDevice(root) == 0
        got = _d2h(ptr, ndev * fpt).reshape(ndev, fpt)
        assert hip.hipSetDevice(0) == 0
    for k in range(ndev):
        with mw.Ocean(seed=11 + k, device=k, **kw) as o:
            v, n, c = o.evaluate(1.5)
        assert (got[k, :NN * 3].reshape(NN, 3) == v).all() and (got[k, NN * 6:] == c[:, 0]).all(), k


def _run_bench(args, env_extra, nproc=1, timeout=600):
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable]
    if nproc > 1:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(repo, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly ONE JSON line, from rank 0 only: got {len(lines)}"
    return json.loads(lines[0])


def test_bench_two_rank_control_flow_on_one_device():
    """The N > 1 control flow of bench.py as the driver launches it (torch.distributed.run, 2 processes), on the 1-GPU box:
    gloo carries barriers / reductions, both ranks drive cuda:0 (an RCCL communicator cannot hold one device twice, so the
    tile API agrees on its fallback and says so).  Rank 0 alone prints; value is the whole-job aggregate over MAX-of-ranks time."""
    d = _run_bench(["--steps", "64", "--warmup", "32", "--no-cpu-baseline", "--preheat-ms", "20"],      # the default workload: N > 1 carries the headline only
                   {"MW_BENCH_BACKEND": "gloo", "MW_BENCH_SAME_DEVICE": "1"}, nproc=2)
    assert d["configs"] is None
    NN = 1024 * 1024
    assert d["n_gpus"] == 2 and d["steps"] == 64 and d["scaling"] == "weak" and d["config"]["tiles"] == 2
    assert d["config"]["steps_per_enqueue"] == 32 and d["config"]["enqueues_per_region"] == 2 and d["config"]["enqueues_timed"] == 2 * d["repeats"] and d["config"]["pass1_time_group"] == 8
    assert abs(d["value"] - 2 * 64 * NN / (d["ms_per_step"] * 1e-3 * 64)) < 1e-6 * d["value"]
    assert d["parity"].startswith("ok") and d["cpu_baseline"] is None
    assert "amplitude 0.41" in d["config"]["workload"] and d["build_id"]
    # the DEFAULT N > 1 line answers by itself: did RCCL see the ranks, what does the gather cost, what does one ocean sharded in time do
    # (SURVEY 8d config 3 "with and without the final RCCL gather", 8e axis 2).  On this box every rank sits on cuda:0, so each process holds
    # its own one-rank communicator (the line says so); with one rank per GPU the same fields carry ncclCommInitRank's world.
    assert d["config"]["api"].startswith("mw_tiles_") and d["tile_api"]["in_use"] and "through mw_tiles_" in d["parity"]
    assert isinstance(d["rccl_ranks"], int) and d["rccl_ranks"] == 1 and "1 rank each" in d["tile_api"]["communicator"]
    pr = d["per_rank"]
    assert len(pr["median_region_ms"]) == 2 and pr["steps"] == [64, 64] and len(pr["grid_points_per_s"]) == 2 and pr["slowest_over_fastest"] >= 1.0
    assert max(pr["median_region_ms"]) <= d["ms_per_step"] * 64 * 1.5          # the job's region is the slowest rank's
    g = d["with_gather"]
    assert g["gathers"] > 0 and g["gathers_per_region"] == 2 and g["rccl_ranks"] == 1 and g["bytes_per_gather_per_tile"] == 28 * NN
    assert g["value"] > 0 and 0.3 < g["relative_to_value"] < 1.3 and len(g["per_rank_median_region_ms"]) == 2
    st = d["strong"]
    assert "error" not in st, st
    assert st["scaling"] == "strong" and st["config"]["parallelism"] == "steps2" and st["config"]["tiles"] == 1 and st["per_rank"]["steps"] == [32, 32]
    assert abs(st["value"] - 64 * NN / (st["ms_per_step"] * 1e-3 * 64)) < 1e-6 * st["value"] and st["vs_tiles_value"] > 0
    # --shard steps: ONE ocean, the K steps split in contiguous blocks (SURVEY 8e axis 2): strong scaling, K steps in total
    s = _run_bench(["--steps", "64", "--warmup", "32", "--no-cpu-baseline", "--preheat-ms", "20", "--shard", "steps"],
                   {"MW_BENCH_BACKEND": "gloo", "MW_BENCH_SAME_DEVICE": "1"}, nproc=2)
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and s["config"]["tiles"] == 1 and s["config"]["parallelism"] == "steps2"
    assert s["config"]["steps_per_enqueue"] == 32 and s["config"]["enqueues_per_region"] == 1          # each rank: its 32 of the 64 steps
    assert abs(s["value"] - 64 * NN / (s["ms_per_step"] * 1e-3 * 64)) < 1e-6 * s["value"]


def test_bench_renderer_frames_sharded_over_two_ranks():
    """OceanRenderer semantics, N > 1, --shard steps: ONE ocean, rank r renders frames [lo, hi) of the K after seeking there with the
    Dispersion pass alone (mw_ocean_advance_phase) -- in this semantics only the phase links the frames.  K frames in total, strong scaling."""
    r = _run_bench(["--workload", "renderer1024", "--steps", "64", "--warmup", "32", "--no-cpu-baseline", "--preheat-ms", "20", "--shard", "steps"],
                   {"MW_BENCH_BACKEND": "gloo", "MW_BENCH_SAME_DEVICE": "1"}, nproc=2)
    NN = 1024 * 1024
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["parallelism"].startswith("steps2") and r["parity"].startswith("ok")
    assert r["config"]["frames_per_enqueue"] == 32 and r["config"]["enqueue_sizes_timed"] == [32]              # each rank: its 32 of the 64 frames
    assert abs(r["value"] - 64 * NN / (r["ms_per_step"] * 1e-3 * 64)) < 1e-6 * r["value"]


def test_bench_times_what_it_prints_and_gates_the_tile_path():
    """The driver's command line (--steps 20 --warmup 5): ONE 20-step enqueue, pass-1 time group 5, roofline from 20-step
    launches, literal config-2 parameters, frame-at-a-time figures present; and the tile-API path keeps the parity gate."""
    d = _run_bench(["--workload", "ocean1024", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], {})
    c = d["config"]
    assert d["repeats"] >= 5 and d["timed_ms_total"] >= 50.0 and d["region_ms_stats"]["n"] == d["repeats"]     # >= --min-timed-ms of timed work whatever K
    st = d["roofline"]["launch_us_stats"]
    assert st["p10"] <= st["median"] <= st["p90"] and abs(st["mean"] - d["roofline"]["launch_us"]) < 1e-3 * st["mean"]
    assert c["steps_per_enqueue"] == 20 and c["enqueues_per_region"] == 1 and c["enqueues_timed"] >= 5 and c["enqueue_sizes_timed"] == [20] and c["pass1_time_group"] == 5
    assert d["roofline"]["steps_per_launch"] == 20 and d["roofline"]["bytes_per_launch"] == 52 * 1024 * 1024 * 20
    assert "amplitude 0.41" in c["workload"] and d["parity"].startswith("ok")
    assert abs(d["event_ms_per_step"] - d["ms_per_step"]) < 0.25 * d["ms_per_step"]
    f = d["frame_at_a_time"]
    assert f["device_us_per_step"] > 0 and 0 < f["host_ms_per_frame_registered"] <= f["host_ms_per_frame_pageable"] * 2.0   # PCIe-bound either way
    assert d["single_step_us"] == f["device_us_per_step"]
    t = _run_bench(["--workload", "ocean1024", "--steps", "32", "--warmup", "32", "--no-cpu-baseline"], {"MW_BENCH_FORCE_TILES": "1"})
    assert t["config"]["api"].startswith("mw_tiles_") and "through mw_tiles_" in t["parity"]


def test_default_bench_line_carries_every_single_gpu_baseline_config():
    """the driver's command line prints ONE line whose headline is the 1024^2 config and whose `configs` object holds
    BASELINE configs[3] (4096^2) and configs[4] (the pond), each with its own parity gate, roofline object and bounded CPU baseline."""
    d = _run_bench(["--steps", "20", "--warmup", "5"], {}, timeout=900)
    assert d["config"]["grid"] == 1024 and d["parity"].startswith("ok") and d["cpu_baseline"]["value"] > 0
    o, p = d["configs"]["ocean4096"], d["configs"]["pond"]
    assert "error" not in o and "error" not in p, (o.get("error"), p.get("error"))
    assert o["config"]["grid"] == 4096 and o["steps"] == 64 and o["config"]["steps_per_enqueue"] == 32 and o["parity"].startswith("ok")
    assert o["roofline"]["bytes_per_launch"] == 52 * 4096 * 4096 * 32 and 0.2 < o["roofline"]["frac"] < 1.0
    assert o["cpu_baseline"]["kind"] == "port" and o["cpu_baseline"]["value"] > 0 and o["timed_ms_total"] >= 50.0
    assert p["unit"] == "vertices/s" and p["parity"].startswith("ok") and 0.2 < p["roofline"]["frac"] < 1.0
    assert p["cpu_baseline"]["value"] > 0 and p["timed_ms_total"] >= 50.0 and p["config"]["steps_per_launch"] == 32
    assert d["configs_wall_s"] < 150.0
