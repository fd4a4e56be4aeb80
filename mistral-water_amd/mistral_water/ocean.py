"""Ocean handle + FFTMesh / OceanRenderer mirrors (reference: Assets/Mistral Water/Scripts/*.cs)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as nat


_hip = None


def _d2h(ptr, shape, dtype=np.float32):
    """Copy a raw device pointer handed out by the C ABI into a new host array (hipMemcpy, synchronous)."""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    out = np.empty(shape, dtype)
    rc = _hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2)
    if rc != 0:
        raise RuntimeError(f"hipMemcpy D2H failed ({rc})")
    return out


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class Vector2:
    x: float = 0.0
    y: float = 0.0


class Ocean:
    """Thin RAII wrapper of an ``mw_ocean*``."""

    def __init__(self, *, resolution, unit_width=1.0, length=1.0, wind=(1.0, 1.0), amplitude=1.0, choppiness=1.0,
                 gravity=9.81, t_division=1.0, mult=1.0, seed=1, semantics=nat.MW_SEM_FFTMESH, device=0, ntiles=1):
        """ntiles > 1 (OceanRenderer semantics): a batched handle, tile k = seed + k; every OceanRenderer array then carries a
        leading tile axis (mw_ocean_create_batch)."""
        self._h = C.c_void_p()
        self.params = nat.MwParams(int(resolution), float(unit_width), float(length), float(wind[0]), float(wind[1]),
                                   float(amplitude), float(choppiness), float(gravity), float(t_division), float(mult),
                                   int(seed), int(semantics), int(device))
        self.ntiles = int(ntiles)
        if self.ntiles == 1:
            nat.check(nat.lib().mw_ocean_create(C.byref(self.params), C.byref(self._h)))
        else:
            nat.check(nat.lib().mw_ocean_create_batch(C.byref(self.params), self.ntiles, C.byref(self._h)))
        self.N = nat.lib().mw_ocean_grid_size(self._h)
        self.mesh_resolution = int(resolution)
        self.semantics = int(semantics)

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            nat.lib().mw_ocean_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream: int | None):
        """Run the handle's work on `hip_stream` (a hipStream_t as int).  0 is HIP's legacy default stream -- what
        ``torch.cuda.current_stream().cuda_stream`` is unless a side stream is current; ``None`` returns to the handle's
        own private non-blocking stream (mw_ocean_use_own_stream)."""
        if hip_stream is None:
            nat.check(nat.lib().mw_ocean_use_own_stream(self._h))
        else:
            nat.check(nat.lib().mw_ocean_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        nat.check(nat.lib().mw_ocean_synchronize(self._h))

    def set_choppiness(self, c: float):
        nat.check(nat.lib().mw_ocean_set_choppiness(self._h, C.c_float(c)))

    # -- spectrum ----------------------------------------------------------------------------
    def set_spectrum(self, h0, h0conj):
        h0 = np.ascontiguousarray(h0, np.float32)
        h0conj = np.ascontiguousarray(h0conj, np.float32)
        assert h0.size == 2 * self.N * self.N * self.ntiles and h0conj.size == h0.size
        nat.check(nat.lib().mw_ocean_set_spectrum(self._h, _p(h0), _p(h0conj)))

    def _t(self, *shape):
        """Array shape with the leading tile axis of a batched handle."""
        return ((self.ntiles,) if self.ntiles > 1 else ()) + tuple(shape)

    def get_spectrum(self):
        h0 = np.empty(self._t(self.N, self.N, 2), np.float32)
        h0c = np.empty(self._t(self.N, self.N, 2), np.float32)
        nat.check(nat.lib().mw_ocean_get_spectrum(self._h, _p(h0), _p(h0c)))
        return h0, h0c

    def reinit_spectrum(self, length=None, wind=None, amplitude=None, seed=None):
        """Regenerate the initial spectrum in place (mw_ocean_reinit_spectrum); None keeps the handle's current value.
        OceanRenderer: RenderInitial() again, phase untouched (S/OceanRenderer.cs:98-109)."""
        p = self.params
        length = p.length if length is None else float(length)
        wx, wy = (p.wind_x, p.wind_y) if wind is None else (float(wind[0]), float(wind[1]))
        amplitude = p.amplitude if amplitude is None else float(amplitude)
        seed = p.seed if seed is None else int(seed)
        nat.check(nat.lib().mw_ocean_reinit_spectrum(self._h, C.c_float(length), C.c_float(wx), C.c_float(wy), C.c_float(amplitude),
                                                     C.c_uint64(seed)))
        p.length, p.wind_x, p.wind_y, p.amplitude, p.seed = length, wx, wy, amplitude, seed

    def get_phase(self):
        """OceanRenderer: the stateful phase texture [M, M] (texel (px,py) at [py, px])."""
        ph = np.empty(self._t(self.N, self.N), np.float32)
        nat.check(nat.lib().mw_ocean_get_phase(self._h, _p(ph)))
        return ph

    def set_phase(self, phase):
        ph = np.ascontiguousarray(phase, np.float32)
        assert ph.size == self.N * self.N * self.ntiles
        nat.check(nat.lib().mw_ocean_set_phase(self._h, _p(ph)))

    @property
    def normal_length(self) -> float:
        """OceanRenderer: normalMat._Length (S/OceanRenderer.cs:163) -- set at creation, not by reinit_spectrum; part of a checkpoint."""
        return float(nat.lib().mw_ocean_normal_length(self._h))

    @normal_length.setter
    def normal_length(self, v: float):
        nat.check(nat.lib().mw_ocean_set_normal_length(self._h, C.c_float(v)))

    def set_timer(self, t: float):
        nat.check(nat.lib().mw_ocean_set_timer(self._h, C.c_float(t)))

    def rest_mesh(self):
        n = self.mesh_resolution
        v = np.empty((n * n, 3), np.float32)
        nr = np.empty((n * n, 3), np.float32)
        uv = np.empty((n * n, 2), np.float32)
        idx = np.empty((nat.lib().mw_ocean_index_count(self._h),), np.int32)
        nat.check(nat.lib().mw_ocean_rest_mesh(self._h, _p(v), _p(nr), _p(uv), _p(idx)))
        return v, nr, uv, idx

    # -- FFTMesh semantics -------------------------------------------------------------------
    def evaluate(self, t: float):
        """EvaluateWaves(t): (vertices [N*N,3], normals [N*N,3], colors [N*N,4]) on the host."""
        NN = self.N * self.N
        v = np.empty((NN, 3), np.float32)
        n = np.empty((NN, 3), np.float32)
        c = np.empty((NN, 4), np.float32)
        nat.check(nat.lib().mw_ocean_evaluate(self._h, C.c_float(t), _p(v), _p(n), _p(c)))
        return v, n, c

    def evaluate_into(self, t: float, vertices, normals, colors):
        """EvaluateWaves(t) into caller arrays kept across frames (the reference's vertMeow/normals, S/FFTMesh.cs:90-99);
        register them once with ``host_register`` and the copy runs at PCIe rate instead of the pageable ~9 GB/s."""
        nat.check(nat.lib().mw_ocean_evaluate(self._h, C.c_float(t), _p(vertices), _p(normals), _p(colors)))

    def update(self, delta_time: float):
        NN = self.N * self.N
        v = np.empty((NN, 3), np.float32)
        n = np.empty((NN, 3), np.float32)
        c = np.empty((NN, 4), np.float32)
        nat.check(nat.lib().mw_ocean_update(self._h, C.c_float(delta_time), _p(v), _p(n), _p(c)))
        return v, n, c

    @property
    def timer(self) -> float:
        return nat.lib().mw_ocean_timer(self._h)

    def reset_timer(self):
        nat.check(nat.lib().mw_ocean_reset_timer(self._h))

    @property
    def max_batch(self) -> int:
        return nat.lib().mw_ocean_max_batch(self._h)

    def evaluate_device(self, times, d_vertices: int, d_normals: int, d_white: int, rgba: bool = False):
        """Enqueue len(times) independent time-steps; outputs are raw device pointers (ints)."""
        tt = np.ascontiguousarray(times, np.float32)
        nat.check(nat.lib().mw_ocean_evaluate_device(self._h, _p(tt), tt.size, C.c_void_p(d_vertices),
                                                     C.c_void_p(d_normals), C.c_void_p(d_white),
                                                     nat.MW_OUT_COLOR_RGBA if rgba else nat.MW_OUT_WHITE_SCALAR))

    def profile_kernels(self, nsteps: int = 1, iters: int = 20):
        ms = (C.c_float * 8)()
        names = (C.c_char_p * 8)()
        nk = C.c_int32(0)
        nat.check(nat.lib().mw_ocean_profile_kernels(self._h, nsteps, iters, ms, names, C.byref(nk)))
        return [(names[k].decode(), float(ms[k])) for k in range(nk.value)]

    def profile_kernels_stats(self, nsteps: int = 1, iters: int = 20):
        """[(kernel name, {mean, median, p10, p90, min, max} in ms)] over `iters` launches (mw_ocean_profile_kernels_stats)."""
        st = (C.c_float * 48)()
        names = (C.c_char_p * 8)()
        nk = C.c_int32(0)
        nat.check(nat.lib().mw_ocean_profile_kernels_stats(self._h, nsteps, iters, st, names, C.byref(nk)))
        keys = ("mean", "median", "p10", "p90", "min", "max")
        return [(names[k].decode(), {kk: float(st[6 * k + i]) for i, kk in enumerate(keys)}) for k in range(nk.value)]

    def debug_evaluate_hds(self, t: float):
        """EvaluateWaves(t) plus hds [N*N, 2] exactly as the kernels hold it (test hook of the whitecap stage)."""
        NN = self.N * self.N
        v, n = np.empty((NN, 3), np.float32), np.empty((NN, 3), np.float32)
        c, h = np.empty((NN, 4), np.float32), np.empty((NN, 2), np.float32)
        nat.check(nat.lib().mw_debug_evaluate_hds(self._h, C.c_float(t), _p(v), _p(n), _p(c), _p(h)))
        return v, n, c, h

    def debug_omega_t(self, t: float):
        out = np.empty((self.N, self.N), np.float32)
        nat.check(nat.lib().mw_debug_omega_t(self._h, C.c_float(t), _p(out)))
        return out

    # -- OceanRenderer semantics -------------------------------------------------------------
    def generate_texture(self, delta_time: float):
        M = self.N
        h = np.empty(self._t(M, M), np.float32)
        d = np.empty(self._t(M, M, 2), np.float32)
        n = np.empty(self._t(M, M, 3), np.float32)
        w = np.empty(self._t(M, M), np.float32)
        nat.check(nat.lib().mw_ocean_generate_texture(self._h, C.c_float(delta_time), _p(h), _p(d), _p(n), _p(w)))
        return h, d, n, w

    def generate_texture_rgba(self, delta_time: float):
        """One GenerateTexture() as the reference's four ARGBFloat render targets, [M,M,4] each
        (S/OceanRenderer.cs:143-146): height (Re h, Im h, Re h, Im h), displacement (Re Dx, Im Dx, Re Dz, Im Dz),
        normal (n, 1), white (w, w, w, 1)."""
        M = self.N
        t = [np.empty(self._t(M, M, 4), np.float32) for _ in range(4)]
        nat.check(nat.lib().mw_ocean_generate_texture_rgba(self._h, C.c_float(delta_time), *[_p(a) for a in t]))
        return tuple(t)

    def advance_phase(self, delta_times):
        """The phase texture after len(delta_times) more frames, no textures produced (mw_ocean_advance_phase): seek / skip."""
        dts = np.ascontiguousarray(delta_times, np.float32)
        nat.check(nat.lib().mw_ocean_advance_phase(self._h, _p(dts), int(dts.size)))

    def max_frames(self) -> int:
        return nat.lib().mw_ocean_max_frames(self._h)

    def generate_texture_steps_device(self, delta_times, d_height=None, d_disp=None, d_normal=None, d_white=None, rgba: bool = False):
        """len(delta_times) consecutive GenerateTexture() calls in one enqueue (asynchronous).  Device destinations are raw
        pointers of [n][M*M*...] arrays; None keeps the frames of that texture in the handle (frame_textures)."""
        dts = np.ascontiguousarray(delta_times, np.float32)
        fn = nat.lib().mw_ocean_generate_texture_steps_rgba_device if rgba else nat.lib().mw_ocean_generate_texture_steps_device
        nat.check(fn(self._h, _p(dts), int(dts.size), d_height, d_disp, d_normal, d_white))

    def frame_textures(self, frame: int):
        """Device pointers (ints, None when that texture went to a caller buffer) of frame `frame` of the latest steps call."""
        ptrs = [C.c_void_p() for _ in range(4)]
        nat.check(nat.lib().mw_ocean_frame_textures(self._h, int(frame), *[C.byref(q) for q in ptrs]))
        return tuple(q.value for q in ptrs)

    def generate_texture_steps(self, delta_times, rgba: bool = False):
        """Host form (mw_ocean_generate_texture_steps[_rgba]): -> (height [n,M,M], disp [n,M,M,2], normal [n,M,M,3], white [n,M,M]), or the
        four ARGBFloat targets [n,M,M,4] each with rgba=True."""
        dts = np.ascontiguousarray(delta_times, np.float32)
        n, M = int(dts.size), self.N
        tails = ((4,),) * 4 if rgba else ((), (2,), (3,), ())
        out = tuple(np.empty((n, M, M) + t, np.float32) for t in tails)
        fn = nat.lib().mw_ocean_generate_texture_steps_rgba if rgba else nat.lib().mw_ocean_generate_texture_steps
        nat.check(fn(self._h, _p(dts), n, *[_p(a) for a in out]))
        return out

    def displace_mesh(self):
        """The ocean material's vertex stage (W/TestOcean.shader:61-79) on the resolution^2 mesh, from the textures
        of the latest generate_texture*(): -> (vertices [n,3], normals [n,3], colors [n])."""
        n = self.mesh_resolution * self.mesh_resolution
        v, nr, c = np.empty(self._t(n, 3), np.float32), np.empty(self._t(n, 3), np.float32), np.empty(self._t(n), np.float32)
        nat.check(nat.lib().mw_ocean_displace_mesh(self._h, _p(v), _p(nr), _p(c)))
        return v, nr, c


class _Mesh:
    """The handful of UnityEngine.Mesh members the reference assigns (S/FFTMesh.cs:134-138,277-279)."""
    vertices = normals = colors = uv = indices = None


class FFTMesh:
    """Mirror of ``public class FFTMesh : MonoBehaviour`` (S/FFTMesh.cs): same public fields
    (:9-23), same Awake()/Update() lifecycle (:60-84); the private numerical methods are the GPU."""

    def __init__(self, device: int = 0, seed: int = 1):
        self.choppiness = 1.0          # :9
        self.tDivision = 1.0           # :11
        self.resolution = 50           # :13
        self.unitWidth = 1.0           # :15
        self.generate = False          # :17
        self.length = 1.0              # :19
        self.wind = Vector2(1.0, 1.0)  # :21
        self.amplitude = 1.0           # :23
        self.gravity = 9.81            # G, :52
        self.mesh = _Mesh()
        self._device, self._seed = device, seed
        self.fixedSeed = False         # True: every regeneration reproduces the same sea (tests); the reference draws anew
        self._generation = 0
        self._ocean = None
        self._timer = 0.0

    # SetParams + GenerateMesh (:90-139).  GenerateMesh draws fresh UnityEngine.Random values every time it runs
    # (:114-116), so each regeneration is a NEW sea state: the seed advances with a generation counter.
    def _regenerate(self):
        if self._ocean is not None:
            self._ocean.close()
        seed = self._seed if self.fixedSeed else self._seed + self._generation
        self._generation += 1
        self._ocean = Ocean(resolution=self.resolution, unit_width=self.unitWidth, length=self.length,
                            wind=(self.wind.x, self.wind.y), amplitude=self.amplitude, choppiness=self.choppiness,
                            gravity=self.gravity, t_division=self.tDivision, seed=seed, device=self._device)
        v, n, uv, idx = self._ocean.rest_mesh()
        self.mesh.vertices, self.mesh.normals, self.mesh.uv, self.mesh.indices = v, n, uv, idx

    def Awake(self):  # :75-84
        self._regenerate()

    def Update(self, deltaTime: float):  # :60-73
        if self.generate:
            self._timer = 0.0
            self._regenerate()
            self.generate = False
        self._timer = float(np.float32(self._timer) + np.float32(deltaTime) / np.float32(self.tDivision))  # :70
        self.EvaluateWaves(self._timer)

    def EvaluateWaves(self, t: float):  # :224-280
        self._ocean.set_choppiness(self.choppiness)  # :244-245 read the live field
        v, n, c = self._ocean.evaluate(t)
        self.mesh.vertices, self.mesh.normals, self.mesh.colors = v, n, c  # :277-279

    @property
    def timer(self):
        return self._timer

    @property
    def ocean(self) -> Ocean:
        return self._ocean


class OceanRenderer:
    """Mirror of ``public class OceanRenderer : MonoBehaviour`` (S/OceanRenderer.cs:10-19, :76-110)."""

    def __init__(self, device: int = 0, seed: int = 1):
        self.mult = 2.0            # :11
        self.unitWidth = 1.0       # :12
        self.resolution = 256      # :13
        self.length = 256.0        # :14
        self.choppiness = 1.5      # :16
        self.amplitude = 1.0       # :18
        self.wind = Vector2()      # :19
        self.gravity = 9.81
        self.mesh = _Mesh()
        self.heightTexture = self.displacementTexture = self.normalTexture = self.whiteTexture = None
        self._device, self._seed = device, seed
        self._ocean = None
        self._old = None

    def _key(self):
        return (self.length, self.wind.x, self.wind.y, self.amplitude)

    def _create(self):
        if self._ocean is not None:
            self._ocean.close()
        self._ocean = Ocean(resolution=self.resolution, unit_width=self.unitWidth, length=self.length,
                            wind=(self.wind.x, self.wind.y), amplitude=self.amplitude, choppiness=self.choppiness,
                            gravity=self.gravity, mult=self.mult, seed=self._seed,
                            semantics=nat.MW_SEM_OCEANRENDERER, device=self._device)
        self._old = self._key()

    def Awake(self):  # :76-89  SetParams, GenerateMesh, RenderInitial
        self._create()
        v, n, uv, idx = self._ocean.rest_mesh()
        self.mesh.vertices, self.mesh.normals, self.mesh.uv, self.mesh.indices = v, n, uv, idx

    def Update(self, deltaTime: float):  # :91-110
        self.GenerateTexture(deltaTime)                   # with the values the materials carried into this frame (:93)
        self._ocean.set_choppiness(self.choppiness)       # spectrumMat._Choppiness for the NEXT frame (:96)
        if self._old != self._key():  # :98-109 RenderInitial again with the same seeds; the phase textures keep running
            self._ocean.reinit_spectrum(length=self.length, wind=(self.wind.x, self.wind.y), amplitude=self.amplitude)
            self._old = self._key()

    def GenerateTexture(self, deltaTime: float):  # :216-316
        h, d, n, w = self._ocean.generate_texture(deltaTime)
        self.heightTexture, self.displacementTexture, self.normalTexture, self.whiteTexture = h, d, n, w

    def GenerateTextures(self, deltaTimes):
        """Not in the reference: len(deltaTimes) consecutive GenerateTexture() calls in one enqueue (bit-identical to the per-frame calls);
        returns the frames [n, M, M(, c)] and leaves the LAST one in the component's textures, as n calls of Update would."""
        H, D, Nn, W = self._ocean.generate_texture_steps(deltaTimes)
        self.heightTexture, self.displacementTexture, self.normalTexture, self.whiteTexture = H[-1], D[-1], Nn[-1], W[-1]
        return H, D, Nn, W

    @property
    def ocean(self) -> Ocean:
        return self._ocean


class Tiles:
    """Independent tiles (seed = seed0 + k) on several devices with an RCCL gather of finished outputs
    (include/mistral_water.h, mw_tiles_*).  ``devices`` lists one device ordinal per tile (single-process form); pass
    ``comm_id``/``rank``/``nranks`` instead for the one-process-per-GPU form.  FFTMesh semantics (default): ``evaluate`` /
    ``outputs``; OceanRenderer semantics (``max_steps`` 1, the tile axis is the only one that shards there):
    ``generate_texture`` / ``textures``."""

    def __init__(self, *, resolution, ntiles=1, devices=None, max_steps=1, unit_width=1.0, length=1.0, wind=(1.0, 1.0), amplitude=1.0,
                 choppiness=1.0, gravity=9.81, seed=1, comm_id=None, rank=0, nranks=1, device=0, semantics=nat.MW_SEM_FFTMESH,
                 mult=1.0):
        self._h = C.c_void_p()
        self.params = nat.MwParams(int(resolution), float(unit_width), float(length), float(wind[0]), float(wind[1]),
                                   float(amplitude), float(choppiness), float(gravity), 1.0, float(mult), int(seed), int(semantics), 0)
        if comm_id is None:
            dv = None if devices is None else (C.c_int32 * ntiles)(*[int(d) for d in devices])
            nat.check(nat.lib().mw_tiles_create(C.byref(self.params), int(ntiles), dv, int(max_steps), C.byref(self._h)))
        else:
            buf = (C.c_ubyte * nat.MW_COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
            nat.check(nat.lib().mw_tiles_create_rank(C.byref(self.params), int(device), int(max_steps), buf, int(rank), int(nranks),
                                                     C.byref(self._h)))
        self.N = int(resolution) * (8 if semantics == nat.MW_SEM_OCEANRENDERER else 1)   # synthesis grid (S/OceanRenderer.cs:136)
        self.max_steps = int(max_steps)

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_ubyte * nat.MW_COMM_ID_BYTES)()
        nat.check(nat.lib().mw_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            nat.lib().mw_tiles_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def count(self):
        return nat.lib().mw_tiles_count(self._h)

    @property
    def local_count(self):
        return nat.lib().mw_tiles_local_count(self._h)

    def set_spectrum(self, k, h0, h0conj):
        o = nat.lib().mw_tiles_ocean(self._h, int(k))
        h0 = np.ascontiguousarray(h0, np.float32)
        h0conj = np.ascontiguousarray(h0conj, np.float32)
        nat.check(nat.lib().mw_ocean_set_spectrum(C.c_void_p(o), _p(h0), _p(h0conj)))

    def get_spectrum(self, k):
        o = nat.lib().mw_tiles_ocean(self._h, int(k))
        h0 = np.empty((self.N, self.N, 2), np.float32)
        h0c = np.empty((self.N, self.N, 2), np.float32)
        nat.check(nat.lib().mw_ocean_get_spectrum(C.c_void_p(o), _p(h0), _p(h0c)))
        return h0, h0c

    def evaluate(self, times, rgba: bool = False):
        tt = np.ascontiguousarray(times, np.float32)
        nat.check(nat.lib().mw_tiles_evaluate(self._h, _p(tt), tt.size, nat.MW_OUT_COLOR_RGBA if rgba else nat.MW_OUT_WHITE_SCALAR))

    def outputs(self, k):
        """Device pointers (ints) of local tile k: vertices, normals, white."""
        v, n, w = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nat.check(nat.lib().mw_tiles_outputs(self._h, int(k), C.byref(v), C.byref(n), C.byref(w)))
        return v.value, n.value, w.value

    def generate_texture(self, delta_time: float):
        """OceanRenderer tiles: one GenerateTexture() on every local tile (asynchronous)."""
        nat.check(nat.lib().mw_tiles_generate_texture(self._h, C.c_float(delta_time)))

    def textures(self, k):
        """OceanRenderer tiles: device pointers (ints) of local tile k's height, disp_xz, normal_xyz, white textures."""
        h, d, n, w = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nat.check(nat.lib().mw_tiles_textures(self._h, int(k), C.byref(h), C.byref(d), C.byref(n), C.byref(w)))
        return h.value, d.value, n.value, w.value

    def generate_texture_steps(self, delta_times):
        """OceanRenderer tiles (max_steps > 1): len(delta_times) consecutive GenerateTexture() calls on every local tile, one enqueue each."""
        dts = np.ascontiguousarray(delta_times, np.float32)
        nat.check(nat.lib().mw_tiles_generate_texture_steps(self._h, _p(dts), int(dts.size)))

    def frames(self, k):
        """OceanRenderer tiles (max_steps > 1): device pointers of local tile k's [max_steps] frames: height, disp_xz, normal_xyz, white."""
        h, d, n, w = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nat.check(nat.lib().mw_tiles_frames(self._h, int(k), C.byref(h), C.byref(d), C.byref(n), C.byref(w)))
        return h.value, d.value, n.value, w.value

    def gather(self, step: int = 0, root: int = 0):
        nat.check(nat.lib().mw_tiles_gather(self._h, int(step), int(root)))

    def gathered(self):
        """(device pointer of the root buffer or None, floats per tile)."""
        ptr, fpt = C.c_void_p(), C.c_int64()
        nat.check(nat.lib().mw_tiles_gathered(self._h, C.byref(ptr), C.byref(fpt)))
        return ptr.value, fpt.value

    def synchronize(self):
        nat.check(nat.lib().mw_tiles_synchronize(self._h))


def host_register(array):
    """Page-lock a numpy array for the lifetime of its use as an output buffer (mw_host_register)."""
    nat.check(nat.lib().mw_host_register(_p(array), array.nbytes))


def host_unregister(array):
    nat.check(nat.lib().mw_host_unregister(_p(array)))


def gerstner_displace(pos_xyz, waves, amplitude: float, frequency: float, steepness: float, t: float, device: int = 0):
    """W/MistralWaterLib.cginc:154-180 Displacement() in Gerstner mode on host arrays.
    ``waves`` = [(dir.x, dir.y, speed), ...]; ``amplitude`` is the already x0.01-scaled value (:172)."""
    pos = np.ascontiguousarray(pos_xyz, np.float32)
    wv = np.ascontiguousarray(waves, np.float32).reshape(-1, 3)
    out = np.empty_like(pos)
    nat.check(nat.lib().mw_gerstner_displace(_p(pos), pos.size // 3, _p(wv), wv.shape[0], C.c_float(amplitude),
                                             C.c_float(frequency), C.c_float(steepness), C.c_float(t), _p(out), device))
    return out


def gerstner_displace_steps_device(d_pos: int, nverts: int, waves, amplitude: float, frequency: float, steepness: float,
                                   times, d_out: int, stream: int = 0):
    """len(times) displaced copies of one device-resident lattice in one launch: d_out is [len(times)][nverts*3]."""
    wv = np.ascontiguousarray(waves, np.float32).reshape(-1, 3)
    tt = np.ascontiguousarray(times, np.float32)
    nat.check(nat.lib().mw_gerstner_displace_steps_device(C.c_void_p(d_pos), nverts, _p(wv), wv.shape[0], C.c_float(amplitude),
                                                          C.c_float(frequency), C.c_float(steepness), _p(tt), tt.size,
                                                          C.c_void_p(d_out), C.c_void_p(stream) if stream else None))


class PondMaterial:
    """Displacement properties of the pond material (W/MistralWaterProperty.cginc, W/MistralWaterLib.cginc:53-66), under
    the material's own property names, and its vertex-stage ``Displacement()`` (:154-180) on host or device arrays.

    ``mode``: ``MW_POND_WAVE`` (Wave, :127-152), ``MW_POND_GERSTNER`` (Gerstner, :71-99) or
    ``MW_POND_GERSTNER_LEVEL_ONE`` (GerstnerLevelOne, :101-125)."""

    def __init__(self, mode=nat.MW_POND_GERSTNER, _Amplitude=10.0, _Frequency=2.58, _Speed=1.0, _Steepness=0.99,
                 _Smoothing=1.0, _WSpeed=(0, 0, 0, 0), _WDirectionAB=(0, 0, 0, 0), _WDirectionCD=(0, 0, 0, 0)):
        self.mode = mode
        self._Amplitude, self._Frequency, self._Speed = _Amplitude, _Frequency, _Speed
        self._Steepness, self._Smoothing = _Steepness, _Smoothing
        self._WSpeed, self._WDirectionAB, self._WDirectionCD = tuple(_WSpeed), tuple(_WDirectionAB), tuple(_WDirectionCD)

    def _c(self):
        p = nat.MwPondParams()
        p.mode, p.amplitude, p.frequency, p.speed = self.mode, self._Amplitude, self._Frequency, self._Speed
        p.steepness, p.smoothing = self._Steepness, self._Smoothing
        p.wspeed[:], p.dir_ab[:], p.dir_cd[:] = list(self._WSpeed), list(self._WDirectionAB), list(self._WDirectionCD)
        return p

    def displace(self, pos_xyz, t: float, normals: bool = True, device: int = 0):
        """-> (displaced vertices, normals or None) for host vertices [n,3] at _Time.y = t."""
        pos = np.ascontiguousarray(pos_xyz, np.float32)
        out = np.empty_like(pos)
        nrm = np.empty_like(pos) if normals else None
        p = self._c()
        nat.check(nat.lib().mw_pond_displace(C.byref(p), _p(pos), pos.size // 3, C.c_float(t), _p(out),
                                             _p(nrm) if normals else None, device))
        return out, nrm

    def displace_device(self, d_pos: int, nverts: int, t: float, d_out: int, d_normals: int = 0, stream: int = 0):
        """Device pointers (ints), asynchronous on ``stream`` (a hipStream_t as int, 0 = default stream)."""
        p = self._c()
        nat.check(nat.lib().mw_pond_displace_device(C.byref(p), C.c_void_p(d_pos), nverts, C.c_float(t), C.c_void_p(d_out),
                                                    C.c_void_p(d_normals) if d_normals else None,
                                                    C.c_void_p(stream) if stream else None))
