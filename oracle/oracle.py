"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

ctypes front-end to oracle/liboracle.so (the C restatement of the reference's CPU path,
see fftmesh_oracle.c for the file:line citations) plus ``eval_fft_f64``: the same f64 model
with numpy's FFT in the middle, usable at N = 1024 / 4096 where the O(N^3)/O(N^4) forms are
too slow.  ``eval_fft_f64`` is validated against the literal forms in tests/test_oracle.py.

PARITY UNPINNED -- see the header of fftmesh_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (make).  MW_SANITIZE=1 selects the AddressSanitizer + UBSan build (liboracle_san.so;
    the process must then run with libasan preloaded -- tests/test_sanitizers.py does that in a subprocess)."""
    if os.environ.get("MW_SANITIZE") == "1":
        so = os.path.join(_HERE, "liboracle_san.so")
        subprocess.run(["make", "-C", _HERE, "liboracle_san.so"], check=True, capture_output=True)
        return so
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return so


class _P(C.Structure):
    _fields_ = [("N", C.c_int32), ("unit_width", C.c_float), ("length", C.c_float), ("wind_x", C.c_float),
                ("wind_y", C.c_float), ("amplitude", C.c_float), ("choppiness", C.c_float), ("gravity", C.c_float)]


@dataclass
class Params:
    """Inspector fields of S/FFTMesh.cs:9-23 (+ gravity, a constant 9.81f at :52)."""
    N: int
    unit_width: float = 1.0
    length: float = 1.0
    wind_x: float = 1.0
    wind_y: float = 1.0
    amplitude: float = 1.0
    choppiness: float = 1.0
    gravity: float = 9.81

    def c(self) -> _P:
        return _P(self.N, self.unit_width, self.length, self.wind_x, self.wind_y, self.amplitude,
                  self.choppiness, self.gravity)

    @property
    def commensurate(self) -> bool:
        return np.float32(self.unit_width) * np.float32(self.N) == np.float32(self.length)


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_uniform.restype = C.c_float
        _LIB.orc_uniform.argtypes = [C.c_uint64, C.c_uint64]
        _LIB.orc_dispersion.restype = C.c_float
        _LIB.orc_dispersion.argtypes = [C.POINTER(_P), C.c_int, C.c_int]
        _LIB.orc_phillips.restype = C.c_float
        _LIB.orc_phillips.argtypes = [C.POINTER(_P), C.c_int, C.c_int]
        _LIB.orc_rest_mesh.restype = C.c_int
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def uniform(seed: int, counter: int) -> float:
    return lib().orc_uniform(seed, counter)


def dispersion(p: Params, n: int, m: int) -> float:
    return lib().orc_dispersion(C.byref(p.c()), n, m)


def dispersion_grid(p: Params, t: float):
    """omega(n, m) * t over the whole grid [N, N] in the reference's strict float32 sequence (S/FFTMesh.cs:141-147, :183)."""
    out = np.empty((p.N, p.N), np.float32)
    lib().orc_dispersion_grid.restype = None
    lib().orc_dispersion_grid.argtypes = [C.POINTER(_P), C.c_float, C.c_void_p]
    lib().orc_dispersion_grid(C.byref(p.c()), C.c_float(t), _fp(out))
    return out


def phillips(p: Params, n: int, m: int) -> float:
    return lib().orc_phillips(C.byref(p.c()), n, m)


def generate_spectrum(p: Params, seed: int):
    h0 = np.empty((p.N, p.N, 2), np.float32)
    h0c = np.empty((p.N, p.N, 2), np.float32)
    lib().orc_generate_spectrum(C.byref(p.c()), C.c_uint64(seed), _fp(h0), _fp(h0c))
    return h0, h0c


def rest_mesh(p: Params):
    N = p.N
    v = np.empty((N * N, 3), np.float32)
    n = np.empty((N * N, 3), np.float32)
    uv = np.empty((N * N, 2), np.float32)
    idx = np.empty(((N - 1) * (N - 1) * 6,), np.int32)
    cnt = lib().orc_rest_mesh(C.byref(p.c()), _fp(v), _fp(n), _fp(uv), _fp(idx))
    assert cnt == idx.size
    return v, n, uv, idx


def eval_literal_f32(p: Params, h0, h0c, t: float):
    """EvaluateWaves(t) exactly as S/FFTMesh.cs:224-280, strict float32, O(N^4)."""
    N = p.N
    v = np.empty((N * N, 3), np.float32)
    n = np.empty((N * N, 3), np.float32)
    c = np.empty((N * N, 4), np.float32)
    lib().orc_eval_literal_f32(C.byref(p.c()), _fp(np.ascontiguousarray(h0, np.float32)),
                               _fp(np.ascontiguousarray(h0c, np.float32)), C.c_float(t), _fp(v), _fp(n), _fp(c))
    return v, n, c


def displacement_subset_f32(p: Params, h0, h0c, t: float, vertex_idx):
    vi = np.ascontiguousarray(vertex_idx, np.int32)
    hd = np.empty((vi.size, 3), np.float32)
    nor = np.empty((vi.size, 3), np.float32)
    lib().orc_displacement_subset_f32(C.byref(p.c()), _fp(np.ascontiguousarray(h0, np.float32)),
                                      _fp(np.ascontiguousarray(h0c, np.float32)), C.c_float(t), _fp(vi),
                                      C.c_int(vi.size), _fp(hd), _fp(nor))
    return hd, nor


def eval_f64(p: Params, h0, h0c, t: float):
    """f64 separable O(N^3) evaluation; any unit_width/length."""
    N = p.N
    v = np.empty((N * N, 3), np.float64)
    n = np.empty((N * N, 3), np.float64)
    c = np.empty((N * N, 4), np.float64)
    lib().orc_eval_f64(C.byref(p.c()), _fp(np.ascontiguousarray(h0, np.float32)),
                       _fp(np.ascontiguousarray(h0c, np.float32)), C.c_float(t), _fp(v), _fp(n), _fp(c))
    return v, n, c


def htilde_fields_f64(p: Params, h0, h0c, t: float):
    f = np.empty((5, p.N, p.N, 2), np.float64)
    lib().orc_htilde_fields_f64(C.byref(p.c()), _fp(np.ascontiguousarray(h0, np.float32)),
                                _fp(np.ascontiguousarray(h0c, np.float32)), C.c_float(t), _fp(f))
    return f[..., 0] + 1j * f[..., 1]


def assemble_f64(p: Params, spatial):
    """spatial: complex128 [5,N,N] (H,Dx,Dz,Sx,Sz) -> vertices, normals, colours, hds (f64)."""
    N = p.N
    s = np.empty((5, N, N, 2), np.float64)
    s[..., 0] = spatial.real
    s[..., 1] = spatial.imag
    v = np.empty((N * N, 3), np.float64)
    n = np.empty((N * N, 3), np.float64)
    c = np.empty((N * N, 4), np.float64)
    hds = np.empty((N * N, 2), np.float64)
    lib().orc_assemble_f64(C.byref(p.c()), _fp(s), _fp(v), _fp(n), _fp(c), _fp(hds))
    return v, n, c, hds


def transform_fft_f64(p: Params, fields):
    """The reference's direct sum  sum_ij F(i,j) e^{i(kx_i x_a + kz_j z_b)}  (S/FFTMesh.cs:199-217)
    restated as N^2 * ifft2 with separable pre/post twiddles.  Exact iff the grid is commensurate
    (unit_width == length/N) and N is even:
        k_i x_a = 2 pi (i - N/2)(a - N/2 + 1/2) / N
                = 2 pi i a / N  +  pi i (1/N - 1)  -  pi a  +  pi (N/2 - 1/2)          (mod 2 pi)
    """
    N = p.N
    assert p.commensurate and N % 2 == 0
    i = np.arange(N)
    pre = np.exp(1j * np.pi * i * (1.0 / N - 1.0))           # (-1)^i e^{i pi i/N}
    post = np.exp(1j * np.pi * (N / 2.0 - 0.5 - i))          # (-1)^a e^{i pi (N-1)/2}
    x = fields * pre[None, :, None] * pre[None, None, :]
    y = np.fft.ifft2(x, axes=(1, 2)) * (N * N)
    return y * post[None, :, None] * post[None, None, :]


def transform_matmul_f64(p: Params, fields):
    """The reference's direct sum  sum_ij F(i,j) e^{i(kx_i x_a + kz_j z_b)}  (S/FFTMesh.cs:199-217) for ANY grid (odd N, N not a
    power of two, unit_width != length / N) as two dense products in complex128: E^T (F E) with E[j][b] = e^{i k_j pos_b},
    k_j = 2 pi (j - N/2) / length (:201,204), pos_b the rest coordinate of grid line b (:107-112).  The same arithmetic as
    orc_transform_direct_f64, through BLAS: the checker for large non-FFT grids (N = 1000 in seconds instead of minutes)."""
    N = p.N
    j = np.arange(N, dtype=np.float64)
    k = 2.0 * np.pi * (j - N / 2.0) / float(np.float32(p.length))
    pos = (j - N // 2) * float(np.float32(p.unit_width)) + (float(np.float32(p.unit_width)) / 2.0 if N % 2 == 0 else 0.0)
    E = np.exp(1j * np.outer(k, pos))
    return np.stack([E.T @ (f @ E) for f in fields])


def eval_matmul_f64(p: Params, h0, h0c, t: float, return_hds: bool = False):
    """f64 evaluation of any grid through transform_matmul_f64 (checked against orc_eval_f64 in tests/test_oracle.py)."""
    v, n, c, hds = assemble_f64(p, transform_matmul_f64(p, htilde_fields_f64(p, h0, h0c, t)))
    return (v, n, c, hds) if return_hds else (v, n, c)


def eval_fft_f64(p: Params, h0, h0c, t: float, return_hds: bool = False):
    """f64 evaluation with numpy's FFT (commensurate grids only) -- the large-N checker."""
    spatial = transform_fft_f64(p, htilde_fields_f64(p, h0, h0c, t))
    v, n, c, hds = assemble_f64(p, spatial)
    return (v, n, c, hds) if return_hds else (v, n, c)


def whitecap_f32(N: int, hds, normals):
    c = np.empty((N * N, 4), np.float32)
    lib().orc_whitecap_f32(C.c_int32(N), _fp(np.ascontiguousarray(hds, np.float32)),
                           _fp(np.ascontiguousarray(normals, np.float32)), _fp(c))
    return c


# ======================================= OceanRenderer semantics ========================================
class _RP(C.Structure):
    _fields_ = [("resolution", C.c_int32), ("length", C.c_float), ("wind_x", C.c_float), ("wind_y", C.c_float),
                ("amplitude", C.c_float), ("choppiness", C.c_float), ("gravity", C.c_float), ("mult", C.c_float)]


@dataclass
class RendererParams:
    """Inspector fields of S/OceanRenderer.cs:10-19; textures are (8*resolution)^2."""
    resolution: int
    length: float = 256.0
    wind_x: float = 0.0
    wind_y: float = 0.0
    amplitude: float = 1.0
    choppiness: float = 1.5
    gravity: float = 9.81
    mult: float = 2.0

    @property
    def M(self):
        return 8 * self.resolution

    def c(self):
        return _RP(self.resolution, self.length, self.wind_x, self.wind_y, self.amplitude, self.choppiness, self.gravity,
                   self.mult)


def renderer_initial_spectrum(p: RendererParams, seed: int):
    """F/InitialSpectrum.shader -> [M, M, 4] = (h0.xy, conj(h0').xy), texel (px,py) at [py, px]."""
    init4 = np.empty((p.M, p.M, 4), np.float32)
    lib().orr_initial_spectrum(C.byref(p.c()), C.c_uint64(seed), _fp(init4))
    return init4


def renderer_step_f64(p: RendererParams, init4, phase, delta_time: float, literal_passes: bool = True, normal_params=None):
    """One OceanRenderer.GenerateTexture().  `phase` ([M,M] float32) is advanced in place.
    literal_passes=True runs the 2*log2(M) Stockham gather passes exactly as S/OceanRenderer.cs schedules them;
    False swaps in numpy's fft2 (validated equal in tests) for large M.
    normal_params: the parameters normalMat / whiteMat carry (S/OceanRenderer.cs sets normalMat's _Length only in SetParams,
    :163, so after a length change the normal pass keeps the OLD length); needs literal_passes=False."""
    assert normal_params is None or not literal_passes
    M = p.M
    init4 = np.ascontiguousarray(init4, np.float32)
    assert phase.dtype == np.float32 and phase.flags.c_contiguous
    h = np.empty((M, M), np.float64)
    d = np.empty((M, M, 2), np.float64)
    n = np.empty((M, M, 3), np.float64)
    w = np.empty((M, M), np.float64)
    g = np.empty((M, M), np.float64)
    if literal_passes:
        lib().orr_step_f64(C.byref(p.c()), _fp(init4), _fp(phase), C.c_float(delta_time), _fp(h), _fp(d), _fp(n), _fp(w), _fp(g))
        return h, d, n, w, g
    sd = np.empty((M, M, 4), np.float64)
    sh = np.empty((M, M, 2), np.float64)
    lib().orr_spectra_f64(C.byref(p.c()), _fp(init4), _fp(phase), C.c_float(delta_time), _fp(sd), _fp(sh))
    hx = np.fft.fft2(sd[..., 0] + 1j * sd[..., 1])
    hz = np.fft.fft2(sd[..., 2] + 1j * sd[..., 3])
    hh = np.fft.fft2(sh[..., 0] + 1j * sh[..., 1])
    dtex = np.ascontiguousarray(np.stack([hx.real, hx.imag, hz.real, hz.imag], -1))
    hre = np.ascontiguousarray(hh.real)
    lib().orr_normal_white_f64(C.byref((normal_params or p).c()), _fp(dtex), _fp(hre), _fp(n), _fp(w))
    return hre, np.ascontiguousarray(dtex[..., [0, 2]]), n, w, np.ascontiguousarray(dtex[..., 1])


def renderer_advance_phase(p: RendererParams, init4, phase, delta_time: float):
    """The Dispersion pass alone (F/Dispersion.shader:32-41): `phase` advanced in place by one frame of delta_time, no textures
    (the spectra orr_spectra_f64 also produces are dropped) -- how a test walks the oracle to frame k of a long chain."""
    M = p.M
    init4 = np.ascontiguousarray(init4, np.float32)
    assert phase.dtype == np.float32 and phase.flags.c_contiguous
    sd = np.empty((M, M, 4), np.float64)
    sh = np.empty((M, M, 2), np.float64)
    lib().orr_spectra_f64(C.byref(p.c()), _fp(init4), _fp(phase), C.c_float(delta_time), _fp(sd), _fp(sh))


def renderer_textures_f64(p: RendererParams, init4, phase, delta_time: float):
    """One GenerateTexture() as the four ARGBFloat render targets of S/OceanRenderer.cs:143-146 ([M,M,4] each):
    height (Re h, Im h, Re h, Im h), displacement (Re Dx, Im Dx, Re Dz, Im Dz), normal (n, 1), white (w, w, w, 1).
    Transform by numpy's fft2 (the literal pass schedule is validated equal to it in tests/test_ocean_renderer.py)."""
    M = p.M
    init4 = np.ascontiguousarray(init4, np.float32)
    sd = np.empty((M, M, 4), np.float64)
    sh = np.empty((M, M, 2), np.float64)
    lib().orr_spectra_f64(C.byref(p.c()), _fp(init4), _fp(phase), C.c_float(delta_time), _fp(sd), _fp(sh))
    hx = np.fft.fft2(sd[..., 0] + 1j * sd[..., 1])
    hz = np.fft.fft2(sd[..., 2] + 1j * sd[..., 3])
    hh = np.fft.fft2(sh[..., 0] + 1j * sh[..., 1])
    dtex = np.ascontiguousarray(np.stack([hx.real, hx.imag, hz.real, hz.imag], -1))
    hre = np.ascontiguousarray(hh.real)
    n = np.empty((M, M, 3), np.float64)
    w = np.empty((M, M), np.float64)
    lib().orr_normal_white_f64(C.byref(p.c()), _fp(dtex), _fp(hre), _fp(n), _fp(w))
    htex = np.stack([hh.real, hh.imag, hh.real, hh.imag], -1)
    ntex = np.concatenate([n, np.ones((M, M, 1))], -1)
    wtex = np.stack([w, w, w, np.ones_like(w)], -1)
    return htex, dtex, ntex, wtex


def renderer_mesh_vertex_stage_f64(p: RendererParams, unit_width, height, disp_rb, normal, white):
    """W/TestOcean.shader:61-79 on the res x res mesh of S/OceanRenderer.cs:172-207 -> (vertices, normals, colors)."""
    M, res = p.M, p.resolution
    height = np.ascontiguousarray(height, np.float64)
    disp_rb = np.ascontiguousarray(disp_rb, np.float64)
    normal = np.ascontiguousarray(normal, np.float64)
    white = np.ascontiguousarray(white, np.float64)
    v = np.empty((res * res, 3), np.float64)
    n = np.empty((res * res, 3), np.float64)
    c = np.empty(res * res, np.float64)
    lib().orr_mesh_vertex_stage_f64(C.c_int(M), C.c_int(res), C.c_float(unit_width), _fp(height), _fp(disp_rb), _fp(normal),
                                    _fp(white), _fp(v), _fp(n), _fp(c))
    return v, n, c


def gerstner_f64(pos_xyz, waves, amplitude, frequency, steepness, t):
    """W/MistralWaterLib.cginc:71-99,154-180 in f64 (oracle/gerstner_oracle.c)."""
    pos = np.ascontiguousarray(pos_xyz, np.float32)
    wv = np.ascontiguousarray(waves, np.float32).reshape(-1, 3)
    out = np.empty(pos.shape, np.float64)
    lib().orc_gerstner_f64(_fp(pos), C.c_int64(pos.size // 3), _fp(wv), C.c_int(wv.shape[0]), C.c_float(amplitude),
                           C.c_float(frequency), C.c_float(steepness), C.c_float(t), _fp(out))
    return out


def gerstner_f32_range(pos_xyz, v0, v1, waves, amplitude, frequency, steepness, t, out):
    """CPU_GERSTNER baseline: the shader's float32 arithmetic on vertices [v0, v1); releases the GIL."""
    wv = np.ascontiguousarray(waves, np.float32).reshape(-1, 3)
    lib().orc_gerstner_f32_range(_fp(pos_xyz), C.c_int64(v0), C.c_int64(v1), _fp(wv), C.c_int(wv.shape[0]), C.c_float(amplitude),
                                 C.c_float(frequency), C.c_float(steepness), C.c_float(t), _fp(out))


class PondParams(C.Structure):
    """Material properties of the pond shader (W/MistralWaterLib.cginc:53-66); layout of mw_pond_params."""
    _fields_ = [("mode", C.c_int32), ("amplitude", C.c_float), ("frequency", C.c_float), ("speed", C.c_float),
                ("steepness", C.c_float), ("smoothing", C.c_float), ("wspeed", C.c_float * 4), ("dir_ab", C.c_float * 4),
                ("dir_cd", C.c_float * 4)]


def pond_params(mode, amplitude, frequency, speed=0.0, steepness=0.0, smoothing=1.0, wspeed=(0, 0, 0, 0),
                dir_ab=(0, 0, 0, 0), dir_cd=(0, 0, 0, 0)):
    p = PondParams()
    p.mode, p.amplitude, p.frequency, p.speed, p.steepness, p.smoothing = mode, amplitude, frequency, speed, steepness, smoothing
    p.wspeed[:], p.dir_ab[:], p.dir_cd[:] = list(wspeed), list(dir_ab), list(dir_cd)
    return p


def pond_displace_f64(p, pos_xyz, t):
    """W/MistralWaterLib.cginc:154-180 Displacement() in f64 (oracle/pond_oracle.c) -> (positions, normals)."""
    pos = np.ascontiguousarray(pos_xyz, np.float32)
    out = np.empty(pos.shape, np.float64)
    nrm = np.empty(pos.shape, np.float64)
    rc = lib().orc_pond_displace_f64(C.byref(p), _fp(pos), C.c_int64(pos.size // 3), C.c_float(t), _fp(out), _fp(nrm))
    if rc:
        raise ValueError("unknown pond displacement mode")
    return out, nrm


def cpu_fft_step_f32(p: Params, h0, h0c, t: float, nthreads: int = 1):
    """The CPU_FFT baseline of SURVEY.md 8d (oracle/cpu_fft_baseline.c): float32 radix-2 Stockham 2-D transform over
    `nthreads` pthreads -> (vertices, normals, colors) float32.  Commensurate power-of-two grids only."""
    N = p.N
    h0 = np.ascontiguousarray(h0, np.float32)
    h0c = np.ascontiguousarray(h0c, np.float32)
    v = np.empty((N * N, 3), np.float32)
    n = np.empty((N * N, 3), np.float32)
    c = np.empty((N * N, 4), np.float32)
    rc = lib().orc_cpu_fft_step_f32(C.byref(p.c()), _fp(h0), _fp(h0c), C.c_float(t), C.c_int(nthreads), _fp(v), _fp(n), _fp(c))
    if rc:
        raise ValueError({1: "N must be a power of two >= 8", 2: "grid is not commensurate", 3: "out of memory"}[rc])
    return v, n, c
