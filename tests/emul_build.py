"""Builds and loads tests/emul/libemul.so (g++; strict float32, no FMA contraction)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "emul_fftmesh.cpp")
SO = os.path.join(HERE, "emul", "libemul.so")
CSRC = os.path.join(os.path.dirname(HERE), "mistral-water_amd", "csrc")


def build(defs=()):
    """defs: extra -D macros (a differently configured build of the same kernels, e.g. ("MW_SPLIT_SLOPES=1",)) -> its own .so"""
    if defs:
        so = os.path.join(HERE, "emul", "libemul_" + "_".join(d.replace("=", "") for d in defs) + ".so")
        deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
        if not (os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps)):
            subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + ["-D" + d for d in defs] + ["-o", so, SRC],
                           check=True)
        return so
    if os.environ.get("MW_SANITIZE") == "1":   # AddressSanitizer + UBSan build (tests/test_sanitizers.py, libasan preloaded)
        so = os.path.join(HERE, "emul", "libemul_san.so")
        deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
        if not (os.path.exists(so) and all(os.path.getmtime(d) <= os.path.getmtime(so) for d in deps)):
            subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fno-omit-frame-pointer",
                            "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", so, SRC], check=True)
        return so
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, SRC], check=True)
    return SO


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Emul:
    def __init__(self, defs=()):
        self.L = C.CDLL(build(defs))

    def set_variant(self, force_hs=False, frame=False):
        """force_hs: run the sequential-halo pass 2 (the product's N >= 4096 kernel) at every grid size.
        frame: run the single-step plan's pass 2 (k_pass2_frame: the three fields of a row block side by side)."""
        self.L.emul_set_variant(2 if frame else (1 if force_hs else 0))

    def evaluate(self, p, h0, h0c, times, white_stride=4, pts=0):
        """p: oracle.Params.  Returns (vertices, normals, white) with a leading step axis."""
        N, ns = p.N, len(times)
        v = np.empty((ns, N * N, 3), np.float32)
        n = np.empty((ns, N * N, 3), np.float32)
        w = np.empty((ns, N * N, white_stride), np.float32)
        tt = np.asarray(times, np.float32)
        r = self.L.emul_fftmesh_evaluate(N, pts, C.c_float(p.unit_width), C.c_float(p.length), C.c_float(p.gravity),
                                         C.c_float(p.choppiness), _p(np.ascontiguousarray(h0, np.float32)),
                                         _p(np.ascontiguousarray(h0c, np.float32)), _p(tt), ns, _p(v), _p(n), _p(w),
                                         white_stride)
        assert r == 0, f"emul_fftmesh_evaluate -> {r}"
        return v, n, w

    def fft1d(self, x, pts=16):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        assert self.L.emul_fft1d(x.shape[0], pts, _p(x), _p(y)) == 0
        return y

    def rest_mesh(self, N, unit_width):
        v = np.empty((N * N, 3), np.float32)
        n = np.empty((N * N, 3), np.float32)
        uv = np.empty((N * N, 2), np.float32)
        idx = np.full(((N - 1) * (N - 1) * 6,), -1, np.int32)
        self.L.emul_rest_mesh(N, C.c_float(unit_width), _p(v), _p(n), _p(uv), _p(idx))
        return v, n, uv, idx

    def spectrum(self, p, seed):
        h0 = np.empty((p.N, p.N, 2), np.float32)
        h0c = np.empty((p.N, p.N, 2), np.float32)
        self.L.emul_spectrum(p.N, C.c_float(p.length), C.c_float(p.wind_x), C.c_float(p.wind_y), C.c_float(p.amplitude),
                             C.c_float(p.gravity), C.c_uint64(seed), _p(h0), _p(h0c))
        return h0, h0c

    def omega_t(self, p, t):
        out = np.empty((p.N, p.N), np.float32)
        self.L.emul_omega_t(p.N, C.c_float(p.length), C.c_float(p.gravity), C.c_float(t), _p(out))
        return out

    def or_init(self, rp, seed):
        M = rp.M
        initT = np.empty((M, M, 4), np.float32)   # [px][py]
        phaseT = np.empty((M, M), np.float32)
        self.L.emul_or_init(M, C.c_float(rp.length), C.c_float(rp.wind_x), C.c_float(rp.wind_y), C.c_float(rp.amplitude),
                            C.c_float(rp.gravity), C.c_uint64(seed), _p(initT), _p(phaseT))
        return initT, phaseT

    def or_step(self, rp, initT, phaseT, delta_time, imag=False, packed=False):
        """imag=True also returns heightTexture.g (Im h) and displacementTexture.a (Im Dz) as two extra arrays.
        packed=True: the two-transform plan of planar-texture calls (needs a mirror-symmetric phase; not with imag)."""
        assert not (imag and packed)
        M = rp.M
        hg = np.empty((M, M), np.float32) if imag else None
        da = np.empty((M, M), np.float32) if imag else None
        h = np.empty((M, M), np.float32)
        d = np.empty((M, M, 2), np.float32)
        g = np.empty((M, M), np.float32)
        n = np.empty((M, M, 3), np.float32)
        w = np.empty((M, M), np.float32)
        dt = np.float32(delta_time) * np.float32(rp.mult)
        r = self.L.emul_or_step(M, C.c_float(rp.length), C.c_float(rp.gravity), C.c_float(rp.choppiness), C.c_float(dt),
                                _p(initT), _p(phaseT), _p(h), _p(d), _p(g), _p(n), _p(w),
                                _p(hg) if imag else None, _p(da) if imag else None, 1 if packed else 0)
        assert r == 0
        return (h, d, n, w, g, hg, da) if imag else (h, d, n, w, g)

    def or_pack_rgba(self, h, hg, d, g, da, n, w):
        M = h.shape[0]
        t = [np.empty((M, M, 4), np.float32) for _ in range(4)]
        self.L.emul_or_pack_rgba(M, _p(h), _p(hg), _p(d), _p(g), _p(da), _p(n), _p(w), *[_p(a) for a in t])
        return tuple(t)

    def or_displace_mesh(self, M, res, unit_width, h, d, n, w):
        v, nr, c = np.empty((res * res, 3), np.float32), np.empty((res * res, 3), np.float32), np.empty(res * res, np.float32)
        self.L.emul_or_displace_mesh(M, res, C.c_float(unit_width), _p(h), _p(d), _p(n), _p(w), _p(v), _p(nr), _p(c))
        return v, nr, c

    def czt2d(self, N, unit_width, length, fields):
        """fields: complex [nf, N, N] -> the separable sum of every field through the two chirp-z launches (czt_kernels.h)."""
        fields = np.asarray(fields)
        fin = np.ascontiguousarray(np.stack([fields.real, fields.imag], -1), np.float32)
        out = np.empty_like(fin)
        r = self.L.emul_czt2d(int(N), C.c_float(unit_width), C.c_float(length), fin.shape[0], _p(fin), _p(out))
        assert r == 0, f"emul_czt2d -> {r}"
        return out[..., 0] + 1j * out[..., 1], fin[..., 0].astype(np.float64) + 1j * fin[..., 1]

    def czt_packed(self, p, h0, h0c, t):
        """The product's chirp-z step: three Hermitian-packed planes formed from (h0, h0conj, t) on the index set [0, N]^2, two
        launches -> complex [3, N, N] = (H + i Dx, Sx + i Sz, Dz + i 0)."""
        out = np.empty((3, p.N, p.N, 2), np.float32)
        r = self.L.emul_czt_packed(int(p.N), C.c_float(p.unit_width), C.c_float(p.length), C.c_float(p.gravity),
                                   _p(np.ascontiguousarray(h0, np.float32)), _p(np.ascontiguousarray(h0c, np.float32)), C.c_float(t), _p(out))
        assert r == 0, f"emul_czt_packed -> {r}"
        return out[..., 0] + 1j * out[..., 1]

    def czt_tables_vs_inline(self, p, h0, h0c, t):
        """Elements of the three packed planes on [0, N]^2 whose tabulated form (omega and wave numbers from k_czt_tables' tables: what the
        launches run) differs in any bit from the form with everything computed in place (czt_packed_value)."""
        return int(self.L.emul_czt_tables_vs_inline(int(p.N), C.c_float(p.length), C.c_float(p.gravity),
                                                    _p(np.ascontiguousarray(h0, np.float32)), _p(np.ascontiguousarray(h0c, np.float32)), C.c_float(t)))

    def gerstner_steps(self, pos, waves, amplitude, frequency, steepness, times):
        pos = np.ascontiguousarray(pos, np.float32)
        wv = np.ascontiguousarray(waves, np.float32).reshape(-1, 3)
        tt = np.ascontiguousarray(times, np.float32)
        out = np.empty((tt.size,) + pos.shape, np.float32)
        self.L.emul_gerstner_steps(_p(pos), C.c_long(pos.size // 3), _p(wv), wv.shape[0], C.c_float(amplitude),
                                   C.c_float(frequency), C.c_float(steepness), _p(tt), tt.size, _p(out))
        return out

    def pond(self, params, pos, t):
        """params: a ctypes struct with the mw_pond_params layout (oracle.PondParams) -> (positions, normals) f32."""
        pos = np.ascontiguousarray(pos, np.float32)
        out, nrm = np.empty_like(pos), np.empty_like(pos)
        self.L.emul_pond(C.byref(params), _p(pos), C.c_long(pos.size // 3), C.c_float(t), _p(out), _p(nrm))
        return out, nrm

    def p1_block_map(self, gx, nsteps, tgroup):
        """-> list of (jb, step) or None (padding block) for every block of the 1-D pass-1 grid."""
        nb = self.L.emul_p1_grid_blocks(gx, nsteps, tgroup)
        jb, st = C.c_int(), C.c_int()
        out = []
        for b in range(nb):
            ok = self.L.emul_p1_block_map(b, gx, nsteps, tgroup, C.byref(jb), C.byref(st))
            out.append((jb.value, st.value) if ok else None)
        return out

    def gerstner(self, pos, waves, amplitude, frequency, steepness, t):
        pos = np.ascontiguousarray(pos, np.float32)
        wv = np.ascontiguousarray(waves, np.float32).reshape(-1, 3)
        out = np.empty_like(pos)
        self.L.emul_gerstner(_p(pos), C.c_long(pos.size // 3), _p(wv), wv.shape[0], C.c_float(amplitude),
                             C.c_float(frequency), C.c_float(steepness), C.c_float(t), _p(out))
        return out


_E = None


def load():
    global _E
    if _E is None:
        _E = Emul()
    return _E
