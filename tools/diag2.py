import os, sys, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mistral-water_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch; torch.cuda.is_available()
import mistral_water as mw
from mistral_water import _native as nat
from oracle import oracle as O
import workloads, emul_build
L = nat.lib()
x = np.linspace(-200, 2000, 200001).astype(np.float32)
s = np.empty_like(x); c = np.empty_like(x)
L.mw_debug_sincos.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
L.mw_debug_sincos(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data)
print("device sincos err", np.abs(s - np.sin(x.astype(np.float64))).max(), np.abs(c - np.cos(x.astype(np.float64))).max())
L.mw_debug_get_omega.argtypes = [C.c_void_p, C.c_void_p]
for N in (256, 1024):
    p = workloads.fftmesh_params(N)
    with mw.Ocean(resolution=N, unit_width=1.0, length=float(N), wind=(p.wind_x, p.wind_y), amplitude=p.amplitude) as o:
        om = np.empty((N, N), np.float32)
        L.mw_debug_get_omega(o.handle, om.ctypes.data)
        want = emul_build.load().omega_t(p, 1.0)   # [i][j] omega*1
        d = np.abs(om.T - want)
        print(N, "omega table mismatches:", int((d > 0).sum()), "max abs", d.max(), "w0", 2*np.pi/N)
        wt = o.debug_omega_t(1.0)
        print("   debug_omega_t mismatches", int((wt != want).sum()))
