"""us per call of the single-step pond kernels at 1M vertices: gerstner (1 step through the steps entry) and PondMaterial modes."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mistral-water_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch; torch.cuda.is_available()
import mistral_water as mw
import workloads
nv = 1000 * 1000
pos = torch.rand((nv, 3), device="cuda") * 100
out = torch.empty_like(pos); nrm = torch.empty_like(pos)
def timeit(fn, iters=400):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
w = workloads.pond_waves8()
print("gerstner 1 step x 8 waves: %.2f us" % timeit(lambda: mw.gerstner_displace_steps_device(pos.data_ptr(), nv, w, 0.1, 2.58, 0.99, [0.5], out.data_ptr())))
import ctypes as C, numpy as np
from mistral_water import _native as nat
W = np.ascontiguousarray(w, np.float32).reshape(-1, 3)
def single():
    nat.check(nat.lib().mw_gerstner_displace_device(C.c_void_p(pos.data_ptr()), nv, W.ctypes.data_as(C.c_void_p), 8, C.c_float(0.1), C.c_float(2.58),
                                                    C.c_float(0.99), C.c_float(0.5), C.c_void_p(out.data_ptr()), None))
print("gerstner, the single-step entry (mw_gerstner_displace_device) x 8 waves: %.2f us" % timeit(single))
print("again: %.2f us; steps entry again: %.2f us" % (timeit(single), timeit(lambda: mw.gerstner_displace_steps_device(pos.data_ptr(), nv, w, 0.1, 2.58, 0.99, [0.5], out.data_ptr()))))
for mode in (0, 1, 2):
    pm = mw.PondMaterial(mode=mode) if "mode" in mw.PondMaterial.__init__.__code__.co_varnames else mw.PondMaterial()
    try:
        pm.mode = mode
    except Exception:
        pass
    print("pond mode %d + normals: %.2f us" % (mode, timeit(lambda: pm.displace_device(pos.data_ptr(), nv, 0.5, out.data_ptr(), nrm.data_ptr()))))
