import sys, os
sys.path.insert(0, "/root/repo/mistral-water_amd"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/mistral-water_amd")
import torch, numpy as np, ctypes as C
import mistral_water as mw
L = mw.lib()
rng = np.random.default_rng(0)
for scale in (3.0, 100.0, 500.0, 5000.0, 1e5, 1e6):
    x = rng.uniform(-scale, scale, 1 << 20).astype(np.float32)
    for name in ("mw_debug_sincos", "mw_debug_sincos_fast"):
        s = np.empty_like(x); c = np.empty_like(x)
        getattr(L, name)(x.ctypes.data_as(C.c_void_p), x.size, s.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        es = np.abs(s - np.sin(x.astype(np.float64))).max(); ec = np.abs(c - np.cos(x.astype(np.float64))).max()
        print(f"|x|<{scale:7.0f} {name:22s} max|ds| {es:.2e} max|dc| {ec:.2e}")
