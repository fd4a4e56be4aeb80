"""OceanRenderer frames per enqueue (mw_ocean_generate_texture_steps_device): us per frame against n, device-resident.
usage: [MW_LIB=variants/X.so] python tools/renderer_steps_probe.py [resolution ...]"""
import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/mistral-water_amd", "/root/repo/tests"]
import torch; torch.cuda.init()
import ctypes as C
import numpy as np
import mistral_water as mw
from mistral_water import _native as nat
print("build", nat.build_id())
import os
NS = [int(x) for x in os.environ.get("PROBE_N", "0,1,2,4,8,16,32").split(",")]
for res in [int(x) for x in sys.argv[1:]] or [128]:
    M = 8 * res
    o = mw.Ocean(resolution=res, length=434.48 * M / 1024, wind=(14.45, 12.0), amplitude=0.41, choppiness=0.46, mult=1.5, semantics=nat.MW_SEM_OCEANRENDERER)
    def single(): nat.check(nat.lib().mw_ocean_generate_texture_device(o.handle, C.c_float(1.0 / 60.0), None, None, None, None))
    row = []
    for n in NS:
        if n * M * M * 60 > 40e9:
            continue
        dts = np.full(max(n, 1), 1.0 / 60.0, np.float32)
        f = single if n == 0 else (lambda: o.generate_texture_steps_device(dts))
        reps = max(3, int(2000 / max(n, 1) / (M / 1024) ** 2))
        for _ in range(max(2, reps // 4)): f()
        o.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): f()
            o.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps / max(n, 1))
        row.append(f"n={n if n else 'single'}: {best * 1e6:.2f} us/frame ({120 * M * M / best / 8e12:.3f})")
    print(f"textures {M}^2: " + "; ".join(row))
    o.close()
