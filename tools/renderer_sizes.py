"""One OceanRenderer GenerateTexture() at resolution 64 .. 512 (512^2 .. 4096^2 textures), device-resident: us per frame and fraction of the 120-B figure."""
import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/mistral-water_amd", "/root/repo/tests"]
import torch; torch.cuda.init()
import ctypes as C
import mistral_water as mw
from mistral_water import _native as nat
for res in (64, 128, 256, 512):
    o = mw.Ocean(resolution=res, length=434.48, wind=(14.45, 12.0), amplitude=0.41, choppiness=0.46, mult=1.5, semantics=nat.MW_SEM_OCEANRENDERER)
    def f(): nat.check(nat.lib().mw_ocean_generate_texture_device(o.handle, C.c_float(1.0 / 60.0), None, None, None, None))
    for _ in range(20): f()
    o.synchronize()
    t0 = time.perf_counter()
    n = 200 if res <= 128 else 40
    for _ in range(n): f()
    o.synchronize()
    dt = (time.perf_counter() - t0) / n
    M = 8 * res
    print(f"resolution {res}: textures {M}^2, {dt * 1e6:.1f} us per frame, {M * M / dt:.3g} texels/s, {120 * M * M / dt / 8e12:.3f} of the 120-B roofline")
    o.close()
