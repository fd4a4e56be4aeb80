import sys, ctypes as C
sys.path.insert(0, 'mistral-water_amd'); sys.path.insert(0, 'tests')
import numpy as np
import mistral_water as mw
a = np.zeros((64, 16, 2), np.float32)
for l in range(64):
    for r in range(16):
        a[l, r, 0] = l * 16 + r
        a[l, r, 1] = -(l * 16 + r)
got = a.copy()
mw.check(mw.lib().mw_debug_wave_transpose4(got.ctypes.data_as(C.c_void_p)))
src = got[:, :, 0].astype(int)
print("lane,rho -> source (lane, rho):")
for l in (0, 1, 15, 16, 17, 32, 33, 48, 63):
    print(l, [(int(s) // 16, int(s) % 16) for s in src[l]])
print("im consistent:", (got[:, :, 1] == -got[:, :, 0]).all())
