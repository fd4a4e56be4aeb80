"""Cycle stamps of k_pass1_wave's phases (build with -DMW_TIMING; MW_LIB=variants/<name>.so): per sampled workgroup and wave."""
import os, sys, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mistral-water_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch; torch.cuda.is_available()
import mistral_water as mw
from mistral_water import _native as nat
import workloads
N, B = 4096, 8
p = workloads.fftmesh_config2(N)
o = mw.Ocean(resolution=N, unit_width=1.0, length=float(N), wind=(p.wind_x, p.wind_y), amplitude=p.amplitude)
NN = N * N
dv = torch.empty((B, NN, 3), device="cuda"); dn = torch.empty((B, NN, 3), device="cuda"); dw = torch.empty((B, NN), device="cuda")
for _ in range(4):
    o.evaluate_device([0.1 * k for k in range(B)], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
o.synchronize()
st = np.zeros((2, 64, 8, 32), np.int64)
L = nat.lib()
L.mw_debug_get_stamps.argtypes = [C.c_void_p]
assert L.mw_debug_get_stamps(st.ctypes.data) == 0
names = ["field start", "loads + animate", "dft64", "barrier D", "local exchange (re, im)", "twiddle + dft64", "half0 out + barrier", "store0, barrier, half1 out, barrier, store1"]
s = st[0]
tot = {n: [] for n in names}
shown = 0
for slot in range(64):
    if s[slot, 0, 0] == 0: continue
    for wv in range(4):
        r = s[slot, wv]
        line = [f"wg{slot} w{wv} total {r[26] - r[0]:7d}:"]
        for f in range(3):
            if r[1 + 8 * f] == 0: continue
            prev = r[1 + 8 * f]
            seg = []
            for i, nm in enumerate(names[1:], start=2):
                seg.append(int(r[i + 8 * f] - prev)); tot[nm].append(seg[-1]); prev = r[i + 8 * f]
            line.append(f"f{f} " + "/".join(str(x) for x in seg))
        if shown < 24: print(" ".join(line)); shown += 1
print("mean cycles per phase (counter ticks at 100 MHz if the values look 24x too small):")
for nm in names[1:]:
    if tot[nm]: print(f"  {nm:45s} {np.mean(tot[nm]):9.0f}  min {np.min(tot[nm]):7d} max {np.max(tot[nm]):7d}")
