"""Per-phase cycle stamps of one step's workgroups (needs the -DMW_TIMING build: MW_LIB=.../libmw_timing.so)."""
import os, sys, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mistral-water_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch; torch.cuda.is_available()
import mistral_water as mw
from mistral_water import _native as nat
import workloads
N = int(os.environ.get('N', '1024')); B = int(os.environ.get('B', '8'))
p = workloads.fftmesh_params(N)
o = mw.Ocean(resolution=N, unit_width=1.0, length=float(N), wind=(p.wind_x, p.wind_y), amplitude=p.amplitude)
NN = N * N
dv = torch.empty((B, NN, 3), device="cuda"); dn = torch.empty((B, NN, 3), device="cuda"); dw = torch.empty((B, NN), device="cuda")
for _ in range(5):
    o.evaluate_device([0.1 * k for k in range(B)], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
o.synchronize()
st = np.zeros((2, 64, 16, 32), np.int64)
L = nat.lib()
L.mw_debug_get_stamps.argtypes = [C.c_void_p]
assert L.mw_debug_get_stamps(st.ctypes.data) == 0
names1 = {0: "start", 1: "animate (global loads + sincos)"}
for f in range(3):
    names1.update({2 + 8 * f: f"f{f} build + WAR barrier", 3 + 8 * f: f"f{f} dftP + lds write", 4 + 8 * f: f"f{f} s=1: barrier + lds read",
                   5 + 8 * f: f"f{f} s=1: WAR barrier", 6 + 8 * f: f"f{f} s=1: twiddles + dftP + lds write", 7 + 8 * f: f"f{f} rest of the middle passes",
                   8 + 8 * f: f"f{f} lds read + final + global store"})
names1[26] = "end"
names2 = {0: "start"}
for k in range(3):
    names2.update({1 + 8 * k: f"k{k} WAR barrier", 2 + 8 * k: f"k{k} global load + dftP + lds write",
                   6 + 8 * k: f"k{k} middle passes", 7 + 8 * k: f"k{k} lds read + final + field epilogue"})
names2.update({25: "barrier", 26: "publish hds + barrier", 27: "epilogue + stores"})
if os.environ.get("HS", "1" if N >= 4096 else "0") == "1":     # sequential-halo kernel: fields h, d, [vertices, publish+J, halo, J last row], s
    names2 = {0: "start"}
    for k in range(3):
        names2.update({1 + 8 * k: f"k{k} barrier (WAR / tables)", 2 + 8 * k: f"k{k} global load + dftP + lds write",
                       6 + 8 * k: f"k{k} middle passes", 7 + 8 * k: f"k{k} lds read + final pass (+ normals/white stores for k2)"})
    names2.update({24: "vertex stores", 25: "publish hds + jacobian rows 0..R2-2 (3 barriers)", 26: "halo row load + transform + publish",
                   27: "jacobian of the last row"})
    # time order of the ids in this kernel
    order2 = [0, 1, 2, 6, 7, 9, 10, 14, 15, 24, 25, 26, 27, 17, 18, 22, 23]
else:
    order2 = None
for K, names, nw in ((0, names1, 4), (1, names2, 5)):
    print(f"=== kernel {K}: mean cycles per phase over sampled workgroups (wave 0), and total")
    s = st[K][:, 0, :]
    ok = s[:, 0] > 0
    s = s[ok]
    if not ok.any():
        print("   no workgroup of this kernel was sampled"); continue
    print('   waves sampled: first-wave stamps only; B =', B)
    ids = sorted(names) if (K == 0 or order2 is None) else order2
    prev = s[:, ids[0]]
    tot = 0
    for i in ids[1:]:
        cur = s[:, i]
        d = (cur - prev)
        print(f"  {names[i]:42s} {d.mean():9.0f}  (min {d.min():7d} max {d.max():7d})")
        tot += d.mean()
        prev = cur
    print(f"  TOTAL {tot:9.0f} cycles over {ok.sum()} workgroups; clock ticks at 100 MHz? -> see ratio to kernel time")
for K in (0, 1):
    s = st[K][:, 0, :]
    ok = s[:, 0] > 0
    s = s[ok]
    if not ok.any(): continue
    last = 26 if K == 0 else (27 if order2 is None else 23)
    t0 = s[:, 0].min()
    print(f"kernel {K}: WG start offsets", (s[:, 0] - t0).tolist(), "end offsets", (s[:, last] - t0).tolist())
