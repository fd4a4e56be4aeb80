"""Cycle stamps of the single-step plan's kernels: k_pass2_frame per row-group kind, k_pass1<.., FS> per workgroup with the CU each one ran on.
Needs a -DMW_TIMING -DMW_STAMP_STEP=0 build: run tools/frame_stamps.sh on the GPU box (it builds the variant there)."""
import os, sys, ctypes as C
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mistral-water_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch; torch.cuda.init()
import mistral_water as mw
from mistral_water import _native as nat
import workloads
N = 1024; NN = N * N
p = workloads.fftmesh_config2(N)
o = mw.Ocean(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude, choppiness=p.choppiness)
dv = torch.empty((NN, 3), device="cuda"); dn = torch.empty((NN, 3), device="cuda"); dw = torch.empty((NN,), device="cuda")
for k in range(30):
    o.evaluate_device([0.1 * k], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
o.synchronize()
st = np.zeros((2, 64, 16, 32), np.int64)
L = nat.lib(); L.mw_debug_get_stamps.argtypes = [C.c_void_p]
assert L.mw_debug_get_stamps(st.ctypes.data) == 0
names = ["fetch + stage 0 written", "middle passes", "final pass", "publish hds / normals + noise out", "barrier", "vertex / whitecap stores issued"]
s = st[1]
ok = s[:, 0, 0] > 0
print("workgroups sampled:", int(ok.sum()))
for w, kind in ((0, "height"), (4, "displacement"), (8, "slopes"), (12, "halo")):
    x = s[ok][:, w, :7].astype(np.float64)
    if not (x[:, 0] > 0).all():
        print(kind, "not sampled"); continue
    d = np.diff(x, axis=1)
    print(f"{kind:13s} total {d.sum(axis=1).mean():8.0f} cycles |", " | ".join(f"{n}: {v:.0f}" for n, v in zip(names, d.mean(axis=0))))
for w, kind in ((0, "height"), (4, "displacement"), (8, "slopes")):
    x = s[ok][:, w, :]
    print(f"{kind}: cycles start->end {np.mean(x[:, 6] - x[:, 0]):.0f}, 100-MHz ticks {np.mean(x[:, 31] - x[:, 30]):.1f} -> {np.mean(x[:, 6] - x[:, 0]) / max(np.mean(x[:, 31] - x[:, 30]), 1e-9) / 10:.3f} cycles per ns")
t0 = s[ok][:, 0, 0].min()
print("start offsets of the sampled workgroups (cycles):", (s[ok][:, 0, 0] - t0).tolist())
print("end offsets:", (s[ok][:, 8, 6] - t0).tolist())
# ---- pass 1 of the single-step plan: where every workgroup ran (HW_ID: cu_id bits 11:8, sh 12, se 15:13; XCC_ID bits 3:0) and how long ----
import collections
jobs = None
rows = []
for b in range(768):
    r = st[0][b % 64][4 + b // 64]
    if r[2] <= 0: continue
    hw, xcc = int(r[0]), int(r[1]) & 15
    cu, sh, se, simd, wave = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 4) & 3, hw & 15
    rows.append((b, xcc, se, sh, cu, int(r[2]), int(r[3])))
t0 = min(r[5] for r in rows)
percu = collections.defaultdict(list)
for b, xcc, se, sh, cu, a, e in rows:
    percu[(xcc, se, sh, cu)].append((b, (a - t0) * 10, (e - t0) * 10))
print("pass-1 workgroups that ran:", len(rows), "on", len(percu), "distinct (xcc, se, sh, cu); workgroups per CU histogram:", sorted(collections.Counter(len(v) for v in percu.values()).items()))
dur = collections.defaultdict(list)
for v in percu.values():
    for b, a, e in v: dur[len(v)].append(e - a)
print("mean duration (ns) by number of workgroups sharing the CU:", {k: round(float(np.mean(v))) for k, v in sorted(dur.items())}, "| last end (ns):", max(e for v in percu.values() for _, _, e in v))
byb = sorted(rows)
print("duration by workgroup id (ns), every 16th:", [(b, (e - a) * 10) for b, xcc, se, sh, cu, a, e in byb[::16]])
