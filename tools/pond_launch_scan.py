#!/usr/bin/env python
"""Pond (1M vertices, 8 waves): duration of mw_gerstner_displace_steps_device launches against time values per launch, HIP events between
back-to-back launches; a straight-line fit t(B) = a + b B.  Environment switches of the library (MW_POND_STEPS_PER_WG, MW_POND_XCD) are
read once per process: run one process per setting.     python tools/pond_launch_scan.py [--nv 1000000]  -> one JSON object"""
import argparse, ctypes as C, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "mistral-water_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--nv", type=int, default=1000 * 1000); ap.add_argument("--launches", type=int, default=200)
a = ap.parse_args()
import numpy as np
import torch
torch.cuda.init()
import workloads
from mistral_water import _native as nat
lib = nat.lib()
dev = torch.device("cuda", 0)
nv = a.nv
side = int(round(nv ** 0.5))
g = torch.linspace(-50, 50, side + 1, device=dev)[:-1]
pos = torch.stack(torch.meshgrid(g, g, indexing="ij"), -1)
pos = torch.stack([pos[..., 0], torch.zeros_like(pos[..., 0]), pos[..., 1]], -1).reshape(-1, 3).contiguous()
nv = pos.shape[0]
W = np.ascontiguousarray(workloads.pond_waves8(), np.float32)
P = workloads.POND
out_t = torch.empty((32, nv, 3), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream()
res = {"nv": nv, "env": {k: v for k, v in os.environ.items() if k.startswith("MW_POND")}, "lib": os.environ.get("MW_LIB", "base"), "us_per_launch": {}}
sizes = [4, 8, 16, 24, 32]
for rep in range(2):
    for B in sizes:
        tt = np.array([(k + 1) / 60.0 for k in range(B)], np.float32)
        args = (C.c_void_p(pos.data_ptr()), nv, W.ctypes.data_as(C.c_void_p), 8, C.c_float(P["amplitude"]), C.c_float(P["frequency"]),
                C.c_float(P["steepness"]), tt.ctypes.data_as(C.c_void_p), B, C.c_void_p(out_t.data_ptr()), C.c_void_p(stream.cuda_stream))
        for _ in range(40):
            nat.check(lib.mw_gerstner_displace_steps_device(*args))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.launches + 1)]
        evs[0].record(stream)
        for i in range(a.launches):
            nat.check(lib.mw_gerstner_displace_steps_device(*args))
            evs[i + 1].record(stream)
        torch.cuda.synchronize()
        ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.launches))
        res["us_per_launch"].setdefault(str(B), []).append(round(ms[len(ms) // 2] * 1e3, 2))
Bs = np.array([B for B in sizes if B >= 8], float)
ts = np.array([min(res["us_per_launch"][str(B)]) for B in sizes if B >= 8])
b, a0 = np.polyfit(Bs, ts, 1)
res["fit_B_ge_8"] = {"fixed_us_per_launch": round(float(a0), 2), "us_per_time_value": round(float(b), 4),
                     "steady_TBps": round(12.0 * nv / (b * 1e-6) / 1e12, 3), "frac_at_32": round((12.0 + 12.0 / 32) * nv * 32 / (min(res["us_per_launch"]["32"]) * 1e-6) / 8e12, 4)}
print(json.dumps(res))
