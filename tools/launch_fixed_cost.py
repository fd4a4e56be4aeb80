#!/usr/bin/env python
"""Per-launch fixed cost of the two batched kernels: launch duration against steps per enqueue (in situ, mw_ocean_profile_kernels_stats),
a straight-line fit  t(B) = a + b B  per kernel.  a = what a launch pays once (ramp, tail), b = the steady cost of a step.
   python tools/launch_fixed_cost.py [--n 1024]        -> one JSON object on stdout"""
import argparse, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "mistral-water_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=1024); ap.add_argument("--iters", type=int, default=60)
a = ap.parse_args()
import numpy as np
import torch
torch.cuda.init()
import mistral_water as mw, workloads
N = a.n
p = workloads.fftmesh_config2(N)
o = mw.Ocean(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude, choppiness=p.choppiness)
sizes = [1, 2, 4, 8, 12, 16, 20, 24, 32]
o.profile_kernels(nsteps=32, iters=60)      # clocks
rows = {}
for rep in range(2):
    for B in sizes:
        st = o.profile_kernels_stats(nsteps=B, iters=a.iters)
        for nm, s in st:
            rows.setdefault(nm, {}).setdefault(B, []).append(s["median"] * 1e3)
out = {"N": N, "iters": a.iters, "unit": "us per launch (median of the launches, two sweeps)", "kernels": {}}
for nm, d in rows.items():
    Bs = np.array([B for B in sizes if B >= 4], float)
    ts = np.array([min(d[B]) for B in sizes if B >= 4])
    b, a0 = np.polyfit(Bs, ts, 1)
    out["kernels"][nm] = {"us": {str(B): [round(x, 2) for x in d[B]] for B in sizes},
                          "fit_B_ge_4": {"fixed_us_per_launch": round(float(a0), 2), "us_per_step": round(float(b), 3)},
                          "fixed_share_at_20": round(float(a0 / (a0 + 20 * b)), 4), "fixed_share_at_32": round(float(a0 / (a0 + 32 * b)), 4)}
print(json.dumps(out))
