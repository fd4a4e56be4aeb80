#!/usr/bin/env python
"""Frame-at-a-time timing of the 1024^2 FFTMesh step outside bench.py: which stream, memguard on/off, back-to-back vs synchronised.
   python tools/frame_probe.py [--guard] [--n 1024]"""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "mistral-water_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
ap = argparse.ArgumentParser(); ap.add_argument("--guard", action="store_true"); ap.add_argument("--n", type=int, default=1024)
a = ap.parse_args()
if a.guard:
    import memguard; memguard.install()
import torch
torch.cuda.init()
import mistral_water as mw, workloads
N = a.n; NN = N * N
p = workloads.fftmesh_config2(N)
dev = torch.device("cuda", 0)
dv = torch.empty((NN, 3), dtype=torch.float32, device=dev); dn = torch.empty_like(dv); dw = torch.empty((NN,), dtype=torch.float32, device=dev)
out = {"guard": a.guard, "N": N}
for name in ("own_stream", "torch_null_stream", "torch_side_stream"):
    o = mw.Ocean(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude, choppiness=p.choppiness)
    side = None
    if name == "torch_null_stream":
        o.set_stream(torch.cuda.current_stream().cuda_stream)
    elif name == "torch_side_stream":
        side = torch.cuda.Stream(); o.set_stream(side.cuda_stream)
    def call(k):
        o.evaluate_device([(k + 1) / 60.0], dv.data_ptr(), dn.data_ptr(), dw.data_ptr())
    for k in range(50): call(k)
    o.synchronize()
    t0 = time.perf_counter()
    for k in range(200): call(k)
    t1 = time.perf_counter()
    o.synchronize()
    t2 = time.perf_counter()
    lat = []
    for k in range(100):
        s = time.perf_counter(); call(k); o.synchronize(); lat.append(time.perf_counter() - s)
    out[name] = {"back_to_back_us_per_step": (t2 - t0) / 200 * 1e6, "host_enqueue_us_per_call": (t1 - t0) / 200 * 1e6,
                 "sync_latency_us_median": sorted(lat)[50] * 1e6, "sync_latency_us_min": min(lat) * 1e6}
    o.close()
print(json.dumps(out))
