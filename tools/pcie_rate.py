"""PCIe-inclusive rate of the host-buffer boundary (mw_ocean_evaluate: one step, results copied into caller arrays),
next to the device-resident rate bench.py reports.  Never the headline value -- DESIGN.md section 1 quotes it."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mistral-water_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (initialises HIP before the library loads)
import mistral_water as mw
import workloads

for N in (256, 1024):
    p = workloads.fftmesh_params(N)
    o = mw.Ocean(resolution=N, unit_width=p.unit_width, length=p.length, wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                 choppiness=p.choppiness, gravity=p.gravity, seed=1)
    for k in range(5):
        o.evaluate(0.1 * k)
    n = 30
    t0 = time.perf_counter()
    for k in range(n):
        o.evaluate(1.0 + k / 60.0)
    el = (time.perf_counter() - t0) / n
    bytes_out = N * N * (12 + 12 + 16)
    print(f"N={N}: mw_ocean_evaluate (host arrays, Color output) {el*1e3:.3f} ms/step = {N*N/el:.3e} grid-points/s, "
          f"{bytes_out/el/1e9:.1f} GB/s device-to-host")
    import numpy as np
    v, nr, c = np.empty((N * N, 3), np.float32), np.empty((N * N, 3), np.float32), np.empty((N * N, 4), np.float32)
    for arr in (v, nr, c):
        mw.host_register(arr)
    for k in range(5):
        o.evaluate_into(0.1 * k, v, nr, c)
    t0 = time.perf_counter()
    for k in range(n):
        o.evaluate_into(1.0 + k / 60.0, v, nr, c)
    el = (time.perf_counter() - t0) / n
    print(f"N={N}: the same into arrays registered once with mw_host_register {el*1e3:.3f} ms/step = {N*N/el:.3e} "
          f"grid-points/s, {bytes_out/el/1e9:.1f} GB/s")
    for arr in (v, nr, c):
        mw.host_unregister(arr)
    o.close()
