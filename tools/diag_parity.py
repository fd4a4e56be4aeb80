"""Diagnostic (not a test): error table of the GPU path and of the host emulation vs the f64 oracle."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "mistral-water_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch  # noqa
torch.cuda.is_available()
import mistral_water as mw
from oracle import oracle as O
import workloads, emul_build
E = emul_build.load()
for N in [int(x) for x in (sys.argv[1:] or ["256", "1024"])]:
    p = workloads.fftmesh_params(N)
    h0, h0c = O.generate_spectrum(p, 1)
    rest = O.rest_mesh(p)[0]
    with mw.Ocean(resolution=N, unit_width=1.0, length=float(N), wind=(p.wind_x, p.wind_y), amplitude=p.amplitude,
                  choppiness=p.choppiness) as o:
        o.set_spectrum(h0, h0c)
        for t in (0.0, 1.0, 16.65, 100.0):
            v, n, c = o.evaluate(t)
            vf, nf, cf, hds = O.eval_fft_f64(p, h0, h0c, t, True)
            ev, en, ew = E.evaluate(p, h0, h0c, [t])
            sc = np.abs(vf - rest).max()
            print(f"N={N} t={t}: GPU  h {np.abs(v[:,1]-vf[:,1]).max()/sc:.2e} n {np.abs(n-nf).max():.2e} w {np.abs(c-cf).max():.2e}"
                  f" | EMUL h {np.abs(ev[0][:,1]-vf[:,1]).max()/sc:.2e} n {np.abs(en[0]-nf).max():.2e} w {np.abs(ew[0]-cf).max():.2e}"
                  f" | GPU-EMUL h {np.abs(v[:,1]-ev[0][:,1]).max()/sc:.2e} n {np.abs(n-en[0]).max():.2e}")
