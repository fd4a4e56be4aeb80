#!/usr/bin/env python
"""Which kernel-enforced memory limits does the ROCm runtime survive?  (GPU box; output -> profiles/r04_memguard_probe.txt)
Each mode runs in a child: set the limit, initialise HIP through the product library, run one 256^2 step, then ask numpy for 200 GB."""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "mistral-water_amd")
import memguard
print("mode", memguard.install(64))
import numpy as np
import torch; torch.cuda.init()      # PyTorch's bundled HIP runtime initialises first (INTEGRATION.md)
import mistral_water as mw
from mistral_water import FFTMesh
m = FFTMesh(); m.resolution = 256; m.length = 256.0; m.unitWidth = 1.0; m.Awake(); m.Update(1.0 / 60)
torch.zeros(8, device="cuda").sum().item()
print("hip ok")
try:
    a = np.ones(200 * 2**30 // 8); print("200 GB allocated and touched?!", a[-1])
except MemoryError:
    print("numpy 200 GB -> MemoryError (limit enforced)")
'''
for mode in ("data", "as", "watchdog"):
    env = dict(os.environ, MW_MEMGUARD=mode, MW_HOST_MEM_CAP_GB="64")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(f"== MW_MEMGUARD={mode}: rc {r.returncode}\n" + r.stdout[-400:] + r.stderr[-600:])
