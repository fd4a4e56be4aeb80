"""FETCH_SIZE calibration driver: run under `rocprofv3 --pmc FETCH_SIZE` (tools/calib_fetch.sh)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "mistral-water_amd"))
import torch; torch.cuda.is_available()
import mistral_water as mw
L = mw.lib()
for width in (4, 8, 16):
    assert L.mw_debug_stream_read(1 << 30, width, 3) == 0   # 1 GiB >> 256 MiB Infinity Cache
