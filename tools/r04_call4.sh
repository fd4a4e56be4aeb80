#!/bin/bash
bash tools/build_variant.sh p64w_t -DMW_TIMING > /dev/null 2>&1
MW_LIB=variants/p64w_t.so timeout 300 python tools/phase_timing_wave.py 2>&1 | grep -v amdgpu.ids | tail -40
