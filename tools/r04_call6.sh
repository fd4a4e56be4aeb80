#!/bin/bash
bash tools/build_variant.sh p16_t -DMW_TIMING -DMW_PT1_4096=16 > /dev/null 2>&1
N=4096 B=8 MW_LIB=variants/p16_t.so timeout 300 python tools/phase_timing.py 2>&1 | grep -v amdgpu.ids | tail -60
