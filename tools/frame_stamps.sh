#!/bin/bash
# Cycle stamps + HW_ID placement of the single-step plan's two kernels (GPU box; builds the -DMW_TIMING variant there: variants/ does not travel)
cd "$(dirname "$0")/.."
bash tools/build_variant.sh timing0 -DMW_TIMING -DMW_STAMP_STEP=0 > /dev/null 2>&1
mkdir -p gpurun_out/frame_ab
MW_LIB=variants/timing0.so timeout 300 python tools/frame_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/frame_ab/stamps.txt
