#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO=V ...]  ->  variants/NAME.so (A/B kernel variants; select with MW_LIB=variants/NAME.so)
name=$1; shift
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value -fno-slp-vectorize "$@" \
  -o variants/$name.so mistral-water_amd/csrc/mistral_water.hip
