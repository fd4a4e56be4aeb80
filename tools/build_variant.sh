#!/bin/bash
# tools/build_variant.sh NAME [-DMACRO=V ...]  ->  variants/NAME.so (A/B kernel variants; select with MW_LIB=variants/NAME.so)
# prints VGPRs / spills / occupancy of the pass kernels (-Rpass-analysis=kernel-resource-usage)
name=$1; shift
mkdir -p variants
set -- "$@" -DMW_LAB   # a variant is a lab build: mw_build_id() says "lab:NAME"; run bench.py / tests on it with MW_ALLOW_LAB=1
hash=$(PYTHONPATH=mistral-water_amd python3 -c "import sys; from mistral_water._native import source_hash; print(source_hash(sys.argv[1:]))" "$@")
hipcc -DMW_BUILD_HASH="\"$hash\"" -DMW_BUILD_TAG="\"$name\"" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value -fno-slp-vectorize -mllvm -pragma-unroll-threshold=1000000 -Xclang -target-feature -Xclang -load-store-opt "$@" \
  -Rpass-analysis=kernel-resource-usage -o variants/$name.so mistral-water_amd/csrc/mistral_water.hip 2> variants/$name.res
grep -E "error" variants/$name.res | head -5
python3 tools/resource_report.py variants/$name.res ${MW_REPORT_FILTER:-k_pass}
