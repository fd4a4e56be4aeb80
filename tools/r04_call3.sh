#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
bash tools/build_variant.sh p1_16 -DMW_PT1_4096=16 > /dev/null 2>&1
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean4096 32 256" base p1_16 base 2>&1 | tee $out/ab_p64_4096.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "4096 or large or literal or both_forms or direct" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8
