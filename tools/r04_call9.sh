#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
bash tools/build_variant.sh lir0 -DMW_LAST_IN_REGS=0 > /dev/null 2>&1
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean4096 32 256" base lir0 base 2>&1 | tee $out/ab_lir_4096.txt
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean512 32 3200" base lir0 2>&1 | tee -a $out/ab_lir_4096.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "4096 or large or literal or 512 or whitecap or parity" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
