#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
bash tools/build_variant.sh nokeep -DMW_KEEP_T1=0 > /dev/null 2>&1
for wl in "ocean1024 32 1600" "ocean1024 20 1000" "ocean2048 32 320" "ocean512 32 3200"; do
  ABV_EXTRA="--no-latency" bash tools/abv.sh "$wl" base nokeep base nokeep
done 2>&1 | tee $out/ab_keep_t1.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "1024 or 2048 or large or literal or whitecap or parity or frame" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
