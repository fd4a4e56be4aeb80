#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
bash tools/build_variant.sh rowbar -DMW_ROW_BARRIERS=1 > /dev/null 2>&1
for v in base rowbar base rowbar; do ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean1024 32 1600" $v; done 2>&1 | tee $out/ab_rowsync_1024.txt
timeout 200 python tools/frame_probe.py 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('frame', {k: (round(v['back_to_back_us_per_step'], 2), round(v['sync_latency_us_median'], 2)) for k, v in d.items() if isinstance(v, dict)})"
timeout 900 python -m pytest tests -m gpu -q -x -k "1024 or whitecap or frame or literal" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5
