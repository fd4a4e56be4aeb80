#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
for n in 50 100 1000 2000; do
  timeout 300 python bench.py --workload direct --direct-n $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print('N=$n', round(d['ms_per_step']*1e3, 1), 'us/step', [(k['name'][:24], round(k['us_per_step'], 1)) for k in r['kernels']], 'frac', round(r['frac'], 3), d['parity'][:2])"
done 2>&1 | tee $out/czt_fused.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "direct or both_forms or shipped or whitecap_stage or inspector or Inspector" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
