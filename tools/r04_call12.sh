#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
bash tools/build_variant.sh czt_nox -DMW_CZT_XCD_GROUP=0 > /dev/null 2>&1
for v in "" variants/czt_nox.so; do for n in 100 1000 2000; do
  MW_LIB=$v timeout 300 python bench.py --workload direct --direct-n $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print('$v N=$n', round(d['ms_per_step']*1e3, 1), 'us/step', [(k['name'][:20], round(k['us_per_step'], 1)) for k in r['kernels']], 'frac', round(r['frac'], 3), d['parity'][:2])"
done; done 2>&1 | tee $out/czt_xcd.txt
MW_BENCH_FORCE_TILES=1 timeout 300 python bench.py --steps 640 --warmup 64 --gather --no-cpu-baseline --no-latency 2> $out/gather.err | tail -1 > $out/bench_tiles_gather.json
python -c "
import json; d = json.loads(open('$out/bench_tiles_gather.json').read()); print('tiles path', d['value'], d['ms_per_step'], d['config']['api'], d.get('with_gather'))"
timeout 600 python -m pytest tests -m gpu -q -x -k "direct or both_forms or shipped or baseline_config3" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4
