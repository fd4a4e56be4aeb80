#!/bin/bash
# chirp-z small grids: the tree (tables staged in LDS + early table loads, wave-level line exchanges, one launch up to N = 32) against variants
# built beforehand:  cztnows = -DMW_CZT_WAVE_SYNC=0,  cztold = -DMW_CZT_PRELOAD=0 -DMW_CZT_WAVE_SYNC=0 -DMW_CZT_ONE_MAX_N=0 (the plan before)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/czt_preload_ab.txt; : > $out
for rep in 1 2; do
for lib in "" variants/cztnows.so variants/cztold.so; do
  for n in 12 24 50 100 200 500 1000; do
    r=$(MW_LIB=$lib timeout 300 python bench.py --workload direct --direct-n $n --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$r" | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('lib=%-20s N=%-5d %.2f us/step  parity %s' % ('${lib:-tree}', $n, d['ms_per_step']*1e3, (d.get('parity') or 'none')[:2]))
except Exception as e: print('lib=${lib:-tree} N=$n FAILED', e)
" >> $out
  done
done
done
cat $out
