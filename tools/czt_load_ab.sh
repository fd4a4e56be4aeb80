#!/bin/bash
# chirp-z input loads: eight elements requested together (the tree) against element by element (cl1 = -DMW_CZT_LOAD_CHUNK=1) and four (cl4); built on the box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export MW_REPORT_FILTER="k_czt<2048"
bash tools/build_variant.sh cl1 -DMW_CZT_LOAD_CHUNK=1 2>&1 | tail -1; bash tools/build_variant.sh cl8 -DMW_CZT_LOAD_CHUNK=8 2>&1 | tail -1
out=gpurun_out/czt_load_ab.txt; : > $out
for rep in 1 2; do
for lib in "" variants/cl8.so variants/cl1.so; do
  for n in 12 20 50 100 200 500 1000 2000; do
    r=$(MW_LIB=$lib timeout 300 python bench.py --workload direct --direct-n $n --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "$r" | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('lib=%-18s N=%-5d %.2f us/step  parity %s' % ('${lib:-tree}', $n, d['ms_per_step']*1e3, (d.get('parity') or 'none')[:2]))
except Exception as e: print('lib=${lib:-tree} N=$n FAILED', e)
" >> $out
  done
done
done
cat $out
