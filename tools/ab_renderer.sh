#!/bin/bash
# A/B of kernel variants on the OceanRenderer frame: tools/ab_renderer.sh <tiles> <steps> variant1 variant2 ...  ("base" = in-tree .so)
t=$1; st=$2; shift 2
for v in "$@"; do
  if [ "$v" == "base" ]; then lib=""; else lib="variants/$v.so"; fi
  env MW_LIB=$lib python bench.py --workload renderer1024 --tiles $t --steps $st --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('renderer1024 tiles=$t %-10s' % '$v', '%.4g %s' % (d['value'], d['unit']), '%.2f us/step' % (d['ms_per_step']*1e3), 'frac %.3f' % d['roofline']['frac'], (d.get('parity') or 'none')[:2])"
done
