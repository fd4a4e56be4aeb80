#!/bin/bash
# A/B of kernel variants on the pond workload: tools/ab_pond.sh <steps> variant1 variant2 ...  ("base" = in-tree .so)
st=$1; shift
for v in "$@"; do
  if [ "$v" == "base" ]; then lib=""; else lib="variants/$v.so"; fi
  env MW_LIB=$lib python bench.py --workload pond --steps $st --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pond %-10s' % '$v', '%.4g %s' % (d['value'], d['unit']), '%.3f us/step' % (d['ms_per_step']*1e3), 'frac %.3f' % d['roofline']['frac'], (d.get('parity') or 'none')[:2])"
done
