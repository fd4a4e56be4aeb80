#!/bin/bash
# A/B helper: tools/ab.sh "<ENVVAR>" v1 v2 ... [-- extra bench args]; prints value, us/step, k_pass2 frac and per-kernel us.
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  out=$(env MW_ALLOW_LAB=1 $var=$v python bench.py --steps 6400 --warmup 640 --no-cpu-baseline "$@" 2>&1 | tail -1)
  echo "$out" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$var=$v', '%.4g'%d['value'], '%.2f us/step'%(d['ms_per_step']*1e3), 'frac %.3f'%r['frac'], [round(k['us_per_launch'],1) for k in r.get('kernels',[])])"
done
