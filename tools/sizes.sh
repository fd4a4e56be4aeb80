#!/bin/bash
# steady-state sweep over grid sizes: tools/sizes.sh [extra bench args]
for w in "ocean256 32 20000" "ocean512 32 8000" "ocean1024 32 4000" "ocean2048 32 640" "ocean4096 32 192"; do
  set -- $w
  python bench.py --workload $1 --batch $2 --steps $3 --warmup $2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 b=$2', '%.4g'%d['value'], '%.2f us/step'%(d['ms_per_step']*1e3), 'whole-step frac %.3f'%d['hbm_roofline_frac_whole_step'], 'k_pass2 frac %.3f'%r['frac'], [round(k['us_per_launch'],1) for k in r.get('kernels',[])])"
done
