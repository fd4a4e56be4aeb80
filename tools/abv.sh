#!/bin/bash
# A/B of kernel variants on the GPU box: tools/abv.sh "<workload> <batch> <steps>" variant1 variant2 ...   ("base" = the in-tree .so)
# prints value, us/step, whole-step frac, k_pass2 frac and per-kernel us for each variant; parity gate stays ON.
set -- $@
wl=$1; b=$2; st=$3; shift 3
for v in "$@"; do
  if [ "$v" == "base" ]; then lib=""; else lib="variants/$v.so"; fi
  out=$(env MW_ALLOW_LAB=1 MW_LIB=$lib python bench.py --workload $wl --batch $b --steps $st --warmup $b --no-cpu-baseline $ABV_EXTRA 2>&1 | tail -1)
  echo "$out" | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('$wl b=$b %-14s' % '$v', '%.4g pts/s'%d['value'], '%.2f us/step'%(d['ms_per_step']*1e3), 'step frac %.3f'%d['hbm_roofline_frac_whole_step'], 'p2 frac %.3f'%r['frac'], [round(k['us_per_launch'],1) for k in r.get('kernels',[])], (d['parity'] or 'none')[:2])
except Exception as e:
    print('$wl $v FAILED', e)
"
  [ -z "$lib" ] || true
done
