#!/bin/bash
# A/B of run-time knobs (environment variables of libmistral_water.so) on the GPU box:
#   tools/ab_env.sh "<workload> <batch> <steps>" "VAR=V [VAR2=V2]" "..." ...      ("-" = no variables)
# prints value, us/step, whole-step frac, k_pass2 frac and per-kernel us per setting; parity gate stays ON.
set -- "$@"
read wl b st <<< "$1"; shift
for envs in "$@"; do
  [ "$envs" == "-" ] && envs=""
  out=$(env $envs python bench.py --workload $wl --batch $b --steps $st --warmup $b --no-cpu-baseline --no-latency $ABV_EXTRA 2>&1 | tail -1)
  echo "$out" | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('$wl b=$b %-28s' % '$envs', '%.4g pts/s'%d['value'], '%.2f us/step'%(d['ms_per_step']*1e3), 'step frac %.3f'%d['hbm_roofline_frac_whole_step'], 'p2 frac %.3f'%r['frac'], [round(k['us_per_launch'],1) for k in r.get('kernels',[])], (d['parity'] or 'none')[:2])
except Exception as e:
    print('$wl $envs FAILED', e)
"
done
