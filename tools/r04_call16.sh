#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
(bash tools/build_variant.sh m2k_pf1 -DMW_SPLIT_SLOPES_4096=2 -DMW_KEEP_T1_MAX_N=4096 > /dev/null 2>&1 &
 bash tools/build_variant.sh m2k_pf0 -DMW_SPLIT_SLOPES_4096=2 -DMW_KEEP_T1_MAX_N=4096 -DMW_PF_4096=0 > /dev/null 2>&1 & wait)
for v in "" variants/m2k_pf1.so variants/m2k_pf0.so "" variants/m2k_pf1.so variants/m2k_pf0.so; do
  MW_LIB=$v timeout 400 python bench.py --workload ocean4096 --steps 512 --warmup 64 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s' % ('$v' or 'base'), 'wall %.2f (%.2f..%.2f)' % (d['ms_per_step']*1e3, d['ms_per_step_min']*1e3, d['ms_per_step_max']*1e3), 'events %.2f' % (d['event_ms_per_step']*1e3), 'in-situ', [round(k['us_per_launch'], 1) for k in r['kernels']], 'sum/32 %.2f' % (sum(k['us_per_launch'] for k in r['kernels'])/32), d['parity'][:2])"
done 2>&1 | tee $out/ab_keep_t1_4096_long.txt
