#!/bin/bash
# pond: vertices per lane x steps per workgroup, launch duration against time values per launch (tools/pond_launch_scan.py); variants built beforehand
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/pond_vpl_ab.txt; : > $out
for rep in 1 2; do
for v in vpl4 vpl2 vpl1 vpl8; do
  for spw in 8 4 16; do
    r=$(MW_LIB=variants/$v.so MW_POND_STEPS_PER_WG=$spw timeout 300 python tools/pond_launch_scan.py 2>/dev/null | tail -1)
    echo "$v spw=$spw $r" | python -c "
import sys,json
l=sys.stdin.read().strip(); tag,spw,js=l.split(' ',2)
try:
    d=json.loads(js); print(tag,spw,'us/launch',{k:min(v) for k,v in d['us_per_launch'].items()},d['fit_B_ge_8'])
except Exception as e: print(tag,spw,'FAILED',e,js[:200])
" >> $out
  done
done
done
cat $out
