#!/bin/bash
# twiddle staging: the tree (MW_TW_STAGE=2: batched loads, LDS writes behind the first data requests) against variants built beforehand:
#   tw1 = -DMW_TW_STAGE=1 (batched loads, written at once), tw0 = -DMW_TW_STAGE=0 (the copy loop of rounds 1-4)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/tw_stage_ab.txt; : > $out
for rep in 1 2; do
  for wl in "ocean1024 20 640" "ocean1024 32 640" "ocean4096 32 128" "ocean2048 32 128" "ocean512 32 1280" "ocean256 32 2560"; do
    ABV_EXTRA="--no-latency" bash tools/abv.sh "$wl" base tw1 tw0 >> $out 2>&1
  done
  for v in "" variants/tw1.so variants/tw0.so; do
    for n in 1024 512 256; do
      r=$(MW_LIB=$v timeout 200 python tools/frame_probe.py --n $n 2>/dev/null | tail -1)
      echo "frame n=$n lib=${v:-tree} $(echo "$r" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v['back_to_back_us_per_step'],2) for k,v in d.items() if isinstance(v,dict)})")" >> $out
    done
    r=$(MW_LIB=$v timeout 300 python bench.py --workload renderer1024 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "renderer1024 lib=${v:-tree} $(echo "$r" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/frame' % (d['ms_per_step']*1e3), (d.get('parity') or 'none')[:2])")" >> $out
  done
done
cat $out
