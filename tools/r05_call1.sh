#!/bin/bash
# round 5, GPU call 1: the whole GPU tier on the new build, the default bench line (all configs), the timing-only ablations that bound what
# an in-wave exchange / 16-lane store segments could buy, and the pond grid A/B.
mkdir -p gpurun_out/r05c1
O=gpurun_out/r05c1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 > $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -4 $O/bench_default.err
# ablations (timing only, parity gate off for them)
for spec in "ab_exch1:-DMW_ABLATE_EXCH1=1024" "ab_seg:-DMW_ABLATE_SEG_STORES=1"; do
  name=${spec%%:*}; flags=${spec#*:}
  bash tools/build_variant.sh $name $flags 2>&1 | grep -E "error" | head -3
done
for i in 1 2; do
  ABV_EXTRA="--no-latency --no-parity" bash tools/abv.sh "ocean4096 32 128" base ab_exch1 ab_seg
  ABV_EXTRA="--no-latency --no-parity" bash tools/abv.sh "ocean1024 32 1600" base ab_exch1
  ABV_EXTRA="--no-latency --no-parity" bash tools/abv.sh "ocean2048 32 320" base ab_exch1
done 2>&1 | tee $O/ablations.txt
# pond: XCD-grouped step groups vs the 2-D grid, steps per workgroup
for x in 1 0; do for spw in 8 16 32; do
  echo "MW_POND_XCD=$x MW_POND_STEPS_PER_WG=$spw $(MW_POND_XCD=$x MW_POND_STEPS_PER_WG=$spw python bench.py --workload pond --steps 3200 --warmup 320 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g v/s' % d['value'], 'launch %.1f us' % r['launch_us'], 'frac %.3f' % r['frac'], r['launch_us_stats'], d['parity'][:2])")"
done; done 2>&1 | tee $O/pond_ab.txt
for x in 1 0; do
  echo "MW_POND_XCD=$x spw=4 $(MW_POND_XCD=$x MW_POND_STEPS_PER_WG=4 python bench.py --workload pond --steps 3200 --warmup 320 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g v/s' % d['value'], 'launch %.1f us' % r['launch_us'], 'frac %.3f' % r['frac'])")"
done 2>&1 | tee -a $O/pond_ab.txt
