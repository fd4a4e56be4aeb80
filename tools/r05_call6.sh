#!/bin/bash
O=gpurun_out/r05c6; mkdir -p $O
timeout 900 python -m pytest tests/test_state_and_tiles.py -m gpu -q -k "gather or tiles" 2>&1 | tail -3
for i in 1 2; do
for g in 5 10 20 4; do
  echo "B=20 MW_P1_TGROUP=$g $(MW_P1_TGROUP=$g python bench.py --workload ocean1024 --steps 1000 --batch 20 --warmup 20 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f us/step' % (d['ms_per_step']*1e3), 'tg', d['config']['pass1_time_group'], [round(k['us_per_launch'],1) for k in r['kernels']], d['parity'][:2])")"
done; done 2>&1 | tee $O/tgroup_b20.txt
for i in 1 2; do
  MW_BENCH_FORCE_TILES=1 timeout 300 python bench.py --workload ocean1024 --steps 640 --warmup 64 --gather --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['with_gather']; print('local copy: without %.4g with %.4g  ratio %.3f' % (d['value'], g['value'], g['value']/d['value']), g['region_ms'])"
  MW_TILES_FORCE_RCCL=1 MW_BENCH_FORCE_TILES=1 timeout 300 python bench.py --workload ocean1024 --steps 640 --warmup 64 --gather --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['with_gather']; print('rccl self-send: without %.4g with %.4g  ratio %.3f' % (d['value'], g['value'], g['value']/d['value']), g['region_ms'])"
done 2>&1 | tee $O/gather_cost.txt
