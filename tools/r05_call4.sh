#!/bin/bash
O=gpurun_out/r05c4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "gerstner or pond or bench_times" 2>&1 | tail -5
for i in 1 2 3; do for m in 1 0; do
  echo "MW_POND_MFMA=$m $(MW_POND_MFMA=$m python bench.py --workload pond --steps 3200 --warmup 320 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g v/s' % d['value'], 'launch %.1f us' % r['launch_us'], 'frac %.3f' % r['frac'], 'median %.1f p10 %.1f p90 %.1f' % (r['launch_us_stats']['median'], r['launch_us_stats']['p10'], r['launch_us_stats']['p90']), d['parity'][:2])")"
done; done 2>&1 | tee $O/pond_mfma_ab.txt
MW_POND_MFMA=1 python bench.py --workload pond --steps 20 --warmup 5 --batch 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b20', d['value'], d['roofline']['frac'], d['parity'])"
MW_POND_MFMA=1 python tools/pond_single.py 2>&1 | tail -3
