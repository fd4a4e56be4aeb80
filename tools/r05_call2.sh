#!/bin/bash
# round 5, GPU call 2: GPU tier on the build with the in-wave exchange (LastInWave) and the packed pond math; A/B of both against builds
# with them switched off (same box, interleaved, parity gate on); the single-step plan; the chirp-z bounds.
O=gpurun_out/r05c2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 > $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
bash tools/build_variant.sh nolw -DMW_LAST_IN_WAVE=0 2>&1 | grep -E "error|hs<1024|frame<1024"
bash tools/build_variant.sh nopk -DMW_POND_PACKED=0 2>&1 | grep -E "error"
MW_REPORT_FILTER=k_gerstner python tools/resource_report.py variants/nopk.res k_gerstner
for i in 1 2 3; do
  ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean1024 32 1600" base nolw
  ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean1024 20 1000" base nolw
done 2>&1 | tee $O/ab_last_in_wave.txt
bash tools/frame_variants.sh "base:" "nolw:-DMW_LAST_IN_WAVE=0" 2>&1 | tee $O/frame_ab.txt
bash tools/frame_variants.sh "base:" "nolw:-DMW_LAST_IN_WAVE=0" 2>&1 | tee -a $O/frame_ab.txt
for i in 1 2 3; do for v in "" variants/nopk.so; do
  echo "pond lib=${v:-base} $(MW_LIB=$v python bench.py --workload pond --steps 3200 --warmup 320 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4g v/s' % d['value'], 'launch %.1f us' % r['launch_us'], 'frac %.3f' % r['frac'], 'median %.1f p10 %.1f p90 %.1f' % (r['launch_us_stats']['median'], r['launch_us_stats']['p10'], r['launch_us_stats']['p90']), d['parity'][:2])")"
done; done 2>&1 | tee $O/pond_ab.txt
for n in 50 100 1000 2000; do
  timeout 300 python bench.py --workload direct --direct-n $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/direct_$n.json
  python -c "import json; d=json.load(open('$O/direct_$n.json')); b=d['roofline'].get('transform_bounds') or {}; print($n, '%.1f us/step' % (d['ms_per_step']*1e3), {k: (round(v,3) if isinstance(v,float) else v) for k,v in b.items() if k!='note'})"
done 2>&1 | tee $O/direct_bounds.txt
