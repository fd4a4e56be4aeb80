#!/bin/bash
# round 5, GPU call 3: the in-wave exchange with the builtin's two results read correctly; fused small-grid chirp-z
O=gpurun_out/r05c3; mkdir -p $O
python tools/probe_permlane.py 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 > $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
bash tools/build_variant.sh nolw -DMW_LAST_IN_WAVE=0 2>&1 | grep -E "error|hs<1024|frame<1024"
for i in 1 2 3; do
  ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean1024 32 1600" base nolw
  ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean1024 20 1000" base nolw
done 2>&1 | tee $O/ab_last_in_wave.txt
bash tools/frame_variants.sh "base:" "nolw:-DMW_LAST_IN_WAVE=0" 2>&1 | tee $O/frame_ab.txt
bash tools/frame_variants.sh "nolw:-DMW_LAST_IN_WAVE=0" "base:" 2>&1 | tee -a $O/frame_ab.txt
for f in 1 0; do for n in 12 50 100 128; do
  echo "MW_CZT_FUSED=$f N=$n $(MW_CZT_FUSED=$f timeout 300 python bench.py --workload direct --direct-n $n --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us/step' % (d['ms_per_step']*1e3), [(k['name'][:24], round(k['us_per_step'],2)) for k in d['roofline']['kernels']], (d['parity'] or 'none')[:2])")"
done; done 2>&1 | tee $O/czt_fused_ab.txt
