#!/bin/bash
# A/B of the single-step (frame-at-a-time) plan's kernels on one box: parity first, then back-to-back latency and the kernel trace.
# usage (GPU box): bash tools/frame_ab.sh   -> gpurun_out/frame_ab/
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/frame_ab; mkdir -p $O
timeout 600 python -m pytest tests/test_zz_frame_plan.py -x -q -m gpu -k "single_step" 2>&1 | tail -3 | tee $O/pytest.txt
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do
  set -- $cfg
  echo "== MW_FRAME_KERNEL=$1 MW_P1_FRAME_XCD=$2" | tee -a $O/probe.txt
  MW_FRAME_KERNEL=$1 MW_P1_FRAME_XCD=$2 timeout 300 python tools/frame_probe.py 2>&1 | tail -1 | tee -a $O/probe.txt
done
for cfg in "1 1" "0 0"; do
  set -- $cfg
  rm -rf /tmp/fp_$1$2
  MW_FRAME_KERNEL=$1 MW_P1_FRAME_XCD=$2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fp_$1$2 -o fp --output-format csv -- python tools/frame_probe.py > /dev/null 2>&1
  f=$(find /tmp/fp_$1$2 -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats MW_FRAME_KERNEL=$1 MW_P1_FRAME_XCD=$2" | tee -a $O/kstats.txt
  head -8 "$f" | cut -c1-220 | tee -a $O/kstats.txt
done
