#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun); outputs land in gpurun_out/.
#   1. --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass; no trace domains mixed in)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 1600 --no-cpu-baseline > $OUT/bench_stdout.txt 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr " " "_" | cut -c1-28)
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$tag -o pmc -- python bench.py --steps 64 --warmup 32 --preheat-ms 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
done
find $OUT -name "*.csv" | head -40
# other workloads: per-kernel durations only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_renderer -o renderer -- python bench.py --workload renderer1024 --steps 2000 > $OUT/renderer_stdout.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pond -o pond -- python bench.py --workload pond --steps 3200 > $OUT/pond_stdout.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_4096 -o o4096 -- python bench.py --workload ocean4096 --batch 4 --steps 200 --warmup 4 --no-cpu-baseline > $OUT/o4096_stdout.txt 2>&1
# HBM-side traffic of the other workloads' kernels (FETCH_SIZE / WRITE_SIZE in separate passes)
for w in renderer1024 pond; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o pmc -- python bench.py --workload $w --steps 64 --warmup 8 --preheat-ms 0 --no-parity > /dev/null 2>&1
  done
done
