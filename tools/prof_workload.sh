#!/bin/bash
# rocprofv3 evidence for ONE bench workload on the GPU box (run through gpurun); outputs land in gpurun_out/prof_<tag>/.
#   usage: tools/prof_workload.sh TAG WORKLOAD BATCH STEPS [extra bench args]
#   1. --kernel-trace --stats of the bench command (per-kernel average durations)
#      (bench.py prints mw_build_id(); tools/summarize_profiles.py stores that line, and bench.py quotes the counters as
#      roofline.traffic only while the library is that build)
#   2. separate --pmc passes, one counter group per run, no trace domains mixed in (FETCH_SIZE / WRITE_SIZE cannot share
#      a pass); PMC passes run BATCH-sized launches only (steps = warm-up = BATCH multiples) so that every launch of a
#      kernel processes the same number of units and a per-launch mean is meaningful.
tag=$1; wl=$2; batch=$3; steps=$4; shift 4
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --workload $wl --batch $batch --steps $steps --warmup $batch --no-cpu-baseline --no-latency "$@" > $OUT/bench_stdout.txt 2>&1
tail -c 600 $OUT/bench_stdout.txt
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  ctag=$(echo $c | tr " " "_" | cut -c1-28)
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$ctag -o pmc -- python bench.py --workload $wl --batch $batch --steps $((batch * 2)) --warmup $batch --preheat-ms 0 --no-cpu-baseline --no-parity --no-latency "$@" > /dev/null 2>&1
done
find $OUT -name "*.csv" | wc -l
