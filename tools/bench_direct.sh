#!/bin/bash
# the direct-sum workload at the three sizes of VERDICT r2 item 4 (50 = Inspector default, 100, 1000): gpurun_out/<dir>/direct_<N>.json
out=${1:-gpurun_out/direct}
mkdir -p $out
for n in 50 100 1000; do
  timeout 300 python bench.py --workload direct --direct-n $n --steps 200 --warmup 20 2>$out/direct_$n.err | tail -1 > $out/direct_$n.json
done
python tools/bench_summary.py $out/direct_50.json $out/direct_100.json $out/direct_1000.json
