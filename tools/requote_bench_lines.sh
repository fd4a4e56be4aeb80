#!/bin/bash
# After `cp gpurun_out/profiles_<tag>/* profiles/`: take the bench lines that quote roofline.traffic once more, now that the counter
# files of THIS build are in place (inside profile_round.sh they were taken before the files existed: traffic null).
#   gpurun --timeout 1200 -- 'bash tools/requote_bench_lines.sh r04'   then copy gpurun_out/profiles_<tag>/*bench* back again
tag=${1:-r06}
P=gpurun_out/profiles_${tag}; mkdir -p $P
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $P/${tag}_bench_ocean1024_driver_k20.json
timeout 300 python bench.py --steps 640 --warmup 64 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${tag}_bench_ocean1024_b32_steps640.json
timeout 300 python bench.py --workload ocean2048 --steps 128 --warmup 32 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $P/${tag}_bench_ocean2048.json
timeout 300 python bench.py --workload ocean4096 --steps 128 --warmup 32 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $P/${tag}_bench_ocean4096.json
timeout 300 python bench.py --workload pond --steps 3200 --warmup 320 2>/dev/null | tail -1 > $P/${tag}_bench_pond.json
timeout 300 python bench.py --workload renderer1024 --batch 32 --steps 640 --warmup 64 2>/dev/null | tail -1 > $P/${tag}_bench_renderer1024.json
timeout 300 python bench.py --workload renderer1024 --batch 1 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${tag}_bench_renderer1024_frame.json
timeout 300 python bench.py --workload renderer1024 --batch 1 --tiles 4 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${tag}_bench_renderer1024_tiles4.json
python tools/bench_summary.py $P/${tag}_bench_ocean1024_driver_k20.json $P/${tag}_bench_ocean1024_b32_steps640.json $P/${tag}_bench_ocean2048.json $P/${tag}_bench_ocean4096.json $P/${tag}_bench_pond.json $P/${tag}_bench_renderer1024.json $P/${tag}_bench_renderer1024_frame.json $P/${tag}_bench_renderer1024_tiles4.json
