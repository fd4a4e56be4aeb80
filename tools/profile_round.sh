#!/bin/bash
# rocprofv3 evidence of the CURRENT build, every workload at the batch sizes bench.py reports (run through gpurun; ~6 min):
#   tools/profile_round.sh r06     (then set PROFILE_ROUND = "r06" in bench.py)  ->  gpurun_out/prof_<tag>_<workload>/ and the condensed profiles/<tag>_<workload>_b<B>_{kernel_stats.csv,pmc.json}
# Each summary stores the bench line of the traced run (with mw_build_id()): bench.py quotes roofline.traffic from it only while
# the library is that build.  Memory guard: a host-side bug once took the GPU boxes down -- every python
# process below runs under `timeout`, and nothing here allocates more than a few GB.
tag=${1:-r06}
mkdir -p gpurun_out/profiles_${tag}
run() {  # run TAG WORKLOAD BATCH STEPS [extra bench args]: trace + PMC passes, then the summary under profiles/
  local name=$1 b=$3
  timeout 900 bash tools/prof_workload.sh "${tag}_${name}_b${b}" "$2" "$3" "$4" "${@:5}" > gpurun_out/prof_${tag}_${name}_b${b}.log 2>&1
  python tools/summarize_profiles.py gpurun_out/prof_${tag}_${name}_b${b} gpurun_out/profiles_${tag}/${tag}_${name}_b${b} > /dev/null 2>&1 || echo "summary failed: ${name} b${b}"
}
run ocean1024 ocean1024 32 1600
run ocean1024 ocean1024 20 1000          # the driver's --steps 20: one 20-step enqueue per launch
run ocean2048 ocean2048 32 320
run ocean4096 ocean4096 32 128
run pond pond 32 3200
run renderer1024 renderer1024 32 640      # 32 consecutive frames per enqueue (mw_ocean_generate_texture_steps_device)
run renderer1024 renderer1024 1 2000      # one GenerateTexture() per call
run renderer1024 renderer1024 4 500 --tiles 4
run direct1000 direct 1 200 --direct-n 1000     # chirp-z: LDS / VALU counters of k_czt
run direct2000 direct 1 100 --direct-n 2000
for n in 12 50 100 1000 2000; do
  timeout 300 python bench.py --workload direct --direct-n $n --steps 200 --warmup 20 2> gpurun_out/${tag}_direct_$n.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_direct_$n.json
done
MW_BENCH_FORCE_TILES=1 timeout 300 python bench.py --steps 640 --warmup 64 --gather --no-cpu-baseline --no-latency 2> gpurun_out/${tag}_tiles.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_ocean1024_tiles_gather.json
timeout 300 python bench.py --steps 640 --warmup 64 --no-cpu-baseline 2> gpurun_out/${tag}_b32.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_ocean1024_b32_steps640.json
for wl in ocean2048 ocean4096; do timeout 400 python bench.py --workload $wl --steps 128 --warmup 32 --no-cpu-baseline --no-latency 2> gpurun_out/${tag}_$wl.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_$wl.json; done
timeout 300 python bench.py --workload pond --steps 3200 --warmup 320 2> gpurun_out/${tag}_pond.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_pond.json
timeout 300 python bench.py --workload renderer1024 --batch 32 --steps 640 --warmup 64 2> gpurun_out/${tag}_renderer.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_renderer1024.json
timeout 300 python bench.py --workload renderer1024 --batch 1 --steps 2000 --warmup 200 --no-cpu-baseline 2> gpurun_out/${tag}_renderer1.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_renderer1024_frame.json
timeout 300 python bench.py --workload renderer1024 --batch 1 --tiles 4 --steps 500 --warmup 50 --no-cpu-baseline 2> gpurun_out/${tag}_renderer4.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_renderer1024_tiles4.json
timeout 300 python bench.py --steps 20 --warmup 5 2> gpurun_out/${tag}_bench_driver.err | tail -1 > gpurun_out/profiles_${tag}/${tag}_bench_ocean1024_driver_k20.json
# one step per call (the frame-at-a-time plan): latency probe, kernel trace of it, cycle stamps + placement
timeout 300 python tools/frame_probe.py 2> /dev/null | tail -1 > gpurun_out/profiles_${tag}/${tag}_frame_probe.json
rm -rf /tmp/fp_${tag}; TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fp_${tag} -o fp --output-format csv -- python tools/frame_probe.py > /dev/null 2>&1
f=$(find /tmp/fp_${tag} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > gpurun_out/profiles_${tag}/${tag}_frame_kernel_stats.csv
bash tools/frame_stamps.sh > /dev/null 2>&1; cp gpurun_out/frame_ab/stamps.txt gpurun_out/profiles_${tag}/${tag}_frame_stamps.txt 2> /dev/null
P=gpurun_out/profiles_${tag}; python tools/bench_summary.py $P/${tag}_bench_direct_12.json $P/${tag}_bench_direct_50.json $P/${tag}_bench_direct_100.json $P/${tag}_bench_direct_1000.json $P/${tag}_bench_direct_2000.json $P/${tag}_bench_ocean1024_b32_steps640.json $P/${tag}_bench_ocean2048.json $P/${tag}_bench_ocean4096.json $P/${tag}_bench_pond.json $P/${tag}_bench_renderer1024.json $P/${tag}_bench_renderer1024_frame.json $P/${tag}_bench_renderer1024_tiles4.json $P/${tag}_bench_ocean1024_driver_k20.json
ls gpurun_out/profiles_${tag}
# back in the container: cp gpurun_out/profiles_${tag}/* profiles/   (only gpurun_out/ travels back from the GPU box)
