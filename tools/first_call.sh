#!/bin/bash
# The first gpurun call of a round after one that ended without GPU access (round 3): validate, then measure everything that was
# built blind.  ~12 min.  Every python process runs under `timeout`; nothing here allocates more than ~10 GB of host memory.
#   gpurun --timeout 1500 -- 'bash tools/first_call.sh r04'
tag=${1:-r04}
out=gpurun_out/$tag
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > $out/pytest.txt
tail -12 $out/pytest.txt
# frame-at-a-time plan: on / off (run-time switch), one step per call
for lp in 1 0; do
  MW_LATENCY_PLAN=$lp timeout 300 python bench.py --steps 640 --warmup 64 --no-cpu-baseline 2> $out/frame_lp$lp.err | tail -1 > $out/frame_lp$lp.json
done
python - $out <<'PY'
import json, sys
for lp in (1, 0):
    try:
        d = json.loads(open(f"{sys.argv[1]}/frame_lp{lp}.json").read())
        print("MW_LATENCY_PLAN", lp, "device us/step", d["frame_at_a_time"]["device_us_per_step"], "batched value", d["value"], "p2 frac", d["roofline"]["frac"])
    except Exception as e:
        print("frame plan", lp, "FAILED", e)
PY
# direct sum: GEMM form (default) and chirp-z form (opt-in)
bash tools/bench_direct.sh $out/direct_gemm
MW_DIRECT_CZT=1 bash tools/bench_direct.sh $out/direct_czt
# 4096^2 and 2048^2 lines of the final build
for wl in ocean2048 ocean4096; do
  timeout 400 python bench.py --workload $wl --steps 128 --warmup 32 --no-cpu-baseline --no-latency 2> $out/$wl.err | tail -1 > $out/$wl.json
done
python tools/bench_summary.py $out/ocean2048.json $out/ocean4096.json
# 4096^2 pass 2 with TWO rows per workgroup (4 waves, 76 KiB: two workgroups per CU like the 2048^2 plan; half-line reads and a halo row
# per two rows against it) -- A/B against the 4-row plan, parity gate on (compiled 0 spills: 207 / 185 / 217 VGPRs)
(bash tools/build_variant.sh r2x2_pf1 -DMW_R2_4096=2 > /dev/null 2>&1 & bash tools/build_variant.sh r2x2_pf0 -DMW_R2_4096=2 -DMW_PF_4096=0 > /dev/null 2>&1 & bash tools/build_variant.sh r2x2_pf0_he -DMW_R2_4096=2 -DMW_PF_4096=0 -DMW_HS_HALO_EARLY_4096=1 > /dev/null 2>&1 & wait)
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean4096 32 256" base r2x2_pf1 r2x2_pf0 r2x2_pf0_he 2>&1 | tee $out/ab_r2x2_4096.txt
