#!/bin/bash
out=gpurun_out/r04b; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/valu_probe tools/valu_probe.hip && timeout 120 /tmp/valu_probe > $out/valu_probe.txt 2>&1; cat $out/valu_probe.txt
timeout 300 python tools/memguard_probe.py > $out/memguard_probe.txt 2>&1; grep -v amdgpu.ids $out/memguard_probe.txt | tail -20
timeout 200 python tools/frame_probe.py 2>/dev/null | tail -1 | tee $out/frame_probe_noguard.json
timeout 200 python tools/frame_probe.py --guard 2>/dev/null | tail -1 | tee $out/frame_probe_guard.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/frame_trace -o ft -- python tools/frame_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04b/frame_trace/**/ft_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
timeout 900 python -m pytest tests -m gpu -q -x -k "whitecap_stage or baseline_config3 or leaves_a_caller or reinit or bench_times or literal" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8
timeout 300 python bench.py --steps 20 --warmup 5 2> $out/driver.err | tail -1 > $out/bench_driver_k20.json
python tools/bench_summary.py $out/bench_driver_k20.json
python -c "
import json; d=json.loads(open('$out/bench_driver_k20.json').read())
print({k: d[k] for k in ('ms_per_step','ms_per_step_min','ms_per_step_max','event_ms_per_step','wall_over_events','region_ms','event_region_ms','single_step_us')})"
