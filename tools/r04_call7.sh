#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
bash tools/build_variant.sh lat_pf0 -DMW_LATENCY_PF=0 > /dev/null 2>&1 &
bash tools/build_variant.sh lat_t -DMW_TIMING -DMW_STAMP_STEP=0 > /dev/null 2>&1 &
wait
for v in "" variants/lat_pf0.so; do MW_LIB=$v timeout 200 python tools/frame_probe.py 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$v', {k: round(v['back_to_back_us_per_step'], 2) for k, v in d.items() if isinstance(v, dict)}, {k: round(v['sync_latency_us_median'], 2) for k, v in d.items() if isinstance(v, dict)})"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/frame_trace -o ft -- python tools/frame_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04d/frame_trace/**/ft_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
N=1024 B=1 HS=1 MW_LIB=variants/lat_t.so timeout 300 python tools/phase_timing.py 2>&1 | grep -v amdgpu.ids | tail -48
timeout 600 python -m pytest tests/test_zz_frame_plan.py -m gpu -q -x 2>&1 | tail -3
