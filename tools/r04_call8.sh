#!/bin/bash
timeout 600 python -m pytest tests/test_zz_frame_plan.py -m gpu -q -x 2>&1 | tail -5
for fp in 1 0; do MW_FRAME_PIPELINE=$fp timeout 200 python tools/frame_probe.py 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('pipeline $fp', {k: (round(v['back_to_back_us_per_step'], 2), round(v['host_enqueue_us_per_call'], 2), round(v['sync_latency_us_median'], 2)) for k, v in d.items() if isinstance(v, dict)})"; done
MW_FRAME_PIPELINE=1 timeout 200 python tools/frame_probe.py --n 512 2>/dev/null | tail -1
MW_FRAME_PIPELINE=0 timeout 200 python tools/frame_probe.py --n 512 2>/dev/null | tail -1
MW_FRAME_PIPELINE=1 timeout 200 python tools/frame_probe.py --n 2048 2>/dev/null | tail -1
MW_FRAME_PIPELINE=0 timeout 200 python tools/frame_probe.py --n 2048 2>/dev/null | tail -1
