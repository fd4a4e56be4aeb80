#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q -x -k "gerstner or pond" 2>&1 | tail -3
for rep in 1 2 3; do timeout 200 python bench.py --workload pond --steps 3840 --warmup 480 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('%.4g vertices/s' % d['value'], '%.3f us/step' % (d['ms_per_step']*1e3), 'frac', round(d['roofline']['frac'], 3), d['parity'][:2])"; done
