#!/bin/bash
bash tools/build_variant.sh or16 -DMW_PT_OR=16 > /dev/null 2>&1
for v in "" variants/or16.so "" variants/or16.so; do for tiles in 1 4; do
MW_LIB=$v timeout 200 python bench.py --workload renderer1024 --tiles $tiles --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('${v:-base} tiles $tiles', '%.2f us/frame' % (d['ms_per_step']*1e3), 'frac', round(d['roofline']['frac'], 3), d.get('parity'))"; done; done
MW_LIB=variants/or16.so timeout 600 python -m pytest tests/test_ocean_renderer.py tests/test_state_and_tiles.py -m gpu -q -x -k "renderer or oceanrenderer or texture or rgba" 2>&1 | tail -3
