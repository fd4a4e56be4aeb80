#!/bin/bash
for i in 1 2; do
for nc in 0 1; do
  MW_TILES_COPY_NOCU=$nc MW_BENCH_FORCE_TILES=1 timeout 300 python bench.py --workload ocean1024 --steps 640 --warmup 64 --gather --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); g=d['with_gather']; print('NOCU=$nc: without %.4g with %.4g  ratio %.3f' % (d['value'], g['value'], g['value']/d['value']), g['region_ms'], d['parity'][:2])"
done; done
MW_TILES_COPY_NOCU=1 timeout 600 python -m pytest tests/test_state_and_tiles.py -m gpu -q -k "gather" 2>&1 | tail -2
