#!/bin/bash
# per-kernel durations and HBM-side bytes of the batched OceanRenderer frame: tools/prof_or_tiles.sh <tiles>
T=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_or_t$T
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python bench.py --workload renderer1024 --steps 500 --tiles $T --no-cpu-baseline > $OUT/stdout.txt 2>&1
grep -E "k_or_" $OUT/trace/r_kernel_stats.csv | cut -d, -f1-4
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --workload renderer1024 --steps 64 --warmup 8 --preheat-ms 0 --tiles $T --no-cpu-baseline > /dev/null 2>&1
done
python - "$OUT" "$T" <<'PY'
import csv, glob, sys
from collections import defaultdict
out, T = sys.argv[1], int(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/pmc_{c}/**/pmc_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_or_pass" in k or "normal_white" in k:
                acc[k][c].append(float(row["Counter_Value"]))
for k, d in acc.items():
    fe = 2 * 1024 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) / (T * 1048576)
    wr = 1024 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) / (T * 1048576)
    print(f"{k:32s} read {fe:6.2f} B/texel  write {wr:6.2f} B/texel")
PY
