#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the pass kernels for kernel variants:
#   tools/pmc_fw.sh "<workload> <batch>" variant1 variant2 ...      ("base" = the in-tree .so)
# prints per kernel: launch us, 2*FETCH and WRITE in bytes per grid point (N and batch parsed from the workload)
set -- $@
wl=$1; b=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" == "base" ]; then lib=""; else lib="variants/$v.so"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcfw_${v}_$c
    MW_LIB=$lib rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcfw_${v}_$c -o pmc -- python bench.py --workload $wl --batch $b --steps $((b * 2)) --warmup $b --preheat-ms 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
  done
  python - "$v" "$wl" "$b" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
v, wl, b = sys.argv[1], sys.argv[2], int(sys.argv[3])
N = int(re.sub(r"\D", "", wl)); pts = N * N * b
acc = defaultdict(lambda: defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmcfw_{v}_{c}/**/pmc_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_pass" in k:
                acc[k][c].append(float(row["Counter_Value"]))
                acc[k]["us"].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    us = sorted(d["us"])[len(d["us"]) // 2]
    fe = 2 * 1024 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]) / pts
    wr = 1024 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"]) / pts
    print(f"{v:12s} {k:28s} {us:8.1f} us  read {fe:6.2f} B/pt  write {wr:6.2f} B/pt  -> {(fe + wr) * pts / us / 1e6:5.2f} TB/s physical")
PY
done
