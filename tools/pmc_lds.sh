#!/bin/bash
# LDS counters of the pass kernels for kernel variants:  tools/pmc_lds.sh "<workload> <batch>" variant1 variant2 ...  ("base" = in-tree .so)
# prints per kernel: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of LDS-array cycles lost to conflicts), SQ_INSTS_LDS, SQ_WAIT_INST_LDS
set -- $@
wl=$1; b=$2; shift 2
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  if [ "$v" == "base" ]; then lib=""; else lib="variants/$v.so"; fi
  rm -rf gpurun_out/pmclds_$v
  MW_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmclds_$v -o pmc -- python bench.py --workload $wl --batch $b --steps $((b * 2)) --warmup $b --preheat-ms 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
  python - "$v" <<'PY'
import csv, glob, sys
from collections import defaultdict
v = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"gpurun_out/pmclds_{v}/**/pmc_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_pass" in k or "k_or_" in k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(x) / len(x) for c, x in d.items()}
    print(f"{v:10s} {k:30s} conflict/active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):6.3f}  active {m.get('SQ_LDS_IDX_ACTIVE', 0):.3e}  insts {m.get('SQ_INSTS_LDS', 0):.3e}  wait {m.get('SQ_WAIT_INST_LDS', 0):.3e}")
PY
done
