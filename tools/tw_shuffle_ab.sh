#!/bin/bash
# A/B of the twiddle-row source in the first radix-8 pass (LDS table vs wave shuffle), 512^2 (pass 2 at 8 points per thread):
# bench line + rocprofv3 kernel stats + LDS counters of both builds -> gpurun_out/tw_ab/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tw_ab; rm -rf $OUT; mkdir -p $OUT
for v in cur twshuffle; do
  for rep in 1 2; do
    MW_LIB=variants/$v.so python bench.py --workload ocean512 --steps 8000 --warmup 64 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_${v}_$rep.json
  done
  MW_LIB=variants/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- python bench.py --workload ocean512 --steps 3200 --no-cpu-baseline > /dev/null 2>&1
  MW_LIB=variants/$v.so rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/pmc_$v -o p -- python bench.py --workload ocean512 --steps 64 --warmup 32 --preheat-ms 0 --no-cpu-baseline --no-parity > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, json
from collections import defaultdict
out = {}
for v in ("cur", "twshuffle"):
    b = [json.load(open(f)) for f in sorted(glob.glob(f"gpurun_out/tw_ab/bench_{v}_*.json"))]
    st = {}
    for row in csv.DictReader(open(glob.glob(f"gpurun_out/tw_ab/trace_{v}/**/t_kernel_stats.csv", recursive=True)[0])):
        if "k_pass" in row["Name"]:
            st[row["Name"].split("(")[0]] = float(row["AverageNs"]) / 1e3
    pm = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"gpurun_out/tw_ab/pmc_{v}/**/p_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "k_pass2" in k:
                pm[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[v] = {"grid_points_per_s": [x["value"] for x in b], "us_per_step": [x["ms_per_step"] * 1e3 for x in b],
              "k_pass2_us_per_32_steps_in_situ": [x["roofline"]["launch_us"] for x in b],
              "rocprof_avg_us": st, "k_pass2_counters_per_launch": {k: {c: sum(v_) / len(v_) for c, v_ in d.items()} for k, d in pm.items()}}
json.dump(out, open("gpurun_out/tw_ab/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
