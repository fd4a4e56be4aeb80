"""Condenses the renderer/pond PMC passes of tools/prof_workload.sh: mean 2*FETCH_SIZE and WRITE_SIZE (KiB) per launch and kernel.
usage: python tools/summarize_other_pmc.py gpurun_out/prof_<tag> profiles/r01_other_pmc.json"""
import csv, glob, json, os, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
out = {}
for w in ("renderer1024", "pond"):
    acc = defaultdict(lambda: defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(src, f"pmc_{w}_{c}", "pmc_counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0]
                if any(t in k for t in ("k_or_", "gerstner")):
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[w] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    for k, d in out[w].items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
