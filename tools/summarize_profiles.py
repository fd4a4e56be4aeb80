"""Condenses a tools/prof_workload.sh output directory into the small files committed under profiles/.
usage: python tools/summarize_profiles.py gpurun_out/prof_<tag> profiles/<name>
  <name>_kernel_stats.csv   rocprofv3's own --stats table (means over EVERY launch, clock-ramp and 1-step launches included)
  <name>_kernel_pcts.json   per kernel, from the kernel trace: the full-size launches in time order, the first quarter dropped as
                            warm-up (bench.py's preheat + warm-up steps run at ramping clocks), then mean / median / p10 / p90 / min /
                            max -- what bench.py's roofline.launch_us(_stats) of the same command should reproduce
  <name>_pmc.json           counter means per full-size launch + the bench line of the traced run"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), dst + "_kernel_stats.csv")
# ---- percentiles per kernel from the per-dispatch trace -------------------------------------------------------------------
def _pcts(v):
    v = sorted(v)
    n = len(v)
    q = lambda f: v[min(n - 1, int(round(f * (n - 1))))]
    return {"n": n, "mean_us": sum(v) / n / 1e3, "median_us": q(0.5) / 1e3, "p10_us": q(0.1) / 1e3, "p90_us": q(0.9) / 1e3,
            "min_us": v[0] / 1e3, "max_us": v[-1] / 1e3}
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
per = defaultdict(lambda: defaultdict(list))   # kernel -> launch geometry -> [(start, duration)]
for f in trace:
    for row in csv.DictReader(open(f)):
        geom = (row.get("Grid_Size_X"), row.get("Grid_Size_Y"), row.get("Grid_Size_Z"))
        per[row["Kernel_Name"].split("(")[0]][geom].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
kp = {}
for k, by_geom in per.items():
    # bench.py launches a kernel at several sizes (the timed batch size, 1-step launches of the parity gate and the latency figure, 32-step
    # context launches): the launches of the TIMED size are the geometry with the most launches -- one grid size, one batch size
    geom, rows = max(by_geom.items(), key=lambda kv: len(kv[1]))
    if len(rows) < 8:
        continue
    rows.sort()
    durs = [d for _, d in rows]
    steady = durs[len(durs) // 4:]                             # time order: the first quarter = preheat + warm-up at ramping clocks
    kp[k] = {"grid_size": geom, "all_launches_of_this_size": _pcts(durs), "after_warmup": _pcts(steady),
             "other_sizes": {"x".join(str(g) for g in gg): len(v) for gg, v in by_geom.items() if gg != geom},
             "note": "the launch geometry with the most launches (= the timed batch size); after_warmup = those launches in time order with "
                     "the first 25 % dropped"}
if kp:
    json.dump({"source": src, "kernels": kp}, open(dst + "_kernel_pcts.json", "w"), indent=1)
bench_line = [l for l in open(os.path.join(src, "bench_stdout.txt")) if l.startswith("{")]
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if any(t in k for t in ("k_pass", "gerstner", "k_or_", "k_pond", "k_gemm", "k_direct", "k_czt")):
            pmc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            pmc[k]["_dur_ns_" + row["Counter_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            pmc[k]["_grid_" + row["Counter_Name"]].append(row.get("Grid_Size"))
line = json.loads(bench_line[-1]) if bench_line else None
out = {"source": src, "build_id": (line or {}).get("build_id"), "bench_line": line, "pmc_mean_per_launch": {}}
for k, d in pmc.items():
    # drop the short warm-up launches: keep launches within 15 % of the median duration of that counter pass
    o = {}
    for c, vals in d.items():
        if c.startswith("_dur_ns_") or c.startswith("_grid_"):
            continue
        durs = d["_dur_ns_" + c]
        # the launches of the TIMED size first: the launch geometry with the most launches, as in the kernel trace above (a 20-step run also
        # holds bench.py's 32-step context launches, which are LONGER: the duration cluster below alone once averaged those instead)
        grids = d["_grid_" + c]
        if any(g is not None for g in grids):
            from collections import Counter
            top = Counter(grids).most_common(1)[0][0]
            sel = [i for i, g in enumerate(grids) if g == top]
            vals = [vals[i] for i in sel]
            durs = [durs[i] for i in sel]
        # bench.py also issues 1-step launches (parity gate, the single-step latency figure): the per-launch means are those
        # of the FULL-batch launches, i.e. of the long-duration cluster
        big = sorted(t for t in durs if t >= 0.5 * max(durs))
        med = big[len(big) // 2]
        keep = [v for v, t in zip(vals, durs) if abs(t - med) <= 0.15 * med]   # launches of another size would skew a per-launch mean
        o[c] = sum(keep) / len(keep)
        o.setdefault("_launch_us", {})[c] = med / 1e3
    out["pmc_mean_per_launch"][k] = o
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
print(json.dumps(out["pmc_mean_per_launch"], indent=1)[:3000])
