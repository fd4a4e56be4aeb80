"""Summarise rocprofv3 rocpd (.db) output: per-kernel stats and PMC counters per kernel.
usage: python tools/rocpd_summary.py <results.db> [...]"""
import sqlite3, sys, re
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n[:60]


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("==", path)
    want = ["name", "start", "end", "grid_x", "grid_y", "workgroup_x", "lds_size", "scratch_size", "vgpr_count", "sgpr_count"]
    have = [c for c in want if c in cols]
    if "name" not in cols:
        print(cols)
        continue
    rows = cur.execute(f"select {', '.join(have)} from kernels").fetchall()
    agg = defaultdict(list)
    meta = {}
    for r in rows:
        agg[short(r[0])].append(r[2] - r[1])
        meta[short(r[0])] = r[3:]
    print(f"{'kernel':62s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>9s}  {have[3:]}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:62s} {len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {sum(v)/1e6:9.3f}  {meta[k]}")
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        cc = cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall() if "counter_name" in ccols else []
        if not cc and ccols:
            print("counters_collection cols:", ccols)
    except Exception as e:
        cc = []
    if cc:
        acc = defaultdict(lambda: defaultdict(list))
        for kn, cn, val in cc:
            acc[short(kn)][cn].append(val)
        for kn, d in acc.items():
            print("  ", kn, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
