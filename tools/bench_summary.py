"""Prints one line per bench JSON file: value, us/step, roofline achieved / frac, parity (tools/bench_direct.sh, tools/profile_round.sh)."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        r = d["roofline"]
        cpu = (d.get("cpu_baseline") or {}).get("value")
        print(path, "%.4g %s" % (d["value"], d["unit"]), "%.2f us/step" % (d["ms_per_step"] * 1e3),
              "%s %.4g %s frac %.3f" % (r.get("kernel"), r["achieved"], r["unit"], r["frac"]),
              [round(k.get("us_per_launch", k.get("us_per_step", 0)), 1) for k in (r.get("kernels") or [])],
              "parity", (d.get("parity") or "none")[:2], "cpu", cpu)
    except Exception as e:  # noqa: BLE001
        print(path, "FAILED", e, open(path).read()[-400:])
