"""Condenses a `hipcc -Rpass-analysis=kernel-resource-usage` log: VGPRs / spills / scratch / occupancy / LDS per kernel.
usage: python tools/resource_report.py variants/NAME.res [name filter ...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
filt = sys.argv[2:] or ["k_pass", "k_or_", "k_gerstner", "k_pond"]
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = body.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in body:
        k, v = body.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().splitlines() if names else []
for fn, d in zip(names, dem):
    short = re.sub(r"^void ", "", d.split("(")[0])
    if not any(f in short for f in filt):
        continue
    r = rows[fn]
    print(f"{short:34s} VGPR {r.get('VGPRs','?'):>3s} spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} "
          f"occ {r.get('Occupancy [waves/SIMD]','?'):>2s} SGPR {r.get('TotalSGPRs','?'):>3s} LDS {r.get('LDS Size [bytes/block]','?')}")
