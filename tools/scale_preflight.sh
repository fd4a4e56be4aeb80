#!/bin/bash
# tools/scale_preflight.sh -- run ON AN N-GPU BOX before trusting a scaling curve (no such box has been available to
# the builder: nothing below has produced a number yet, and none is quoted anywhere).
#   1. the real multi-device tests (tile gather over xGMI on 2 / 4 / 8 devices; skipped where fewer are visible)
#   2. the driver's command, bench.py --gpus {2,4,8}: its ONE line carries the three N > 1 forms -- one tile per rank (`value`, weak), the same
#      with the library's RCCL gather of every batch's last step (`with_gather`), one ocean with the time-steps sharded (`strong`) -- and
#      `rccl_ranks` / `per_rank`; checked here: n_gpus, the tile API really in use (library-owned RCCL communicator with world ranks),
#      parity gate green on rank 0, per-GPU rate within 10 % of the N = 1 headline of
#      the SAME box (the path has no data-path collective: anything else is a placement or clock problem worth knowing before the curve).
# usage: bash tools/scale_preflight.sh [max_gpus] [steps]        -> gpurun_out/scale_preflight.txt (+ one JSON line per run beside it)
cd "$(dirname "$0")/.."
MAXG=${1:-8}; K=${2:-640}
OUT=gpurun_out/scale_preflight; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python -c "import sys; sys.path.insert(0,'mistral-water_amd'); import mistral_water as mw; print(mw.lib().mw_device_count())")
echo "devices visible: $NDEV" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_state_and_tiles.py -q -m gpu -k "two_devices or rccl_gather or per_process_form" 2>&1 | tail -3 | tee -a $OUT/summary.txt
timeout 600 python bench.py --workload ocean1024 --steps $K --warmup 64 --no-cpu-baseline --no-latency 2> $OUT/n1.err | tail -1 > $OUT/n1.json
port=29610
for n in 2 4 8; do
  [ $n -le $MAXG ] && [ $n -le $NDEV ] || continue
  port=$((port + 1))
  # the driver's own command: ONE line carries value (tiles), with_gather, strong, rccl_ranks and per-rank rates
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
    bench.py --gpus $n --steps $K --warmup 64 --no-cpu-baseline --no-latency 2> $OUT/n$n.err | grep '^{' | tail -1 > $OUT/n${n}_default.json
done
python - "$OUT" <<'PY' | tee -a $OUT/summary.txt
import glob, json, os, sys
out = sys.argv[1]
def load(p):
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
base = load(os.path.join(out, "n1.json"))
print("N=1 headline of this box: %.4g pts/s" % base.get("value", float("nan")), base.get("parity"))
ok = True
for p in sorted(glob.glob(os.path.join(out, "n[248]_default.json"))):
    d = load(p); tag = os.path.basename(p)[:-5]
    if "error" in d:
        print(tag, "NO RESULT LINE", d["error"]); ok = False; continue
    n = int(tag[1])
    per_gpu = d["value"] / n
    g, st = d.get("with_gather") or {}, d.get("strong") or {}
    checks = {
        "n_gpus": d["n_gpus"] == n,
        "scaling": d["scaling"] == "weak",
        "parity": (d.get("parity") or "").startswith("ok"),
        "tile_api": d["config"]["api"].startswith("mw_tiles_"),
        "rccl_ranks": d.get("rccl_ranks") == n,
        "tiles": d["config"]["tiles"] == n,
        "per_gpu_within_10pct_of_n1": abs(per_gpu / base["value"] - 1.0) < 0.10 if "value" in base else False,
        "ranks_even": d["per_rank"]["slowest_over_fastest"] < 1.10,
        "gather_ran": g.get("gathers", 0) > 0 and g.get("rccl_ranks") == n,
        "gather_within_5pct": g.get("value", 0) > 0.95 * d["value"],
        "strong_ran": st.get("scaling") == "strong" and st.get("config", {}).get("parallelism") == f"steps{n}",
    }
    bad = [k for k, v in checks.items() if not v]
    ok = ok and not bad
    print("%-12s %.4g pts/s  per GPU %.4g (%.3f of N=1)  eff %.3f  with gather %.3f  strong %.4g pts/s  %s" % (
        tag, d["value"], per_gpu, per_gpu / base.get("value", 1), d["value"] / (n * base.get("value", 1)), g.get("relative_to_value", float("nan")),
        st.get("value", float("nan")), "OK" if not bad else "FAILED: " + ", ".join(bad)))
print("PREFLIGHT", "OK" if ok else "FAILED")
PY
