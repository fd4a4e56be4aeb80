#!/bin/bash
# tools/sweep.sh "<workload> <batch> <steps>" "name:-Dflags" ...  (GPU box) builds each variant there (variants/ does not travel) and runs tools/abv.sh
# over base + all of them, twice, interleaved (one box: the A/B is valid; box-to-box spread is +-3 %)
cd "$(dirname "$0")/.."
cfg=$1; shift
names=(base)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  bash tools/build_variant.sh $name $flags 2>&1 | grep -E "error|hs<1024|hs<4096|hs<2048" | head -3
  names+=($name)
done
for i in 1 2; do ABV_EXTRA="--no-latency" bash tools/abv.sh "$cfg" "${names[@]}"; done
