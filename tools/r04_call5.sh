#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
for cfg in "8 2 0" "8 4 0" "8 2 1" "8 4 1" "4 4 0" "16 2 0"; do set -- $cfg; bash tools/build_variant.sh w_c$1d$2x$3 -DMW_P1W_CHUNK=$1 -DMW_P1W_DEPTH=$2 -DMW_P1W_XPREFETCH=$3 > /dev/null 2>&1 & done; wait
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean4096 32 128" w_c8d2x0 w_c8d4x0 w_c8d2x1 w_c8d4x1 w_c4d4x0 w_c16d2x0 2>&1 | tee $out/ab_p1wave_ring.txt
