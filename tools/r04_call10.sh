#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
(bash tools/build_variant.sh r2_he -DMW_R2_4096=2 -DMW_PF_4096=0 -DMW_HS_HALO_EARLY_4096=1 > /dev/null 2>&1 &
 bash tools/build_variant.sh r2_he_pf1 -DMW_R2_4096=2 -DMW_PF_4096=1 -DMW_HS_HALO_EARLY_4096=1 > /dev/null 2>&1 &
 bash tools/build_variant.sh r4_he -DMW_PF_4096=0 -DMW_HS_HALO_EARLY_4096=1 > /dev/null 2>&1 &
 bash tools/build_variant.sh r4_pf0 -DMW_PF_4096=0 > /dev/null 2>&1 & wait)
for v in r2_he r2_he_pf1 r4_he r4_pf0; do python3 tools/resource_report.py variants/$v.res "k_pass2_hs<4096" | tail -1; done
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean4096 32 256" base r2_he r2_he_pf1 r4_he r4_pf0 base 2>&1 | tee $out/ab_r2_lir_4096.txt
