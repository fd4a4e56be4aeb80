#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
(bash tools/build_variant.sh m2k_pf1 -DMW_SPLIT_SLOPES_4096=2 -DMW_KEEP_T1_MAX_N=4096 > /dev/null 2>&1 &
 bash tools/build_variant.sh m2k_pf0 -DMW_SPLIT_SLOPES_4096=2 -DMW_KEEP_T1_MAX_N=4096 -DMW_PF_4096=0 > /dev/null 2>&1 &
 bash tools/build_variant.sh m2k_pf0_he -DMW_SPLIT_SLOPES_4096=2 -DMW_KEEP_T1_MAX_N=4096 -DMW_PF_4096=0 -DMW_HS_HALO_EARLY_4096=1 > /dev/null 2>&1 &
 bash tools/build_variant.sh m2_pf1 -DMW_SPLIT_SLOPES_4096=2 > /dev/null 2>&1 & wait)
ABV_EXTRA="--no-latency" bash tools/abv.sh "ocean4096 32 256" base m2k_pf1 m2k_pf0 m2k_pf0_he m2_pf1 base m2k_pf1 2>&1 | tee $out/ab_keep_t1_4096.txt
