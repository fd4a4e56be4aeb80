#!/bin/bash
# Round-2 rocprofv3 evidence, every workload at the batch bench.py reports it with (run through gpurun):
#   tools/profile_r2.sh            ->  gpurun_out/prof_r2_<workload>/ ; condense with tools/summarize_profiles.py
bash tools/prof_workload.sh r2_ocean1024 ocean1024 32 1600
bash tools/prof_workload.sh r2_ocean2048 ocean2048 32 320
bash tools/prof_workload.sh r2_ocean4096 ocean4096 32 128
bash tools/prof_workload.sh r2_renderer1024 renderer1024 1 2000
bash tools/prof_workload.sh r2_pond pond 32 3200
