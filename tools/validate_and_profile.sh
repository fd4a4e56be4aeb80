#!/bin/bash
# One gpurun call: the whole GPU test tier, then every rocprofv3 pass and bench line of the round (tools/profile_round.sh).
#   gpurun --timeout 2400 -- 'bash tools/validate_and_profile.sh r06'      then: cp gpurun_out/profiles_r06/* profiles/
tag=${1:-r06}
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 > gpurun_out/$tag/pytest_gpu.txt
tail -6 gpurun_out/$tag/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round.sh $tag 2>&1 | tail -30
