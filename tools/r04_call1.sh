#!/bin/bash
# Round 4, first GPU call: memory-guard probe, full validation, the blind-built experiments of round 3, then the profiles of this build.
out=gpurun_out/r04; mkdir -p $out
timeout 600 python tools/memguard_probe.py > $out/memguard_probe.txt 2>&1; cat $out/memguard_probe.txt | tail -30
bash tools/first_call.sh r04
bash tools/profile_round.sh r04 2>&1 | tail -40
