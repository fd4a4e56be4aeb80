#!/bin/bash
# lone-frame stagger sweep: every second workgroup starts late; back-to-back us per step (tools/frame_probe.py, own stream)
O=gpurun_out/r05c8; mkdir -p $O
probe() { python tools/frame_probe.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f %.2f' % (d['own_stream']['back_to_back_us_per_step'], d['torch_null_stream']['back_to_back_us_per_step']))"; }
for rep in 1 2; do
echo "base $(probe)"
for s2 in 2 4 6 8 12; do echo "S2=$s2 $(MW_FRAME_STAGGER2=$s2 probe)"; done
for s1 in 2 4 6 8; do echo "S1=$s1 $(MW_FRAME_STAGGER1=$s1 probe)"; done
for s1 in 4,5 6,5 4,8; do echo "S1=$s1 $(MW_FRAME_STAGGER1=$s1 probe)"; done
for s2 in 4,3 6,3 6,7; do echo "S2=$s2 $(MW_FRAME_STAGGER2=$s2 probe)"; done
echo "S1=4 S2=6 $(MW_FRAME_STAGGER1=4 MW_FRAME_STAGGER2=6 probe)"
echo "S1=6 S2=8 $(MW_FRAME_STAGGER1=6 MW_FRAME_STAGGER2=8 probe)"
done 2>&1 | tee $O/stagger.txt
MW_FRAME_STAGGER1=4 MW_FRAME_STAGGER2=6 timeout 600 python -m pytest tests/test_zz_frame_plan.py -m gpu -q -k "single_step" 2>&1 | tail -2
