#!/bin/bash
# tools/frame_variants.sh "name:-Dflag -Dflag" ...   (GPU box) builds each variant, checks frame-plan parity, prints back-to-back latency and kernel times
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/frame_ab; mkdir -p $O; : > $O/variants.txt
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  if [ "$name" == base ]; then lib=""; else bash tools/build_variant.sh $name $flags > /dev/null 2>&1; lib=variants/$name.so; fi
  par=$(MW_LIB=$lib timeout 600 python -m pytest tests/test_zz_frame_plan.py -x -q -m gpu -k single_step 2>&1 | tail -1)
  b2b=$(MW_LIB=$lib timeout 300 python tools/frame_probe.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['own_stream']['back_to_back_us_per_step'], '%.2f' % d['torch_null_stream']['back_to_back_us_per_step'])")
  rm -rf /tmp/fp_$name
  MW_LIB=$lib timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fp_$name -o fp --output-format csv -- python tools/frame_probe.py > /dev/null 2>&1
  f=$(find /tmp/fp_$name -name "*kernel_stats.csv" | head -1)
  ks=$(python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(" ".join("%s=%.2fus" % (r["Name"].split("<")[0].replace("void ", ""), float(r["AverageNs"]) / 1e3) for r in rows[:2]))
PY
)
  echo "$name [$flags] | parity: $par | back-to-back us: $b2b | $ks" | tee -a $O/variants.txt
done
