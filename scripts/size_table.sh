#!/bin/bash
# HBM-peak fraction of every op kernel against the batch size: bash scripts/size_table.sh
# box / tri in both layouts (tiled: one buffer per segment; plain: one array per operand)
mkdir -p gpurun_out; OUT=gpurun_out/size_table.txt; : > $OUT
for spec in "box tiled" "box plain" "tri tiled" "tri plain" "euclid plain" "angular plain"; do
  set -- $spec; w=$1; layout=$2
  for lg in 16 18 20 22 24 25 26; do
    timeout -k 10 100 python3 bench.py --only $w --layout $layout --log2-ops $lg --steps 200 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $layout 2^$lg', r['ms_per_step'], r['roofline']['frac'])" >> $OUT || exit 1
  done
done
cat $OUT
