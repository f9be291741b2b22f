#!/bin/bash
# owning against chained shape of the beat kernels by segment size: bash scripts/beat_shapes.sh
mkdir -p gpurun_out; OUT=gpurun_out/beat_shapes.txt; : > $OUT
for spec in "euclid 0 chained" "euclid 23 owning" "angular 0 owning" "angular 27 chained"; do
  set -- $spec; w=$1; v=$2; name=$3
  for lg in 18 20 22 23 24 25 26; do
    timeout -k 10 100 python3 bench.py --only $w --variant $v --log2-ops $lg --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $name 2^$lg', r['ms_per_step'], r['roofline']['frac'])" >> $OUT || exit 1
  done
done
cat $OUT
