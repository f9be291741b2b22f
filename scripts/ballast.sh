#!/bin/bash
# does other resident memory change a beat kernel's rate?  bash scripts/ballast.sh
mkdir -p gpurun_out; OUT=gpurun_out/ballast.txt; : > $OUT
for w in euclid angular; do for g in 0 8.5 20; do
  timeout -k 10 100 python3 bench.py --only $w --log2-ops 26 --steps 100 --warmup 20 --no-cpu-baseline --ballast-gib $g 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w 2^26 ballast $g GiB', r['ms_per_step'], r['roofline']['frac'])" >> $OUT || exit 1
done; done
cat $OUT
