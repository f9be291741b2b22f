#!/bin/bash
# angular opcode on short segments (chained shape there): 16-byte load pairs per lane and sub-tile (4 = default, 2, 8)
mkdir -p gpurun_out; OUT=gpurun_out/angular_passes.txt; : > $OUT
for rep in 1 2; do for lg in 18 20 22 23; do for p in 4 2 8; do
  timeout -k 10 100 python3 bench.py --only angular --vec-passes $p --log2-ops $lg --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('angular 2^$lg passes $p', r['ms_per_step'], r['roofline']['frac'])" >> $OUT || exit 1
done; done; done
cat $OUT
