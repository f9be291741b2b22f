#!/bin/bash
# angular opcode: the library's choice of shape (variant 0) against owning (23) and chained (27), same process conditions
mkdir -p gpurun_out; OUT=gpurun_out/ang_shape_check.txt; : > $OUT
for rep in 1 2; do for lg in 22 24 26; do for v in 0 23 27; do
  timeout -k 10 200 python3 bench.py --only angular --variant $v --log2-ops $lg --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('angular 2^$lg variant $v', r['ms_per_step'], r['roofline']['frac'])" >> $OUT || exit 1
done; done; done
for rep in 1 2; do for v in 0 23; do
  timeout -k 10 200 python3 bench.py --only mixed --variant $v --log2-ops 24 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed 2^24 variant $v', r['ms_per_step'], r['roofline']['frac'])" >> $OUT || exit 1
done; done
cat $OUT
