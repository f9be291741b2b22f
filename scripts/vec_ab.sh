#!/bin/bash
# A/B of the beat kernels in one build: default (chained tiles) against variant 23 (whole-pair ownership, no workspace),
# and the load pairs per lane; optionally another build of the library (RT_DP_LIB).  bash scripts/vec_ab.sh [other library]
OUT=gpurun_out/vec_ab.txt
OTHER=$1
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 200 --warmup 100 --no-cpu-baseline --log2-ops ${LOG2:-24} "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for rep in 1 2; do
  for w in angular euclid; do
    one "$w chained (default)" $w
    one "$w variant 23 (pair ownership)" $w --variant 23
    [ -n "$OTHER" ] && [ -f "$OTHER" ] && RT_DP_LIB=$OTHER one "$w other library" $w
    for p in 2 4 8; do one "$w chained, passes $p" $w --vec-passes $p; done
  done
done
cat $OUT
