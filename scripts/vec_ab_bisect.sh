#!/bin/bash
# Interleaved A/B of the beat kernels over several builds of the library: bash scripts/vec_ab_bisect.sh <label=path> ...
OUT=gpurun_out/vec_ab_bisect.txt
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 200 --warmup 100 --no-cpu-baseline --log2-ops ${LOG2:-24} "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for rep in 1 2; do
  for w in angular euclid; do
    one "$w current" $w
    for spec in "$@"; do RT_DP_LIB=$PWD/${spec#*=} one "$w ${spec%%=*}" $w; done
  done
done
cat $OUT
