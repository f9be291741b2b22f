#!/bin/bash
# A/B of the beat kernels: the current library (default walk, variants 21 / 22) against a build with another kernels.hip
# (RT_DP_LIB), interleaved, 200 timed launches each.  bash scripts/vec_ab.sh [other library]
OUT=gpurun_out/vec_ab.txt
OTHER=${1:-$PWD/raytracer_amd/lib/ab/librt_dp_r01vec.so}
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 200 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for rep in 1 2; do
  for w in angular euclid; do
    one "$w default" $w
    [ -f "$OTHER" ] && RT_DP_LIB=$OTHER one "$w other library" $w
    one "$w variant 21" $w --variant 21
    one "$w variant 22" $w --variant 22
    one "$w passes 4" $w --vec-passes 4
    one "$w passes 8" $w --vec-passes 8
  done
done
cat $OUT
