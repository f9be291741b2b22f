#!/bin/bash
# Beat kernels against the nominal tile and the load pairs per lane: bash scripts/vec_tile_sweep.sh
OUT=gpurun_out/vec_tile_sweep.txt
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 200 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for t in 512 508 504 496; do one "angular P4 tile $t" angular --vec-tile $t; done
for t in 1024 1016 1008 992 768 512; do one "angular P8 tile $t" angular --vec-passes 8 --vec-tile $t; done
for t in 512 508 504 496 1024 1008; do one "euclid P8 tile $t" euclid --vec-tile $t; done
cat $OUT
