#!/bin/bash
# Wall-clock A/B runs behind profiles/*_footprint.md (no profiler attached): the box kernel against batch size,
# array spacing and the launched part of a large footprint.  bash scripts/footprint_ab.sh <tag>
TAG=${1:-r02}
OUT=gpurun_out/footprint_ab_$TAG.txt
mkdir -p gpurun_out; : > $OUT
one() { # label, args...
  local label=$1; shift
  timeout -k 10 150 python bench.py --only box --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['box']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
one "2^24" --log2-ops 24
one "2^25" --log2-ops 25
one "2^26" --log2-ops 26
one "2^26 arrays, first 2^24 ops launched" --log2-ops 26 --launch-ops 16777216
one "2^26 arrays, first 2^25 ops launched" --log2-ops 26 --launch-ops 33554432
one "2^26 one pool, natural spacing" --log2-ops 26 --stagger-kib -1
one "2^26 one pool, +4 KiB per array" --log2-ops 26 --stagger-kib 4
one "2^26 one pool, +68 KiB per array" --log2-ops 26 --stagger-kib 68
one "2^26 one pool, +2052 KiB per array" --log2-ops 26 --stagger-kib 2052
one "2^26 one pool, +33796 KiB per array" --log2-ops 26 --stagger-kib 33796
one "60000000 ops" --ops 60000000
one "2^24 one pool, +2052 KiB per array" --log2-ops 24 --stagger-kib 2052
one "2^27" --log2-ops 27
cat $OUT
