#!/bin/bash
# kernel durations of the beat kernels on small segments: bash scripts/small_beats.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/small_beats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in euclid angular; do for lg in 14 16 20 22; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${w}_$lg -- python3 $ROOT/bench.py --only $w --log2-ops $lg --steps 50 --warmup 10 --no-cpu-baseline > $OUT/${w}_$lg.log 2>&1 || exit 1
  f=$(find $OUT/${w}_$lg -name '*kernel_stats.csv' | head -1)
  echo "$w 2^$lg: $(grep -E 'k_vec' $f | cut -d, -f1-4 | tr '\n' ' ')"
done; done | tee $OUT/summary.txt
