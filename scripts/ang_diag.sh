#!/bin/bash
# Angular beat kernels against residency caps: default (owning shape), forced walks (21 plain, 22 four sums per LDS round trip),
# chained shape (27) and the diagnostic shapes without walk (24) / without walk and LDS stage (25).  bash scripts/ang_diag.sh [variants]
for v in ${@:-0 21 22 27 24 25}; do for w in 0 3 4 5 6 99; do for p in 0 2 8; do
timeout -k 10 100 python3 bench.py --only angular --variant $v --wgs-per-cu $w --vec-passes $p --steps 200 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('angular variant $v wgs $w passes $p', r['ms_per_step'], r['roofline']['frac'])" || exit 1
done; done; done
