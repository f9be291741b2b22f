#!/bin/bash
# Tiled box / triangle kernels against the residency cap at small and medium batch sizes (where the cap should start to apply):
#   bash scripts/small_tiled.sh [log2 sizes ...]
for w in box tri; do for lg in ${@:-17 18 19 20 21 22 23}; do for c in 0 99 3 4 5; do
timeout -k 10 100 python3 bench.py --only $w --log2-ops $lg --wgs-per-cu $c --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w 2^$lg wgs $c', r['ms_per_step'], r['roofline']['frac'])" || exit 1
done; done; done
