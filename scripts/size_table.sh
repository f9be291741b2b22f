#!/bin/bash
# HBM-peak fraction of every op kernel against the batch size (the table in DESIGN.md): bash scripts/size_table.sh
for w in box tri euclid angular; do
  for lg in 20 22 24 25 26; do
    timeout -k 10 100 python bench.py --only $w --log2-ops $lg --steps 200 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w 2^$lg', r['ms_per_step'], r['roofline']['frac'])" || exit 1
  done
done
