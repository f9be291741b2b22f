OUT=gpurun_out/small_caps.txt
: > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 300 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for lg in 20 22 23; do
for w in box tri; do
  for c in 3 4 5 6 8 99; do one "$w 2^$lg cap $c" $w --log2-ops $lg --wgs-per-cu $c; done
  one "$w 2^$lg V2 default cap" $w --log2-ops $lg --ops-per-lane 2
done
done
cat $OUT
