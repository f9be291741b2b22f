OUT=gpurun_out/vec_epl.txt
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 200 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
one "angular 16-byte loads P4" angular
for p in 4 8 16; do
  for c in 99 8 6 4; do one "angular 8-byte loads P$p cap $c" angular --variant 26 --vec-passes $p --wgs-per-cu $c; done
done
cat $OUT
