OUT=gpurun_out/vec_diag.txt
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 200 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for w in angular euclid; do
  one "$w full" $w
  one "$w no walk (24)" $w --variant 24
  one "$w no walk, no LDS (25)" $w --variant 25
  one "$w no walk (24) cap 4" $w --variant 24 --wgs-per-cu 4
  one "$w no walk, no LDS (25) cap 4" $w --variant 25 --wgs-per-cu 4
  one "$w no walk, no LDS (25) cap 6" $w --variant 25 --wgs-per-cu 6
done
cat $OUT
