OUT=gpurun_out/footprint_ab3_r02.txt
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; local w=$2; shift; shift
  timeout -k 10 150 python bench.py --only $w --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['$w']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
for w in 3 4 5 7 99; do one "box 2^26 wgs $w" box --log2-ops 26 --wgs-per-cu $w; done
one "box 2^26 V2" box --log2-ops 26 --ops-per-lane 2
one "box 2^26 V2 wgs4" box --log2-ops 26 --ops-per-lane 2 --wgs-per-cu 4
one "box 2^26 no nt" box --log2-ops 26 --nontemporal 2
one "box 2^26 no nt wgs 99" box --log2-ops 26 --nontemporal 2 --wgs-per-cu 99
one "tri 2^24" tri --log2-ops 24
one "tri 2^25" tri --log2-ops 25
one "tri 2^26" tri --log2-ops 26
one "tri 2^26 wgs4" tri --log2-ops 26 --wgs-per-cu 4
one "tri 2^26 wgs7" tri --log2-ops 26 --wgs-per-cu 7
one "tri 2^27" tri --log2-ops 27
cat $OUT
