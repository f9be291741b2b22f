OUT=gpurun_out/footprint_ab2_r02.txt
mkdir -p gpurun_out; : > $OUT
one() { local label=$1; shift
  timeout -k 10 150 python bench.py --only box --steps 100 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s kernel_ms %8.4f frac %.4f' % ('$label', r['per_opcode']['box']['kernel_ms'], r['roofline']['frac']))" >> $OUT || { echo "FAILED $label" >> $OUT; exit 1; }
}
one "2^24 pool natural" --log2-ops 24 --stagger-kib -1
for s in 1 4 16 64 256 1024 4096 16384; do one "2^24 pool +$s KiB" --log2-ops 24 --stagger-kib $s; done
for s in 1 16 256 1024 4096 16384 65536 65540 131072; do one "2^26 pool +$s KiB" --log2-ops 26 --stagger-kib $s; done
one "2^25 pool natural" --log2-ops 25 --stagger-kib -1
for s in 4 32768 65536; do one "2^25 pool +$s KiB" --log2-ops 25 --stagger-kib $s; done
cat $OUT
