#!/bin/bash
# The default line's five workloads (no extras) in the driver's form (--steps 20 --warmup 5) for several kernel choices,
# interleaved: what a kernel does inside the bench process (other batches allocated) can differ from `--only`.
OUT=gpurun_out/bench_ctx.txt
: > $OUT
show() { python -c "
import sys,json
r=json.loads(open('$1').read().strip().splitlines()[-1])
print('%-28s' % '$2', ' '.join('%s %.4f' % (k, v['roofline']['frac']) for k, v in r['per_opcode'].items() if 'roofline' in v and k in ('box','tri','euclid','angular','mixed')))" >> $OUT; }
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/b.json 2>/dev/null && show gpurun_out/b.json "default"
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --variant 23 > gpurun_out/b.json 2>/dev/null && show gpurun_out/b.json "variant 23"
[ -n "$1" ] && RT_DP_LIB=$1 timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/b.json 2>/dev/null && show gpurun_out/b.json "other library"
done
cat $OUT
