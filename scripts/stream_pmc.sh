#!/bin/bash
# counters of the record-stream kernels (mode 2, 2^24 records): bash scripts/stream_pmc.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stream_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --only stream --stream-mode 2 --log2-ops 24 --steps 3 --warmup 1"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for p in ('p1', 'p2', 'p3'):
    f = glob.glob(sys.argv[1] + '/' + p + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_stream' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
    for k in acc:
        print(p, k, 'dispatches', len(cnt[k]), ' '.join('%s %.4g' % (n, v / len(cnt[k])) for n, v in sorted(acc[k].items())))
PY
