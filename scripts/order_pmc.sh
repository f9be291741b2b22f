#!/bin/bash
# translation counters of the beat kernels, alone and alternating: bash scripts/order_pmc.sh
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/order_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $OUT/p1 -- python3 $ROOT/scripts/probe/order_probe.py 26 0 EEEE,AAAA,EAEA > $OUT/p1.log 2>&1 || exit 1
python3 - $OUT/p1 <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'k_vec' not in r['Kernel_Name']: continue
    d = rows.setdefault(int(r['Dispatch_Id']), {'k': 'E' if 'chain' in r['Kernel_Name'] else 'A'})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
seq = [rows[k] for k in sorted(rows)]
# phases: EEEE = 4 warm + 24, AAAA = 4 + 24, EAEA = 4 + 24
names = ['GRBM_GUI_ACTIVE', 'GRBM_UTCL2_BUSY', 'TCP_UTCL1_TRANSLATION_MISS_sum', 'TCP_CLIENT_UTCL1_INFLIGHT_sum', 'TCP_UTCL1_REQUEST_sum']
for label, lo, hi in (('alone E', 4, 28), ('alone A', 32, 56), ('alternating', 60, 84)):
    part = seq[lo:hi]
    for k in 'EA':
        sel = [d for d in part if d['k'] == k]
        if sel: print(label, k, len(sel), ' '.join('%s %.4g' % (n, sum(d.get(n, 0) for d in sel) / len(sel)) for n in names))
PY
