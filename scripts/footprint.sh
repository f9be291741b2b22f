#!/bin/bash
# Counter passes behind profiles/*_footprint.md: the box kernel at two batch sizes (2^24 ops = 2.6 GB and 2^26 ops =
# 10.5 GB of operands and results), translation-cache, L2 <-> fabric and stall counters, every pass in a
# rocprofv3 run of its own.  Usage (GPU box, repo root): bash scripts/footprint.sh <tag> [workload]
set -o pipefail
TAG=${1:-r02}
WL=${2:-box}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/footprint_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "utcl1:TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
 "utcl1stall:TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "utcl1more:TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum"
 "grbm:GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"
 "eastall:TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "ealevel:TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"
 "tcc:TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_CYCLE_sum"
)
for lg in 24 26; do
  for spec in "${PASSES[@]}"; do
    name=${spec%%:*}; ctrs=${spec#*:}
    echo "== $WL 2^$lg $name" | tee -a "$OUT/log.txt"
    timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/${WL}_${lg}_$name" -- python3 $ROOT/bench.py --only $WL --log2-ops $lg --steps 3 --warmup 1 --no-cpu-baseline >> "$OUT/log.txt" 2>&1 || { echo "FAILED: $name" | tee -a "$OUT/log.txt"; exit 1; }
  done
done
echo "footprint passes done" | tee -a "$OUT/log.txt"
