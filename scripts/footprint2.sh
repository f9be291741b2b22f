set -o pipefail
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/footprint_r02
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
PASSES=(
 "utcl1lat:TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_LFIFO_FULL TCP_LFIFO_STALL_CYCLES_sum"
 "tcpstall:TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "sqwait:SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM"
)
for lg in 24 25 26; do
  for spec in "${PASSES[@]}"; do
    name=${spec%%:*}; ctrs=${spec#*:}
    echo "== box 2^$lg $name" | tee -a "$OUT/log2.txt"
    timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/box_${lg}_$name" -- python3 $ROOT/bench.py --only box --log2-ops $lg --steps 3 --warmup 1 --no-cpu-baseline >> "$OUT/log2.txt" 2>&1 || { echo "FAILED: $name" | tee -a "$OUT/log2.txt"; exit 1; }
  done
done
for lg in 25; do
  for spec in "grbm:GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" "utcl1:TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" "ealevel:TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"; do
    name=${spec%%:*}; ctrs=${spec#*:}
    timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d "$OUT/box_${lg}_$name" -- python3 $ROOT/bench.py --only box --log2-ops $lg --steps 3 --warmup 1 --no-cpu-baseline >> "$OUT/log2.txt" 2>&1 || exit 1
  done
done
echo done2 | tee -a "$OUT/log2.txt"
