#!/bin/bash
# rocprofv3 passes behind the numbers in profiles/: one kernel-trace run of the
# benchmark, then FETCH_SIZE / WRITE_SIZE counter passes (each in a run of its
# own) for the benchmark and for the membench calibration kernels.
# Usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag>
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

run() { # name, limit, command...
    local name=$1 limit=$2; shift 2
    echo "== $name" | tee -a "$OUT/log.txt"
    timeout -s KILL "$limit" "$@" >> "$OUT/log.txt" 2>&1 || { echo "FAILED: $name" | tee -a "$OUT/log.txt"; exit 1; }
}

BENCH="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
run kernel_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH
for wl in box tri; do   # the same two opcodes over one array per operand
    run "fetch_${wl}_plain" 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_${wl}_plain" -- $BENCH --only $wl --layout plain --steps 3 --warmup 1
    run "write_${wl}_plain" 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_${wl}_plain" -- $BENCH --only $wl --layout plain --steps 3 --warmup 1
    run "valu_${wl}_plain" 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_valu_${wl}_plain" -- $BENCH --only $wl --layout plain --steps 3 --warmup 1
    run "grbm_${wl}_plain" 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm_${wl}_plain" -- $BENCH --only $wl --layout plain --steps 3 --warmup 1
done
for wl in box tri euclid angular; do
    run "fetch_$wl" 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$wl" -- $BENCH --only $wl --steps 3 --warmup 1
    run "write_$wl" 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$wl" -- $BENCH --only $wl --steps 3 --warmup 1
done
for wl in box tri euclid angular; do
    run "valu_$wl" 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_valu_$wl" -- $BENCH --only $wl --steps 3 --warmup 1
    run "grbm_$wl" 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm_$wl" -- $BENCH --only $wl --steps 3 --warmup 1
done
# short angular segments (the quarters of a mixed batch) run the chained kernel: a pass of its own, keyed by kernel name in traffic.json
for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "valu:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" "grbm:GRBM_GUI_ACTIVE"; do
    run "${c%%:*}_angular_short" 120 rocprofv3 --pmc ${c#*:} --output-format csv -d "$OUT/pmc_${c%%:*}_angular_short" -- $BENCH --only angular --log2-ops 22 --steps 3 --warmup 1
done
# the record-stream front end (evaluate in place, then fold): bytes per record of its three kernels
for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "valu:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
    run "${c%%:*}_stream" 120 rocprofv3 --pmc ${c#*:} --output-format csv -d "$OUT/pmc_${c%%:*}_stream" -- python3 $ROOT/bench.py --only stream --stream-mode 2 --steps 3 --warmup 1
done
# the drivers above the op kernels: record stream (both modes), k-NN (exhaustive and tree), traversal (both walks)
for spec in "stream1:--only stream --stream-mode 1" "stream2:--only stream --stream-mode 2" "knn_exhaustive:--only knn --knn-tree 0" \
            "knn_tree_walk:--only knn --knn-tree 1" "knn_tree_leafmajor:--only knn --knn-tree 2" "trace_dfs:--only trace" "trace_dfs_sorted:--only trace --trace-sorted 1" "trace_wavefront:--only trace --trace-wavefront 1" "trace_device_built:--only trace --trace-build device" \
            "knn_angular:--only knn --knn-tree 0 --knn-metric angular" "derive_box:--only box --derive 1 --no-cpu-baseline" "derive_tri:--only tri --derive 1 --no-cpu-baseline"; do
    name=${spec%%:*}; args=${spec#*:}
    run "kt_$name" 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$name" -- python3 $ROOT/bench.py --steps 3 --warmup 1 $args
done
run fetch_membench 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_membench" -- "$ROOT/raytracer_amd/lib/membench" 1024
run write_membench 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_membench" -- "$ROOT/raytracer_amd/lib/membench" 1024
echo "profile passes done" | tee -a "$OUT/log.txt"
