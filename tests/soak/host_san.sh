#!/bin/bash
# Host sanitizers over the library's host code.
#   bash tests/soak/host_san.sh build    (anywhere: cross-compiles)  -> raytracer_amd/lib_san/
#   bash tests/soak/host_san.sh run      (on a GPU box)              -> gpurun_out/host_san.txt
# The device code is the normal gfx950 build; only the host side of every source carries
# AddressSanitizer and UndefinedBehaviorSanitizer instrumentation.
set -o pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SAN=$ROOT/raytracer_amd/lib_san
CSRC=$ROOT/raytracer_amd/csrc
CLANG=/opt/rocm/lib/llvm/bin/clang++
case "$1" in
build)
    mkdir -p "$SAN"
    srcs=""
    for s in kernels.hip knn.hip stream.hip stream_fused.hip trace.hip compact.hip pairs.hip build_device.hip build_kd.hip api.cpp build_index.cpp shard.cpp; do srcs="$srcs $CSRC/$s"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
        -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
        -I "$ROOT/include" -I "$CSRC" -shared -o "$SAN/librt_dp.so" $srcs || exit 1
    $CLANG -O1 -g -std=c++17 -fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer -I "$ROOT/include" -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ "$ROOT/tests/soak/host_san.cpp" \
        -o "$SAN/host_san" -L "$SAN" -lrt_dp -L /opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib || exit 1
    ls -l "$SAN"
    ;;
run)
    mkdir -p "$ROOT/gpurun_out"
    # the HIP runtime keeps its own allocations until exit and reserves address space early
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
    export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    timeout -k 10 600 "$SAN/host_san" > "$ROOT/gpurun_out/host_san.txt" 2>&1
    rc=$?
    tail -40 "$ROOT/gpurun_out/host_san.txt"
    exit $rc
    ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
