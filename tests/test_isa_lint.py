"""The numerics contract, read off the machine code: the gfx950 code objects inside librt_dp.so are disassembled
and every kernel is checked for floating-point instructions the RTL has no counterpart for.

The RTL has separate adders and multipliers (no fused multiply-add) and its min / max are compare-selects with their
own NaN and signed-zero behaviour (reference src/main/scala/raytracer_datapath/RecFNCompareSelect.scala:41-50), so a
kernel that evaluates datapath ops must contain
  * no v_fma / v_fmac / v_mad / v_mac (any float form, packed or not), and
  * no v_min / v_max / v_med3 / v_min3 / v_max3 on floats.
Integer forms (v_mad_u64_u32, v_min_u32 ...: address arithmetic) are not meant.

Two explicit exceptions, both outside the datapath ops themselves:
  * IEEE division and square root are expanded by the compiler into v_div_scale / v_rcp / fused steps / v_div_fmas /
    v_div_fixup.  Kernels that divide (ray pre-processing, t = t_num / t_denom of the traversal, the angular score)
    may contain fused ops, but exactly as many as their divisions and square roots account for: 5 per division,
    2 per square root.
  * Bounding-box arithmetic of the search structures (k-d leaf bounds, hierarchy builders) is not a datapath op and
    uses float min / max on purpose.
Any kernel not named in an exception is held to the strict rule, new kernels included.
"""
import os
import re
import shutil
import subprocess
import tempfile
from collections import Counter, defaultdict

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "raytracer_amd", "lib", "librt_dp.so")
LLVM = "/opt/rocm/lib/llvm/bin"
OBJDUMP = os.path.join(LLVM, "llvm-objdump")

FLOAT = r"(f16|f32|f64|bf16)"
FUSED = re.compile(r"^v_(pk_)?(fma|fmac|mad|mac|madak|madmk|fmaak|fmamk)(_legacy|_mix|_mixlo|_mixhi)?_" + FLOAT)
MINMAX = re.compile(r"^v_(pk_)?(min3?|max3?|med3|minimum3?|maximum3?)(_legacy)?_" + FLOAT)

# kernels that divide: fused ops allowed, but only the division / square-root expansions
DIVIDES = re.compile(r"k_ray_prepare|k_box_derive|k_tri_derive|k_box_tiledILb[01]ELb1E|k_tri_tiledILb[01]ELb1E|k_trace_init|k_trace_tri|k_trace_dfs|k_knn_euclid16ILb1E|k_knn_tree_ang16|k_bvh_codes")
# bounds of search structures (not datapath ops): float min / max allowed
BOUNDS = re.compile(r"k_knn_tree16|k_knn_tree_ang16|k_knn_pairs|k_bvh_bounds|k_bvh_codes|k_lbvh_boxes|k_kd_")
# the kernels that evaluate datapath ops and must exist in the library (so a renamed kernel cannot dodge the check)
MUST_SEE = ["5k_boxILi1ELb1E", "5k_triILi1ELb1E", "k_box_tiledILb1ELb0E", "k_tri_tiledILb1ELb0E", "5k_vecILb1ELb1E", "5k_vecILb0ELb1E", "k_stream_eval", "k_stream_fold", "k_stream_chainE",
            "k_knn_euclid16ILb0E", "k_trace_box", "k_trace_dfs"]


def disassemble(shared_object, workdir):
    """{kernel symbol: Counter(mnemonic)} over all gfx950 code objects bundled in a shared object or object file."""
    local = os.path.join(workdir, os.path.basename(shared_object))
    shutil.copy(shared_object, local)                      # the tool writes the code objects next to its input
    subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=workdir)
    kernels = defaultdict(Counter)
    objs = [f for f in os.listdir(workdir) if "gfx950" in f]
    assert objs, "no gfx950 code object in %s" % shared_object
    for f in objs:
        text = subprocess.run([OBJDUMP, "-d", os.path.join(workdir, f)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                continue
            m = re.match(r"^\s+([a-z]\w+)", line)
            if m and cur:
                kernels[cur][m.group(1)] += 1
    return kernels


def violations(kernels):
    bad = []
    for name, ops in kernels.items():
        fused = sum(c for op, c in ops.items() if FUSED.match(op))
        minmax = sum(c for op, c in ops.items() if MINMAX.match(op))
        if fused:
            divs = sum(c for op, c in ops.items() if op.startswith("v_div_fmas_"))
            sqrts = sum(c for op, c in ops.items() if op.startswith("v_sqrt_"))
            if not DIVIDES.search(name):
                bad.append("%s: %d fused multiply-add instruction(s)" % (name, fused))
            elif fused != 5 * divs + 2 * sqrts:
                bad.append("%s: %d fused ops, but %d divisions and %d square roots account for %d" % (name, fused, divs, sqrts, 5 * divs + 2 * sqrts))
        if minmax and not BOUNDS.search(name):
            bad.append("%s: %d float min/max/med3 instruction(s)" % (name, minmax))
    return bad


needs_tools = pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")


@needs_tools
def test_library_kernels_contain_no_fused_or_minmax_float_ops():
    assert os.path.exists(LIB), "librt_dp.so has not been built"
    with tempfile.TemporaryDirectory() as tmp:
        kernels = disassemble(LIB, tmp)
    assert len(kernels) > 50
    for want in MUST_SEE:
        assert any(want in k for k in kernels), "kernel %s not found in the library" % want
    bad = violations(kernels)
    assert not bad, "\n".join(bad)
    # the op kernels do carry the arithmetic they are meant to: separate multiplies and adds
    box = next(ops for k, ops in kernels.items() if "5k_boxILi1ELb1E" in k)
    assert sum(c for op, c in box.items() if re.match(r"v_(pk_)?mul_f32", op)) > 0
    assert sum(c for op, c in box.items() if re.match(r"v_(pk_)?(add|sub|subrev)_f32", op)) > 0
    assert sum(c for op, c in box.items() if op.startswith("v_cmp_")) > 0


PROBE = r"""
#include <hip/hip_runtime.h>
__global__ void k_probe_fused(const float *a, const float *b, const float *c, float *o) {
    const int i = threadIdx.x;
    o[i] = a[i] * b[i] + c[i];
}
__global__ void k_probe_minmax(const float *a, const float *b, float *o) {
    const int i = threadIdx.x;
    o[i] = fmaxf(a[i], b[i]);
}
__global__ void k_probe_select(const float *a, const float *b, const float *c, float *o) {
    const int i = threadIdx.x;
    const float p = a[i] * b[i];
    const float s = p + c[i];
    o[i] = (s >= a[i]) ? s : a[i];
}
"""


@needs_tools
@pytest.mark.parametrize("contract,expect_fused", [("off", False), ("fast", True)])
def test_the_lint_sees_what_the_contract_flag_prevents(contract, expect_fused):
    """The checker is sensitive: a multiply-add compiled without -ffp-contract=off becomes a v_fma / v_fmac and is
    reported, fmaxf is reported either way, and the compare-select form the kernels use is clean."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not present")
    with tempfile.TemporaryDirectory() as tmp:
        src, obj = os.path.join(tmp, "probe.hip"), os.path.join(tmp, "probe.o")
        with open(src, "w") as f:
            f.write(PROBE)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=" + contract, "-c", src, "-o", obj], check=True, capture_output=True)
        work = os.path.join(tmp, "dis")
        os.mkdir(work)
        kernels = disassemble(obj, work)
    bad = violations(kernels)
    assert any("k_probe_minmax" in b and "min/max" in b for b in bad)
    assert any("k_probe_fused" in b for b in bad) == expect_fused
    assert any("k_probe_select" in b for b in bad) == expect_fused     # its multiply feeds an add as well
