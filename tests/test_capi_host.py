"""Host-side checks that need no GPU: the C-ABI library loads and exports every entry
point include/rt_dp.h declares, the ctypes mirrors match the C layouts, argument
errors are reported without touching a device, and the shard planner cuts only
where the accumulators allow."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from raytracer_amd import capi, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_dp.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_dp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.load()
    names = declared_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(L, name), "librt_dp.so lacks %s" % name
    assert sorted(capi.EXPORTS) == names
    assert b"gfx950" in L.rt_dp_version()


def test_ctypes_layouts_match_the_header(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rt_dp.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(rt_dp_op_batch),sizeof(rt_dp_result_batch),sizeof(rt_dp_config),sizeof(rt_dp_input_bundle),'
                   'sizeof(rt_dp_output_bundle),offsetof(rt_dp_op_batch,box),offsetof(rt_dp_op_batch,euc),offsetof(rt_dp_result_batch,ang));return 0;}\n')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(capi.OpBatch), C.sizeof(capi.ResultBatch), C.sizeof(capi.Config), C.sizeof(capi.InputBundle),
            C.sizeof(capi.OutputBundle), capi.OpBatch.box.offset, capi.OpBatch.euc.offset, capi.ResultBatch.ang.offset]
    assert got == want


def test_argument_errors_without_a_device():
    L = capi.load()
    h = C.c_void_p()
    assert L.rt_dp_create(None, None) == capi.ENULL
    cfg = capi.Config(C.sizeof(capi.Config), -1, 1, 1, 16, 0)
    assert L.rt_dp_create(C.byref(cfg), C.byref(h)) == capi.EINVAL
    assert b"recorded-format Input/Output" in L.rt_dp_last_error()          # UnifiedDatapath.scala:36-39
    cfg = capi.Config(C.sizeof(capi.Config), -1, 0, 0, 16, 0)
    assert L.rt_dp_create(C.byref(cfg), C.byref(h)) == capi.EINVAL
    assert b"standard-format intermediates" in L.rt_dp_last_error()         # :40-43
    cfg = capi.Config(C.sizeof(capi.Config), -1, 0, 1, 8, 0)
    assert L.rt_dp_create(C.byref(cfg), C.byref(h)) == capi.EINVAL
    assert b"16-element vector pairs" in L.rt_dp_last_error()               # :44-47
    cfg = capi.Config(4, -1, 0, 1, 16, 0)
    assert L.rt_dp_create(C.byref(cfg), C.byref(h)) == capi.EINVAL
    assert L.rt_dp_eval_ctx(None, None, None) == capi.ENULL
    assert L.rt_dp_launch(None, None, None, None) == capi.ENULL
    assert L.rt_dp_eval_bundles(None, None, None, 1) == capi.ENULL
    assert L.rt_dp_get_accum(None, None) == capi.ENULL
    # the search / traversal / stream entry points refuse a NULL context before touching anything else
    assert L.rt_dp_eval_bundles_device(None, None, None, 1) == capi.ENULL
    assert L.rt_dp_ray_prepare(None, None, 1, None) == capi.ENULL
    assert L.rt_dp_knn_euclid16(None, None, 1, None, 1, 0, 1, None, None, None) == capi.ENULL
    assert L.rt_dp_knn_angular16(None, None, 1, None, 1, 0, 1, None, None, None) == capi.ENULL
    assert L.rt_dp_knn_tree_euclid16(None, None, None, None, 1, 1, None, None, None, None) == capi.ENULL
    assert L.rt_dp_knn_tree_grouped_euclid16(None, None, None, None, 1, 16, None, None, None) == capi.ENULL
    assert L.rt_dp_trace_closest(None, None, None, None, 1, 4000.0, 0, None, None, None) == capi.ENULL
    assert L.rt_dp_box_compact(None, None, 1, None, None, None, None, None, 0, None) == capi.ENULL
    assert L.rt_dp_pairs_to_beats(None, None, None, None, None, 1, 1, 16, None, None, None, None) == capi.ENULL
    assert L.rt_dp_beats_to_pairs(None, None, None, 1, None, None) == capi.ENULL
    assert L.rt_dp_knn_tree_build_device(None, None, 1, 4, 0, None, None, None, None, None, None) == capi.ENULL
    assert L.rt_dp_knn_home_leaf(None, None, 1, 1, None, None, None, None) == capi.ENULL
    assert L.rt_dp_bvh4_build_device(None, None, 1, 4, None, None, None, None) == capi.ENULL
    assert L.rt_dp_set_option(None, b"variant", 0) == capi.ENULL
    assert L.rt_dp_set_tuning(None, 0, 0, 0) == capi.ENULL
    assert L.rt_dp_synchronize(None) == capi.ENULL
    L.rt_dp_destroy(None)  # a no-op


def test_no_device_is_reported_not_faked():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.RtDpError) as e:
        capi.Context()
    assert e.value.code == capi.ENODEV  # there is no CPU fallback


def test_even_cuts_cover_everything():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            c = shard.even_cuts(n, w)
            assert c[0] == 0 and c[-1] == n and all(a <= b for a, b in zip(c[:-1], c[1:]))
            assert max(b - a for a, b in zip(c[:-1], c[1:])) - min(b - a for a, b in zip(c[:-1], c[1:])) <= 1


def test_vec_cuts_fall_on_pair_boundaries():
    from oracle import gen
    _, _, ctl, _ = gen.vec_beats(5000, 16, seed=2)
    for w in (1, 2, 4, 8):
        cuts = shard.vec_cuts(ctl, w)
        assert cuts[0] == 0 and cuts[-1] == ctl.shape[0]
        for c in cuts[1:-1]:
            assert (ctl[c - 1] >> 16) & 1
        sizes = np.diff(cuts)
        assert sizes.max() - sizes.min() <= 2 * 8  # a pair has at most 8 beats here
    endless = (ctl & 0xFFFF).astype(np.uint32)  # no boundary anywhere: one rank gets everything
    assert shard.vec_cuts(endless, 4) == [0, ctl.shape[0], ctl.shape[0], ctl.shape[0], ctl.shape[0]]
    assert shard.snap_to_pair_boundary(ctl, 0) == 0
    assert shard.snap_to_pair_boundary(ctl, 10 ** 9) == ctl.shape[0]
