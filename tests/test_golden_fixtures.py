"""Replay of the committed vector files (tests/golden/*.vec, made by make_fixtures.py).

The expected outputs in those files were produced by oracle A, this repository's
gate-level restatement of the RTL -- not by a Verilator run of the reference,
whose tool-chain is unavailable offline and whose own tests hold no vectors.
CPU: both oracles must still reproduce them (guards the checker against drift).
GPU: the HIP path must reproduce them bit for bit through rt_dp_eval_bundles.
"""
import glob
import os

import numpy as np
import pytest

from raytracer_amd import vecfile

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.vec")))
u32 = np.uint32


def test_fixture_set_is_complete():
    names = {os.path.basename(p) for p in GOLDEN}
    for suite in ("baseline_ray_box_random_100", "baseline_ray_triangle_random_100", "extended_ray_box_random_100",
                  "extended_ray_triangle_random_100", "extended_euclidean_100", "extended_angular_100"):
        assert suite + ".vec" in names
    for p in GOLDEN:
        name, ext, ins, outs = vecfile.read(p)
        assert ins.shape[0] == outs.shape[0] >= 100 and ext in (0, 1)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
@pytest.mark.parametrize("which", ["A", "B"])
def test_oracles_reproduce_fixtures(oracle, path, which):
    _, extended, ins, want = vecfile.read(path)
    got, _ = oracle.eval_bundles(which, ins.view(oracle.IN_BUNDLE).reshape(-1), extended=bool(extended))
    got = got.view(u32).reshape(-1, 21)
    bad = np.nonzero((got != want).any(1))[0]
    assert bad.size == 0, (bad[:5], got[bad[0]], want[bad[0]])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_reproduces_fixtures(torch_cuda, path):
    from raytracer_amd import capi
    _, extended, ins, want = vecfile.read(path)
    ctx = capi.Context(support_euclidean=16 if extended else 0)
    got = ctx.eval_bundles(ins)
    ctx.close()
    bad = np.nonzero((got != want).any(1))[0]
    assert bad.size == 0, (bad[:5], got[bad[0]], want[bad[0]])
