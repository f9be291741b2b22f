"""The vector-capture hook for the reference's testbench (docs/datapath_test_dump.patch) and the reader of its
output.  The patch is print-only: a monitor that writes one "I" line per accepted input record and one "O" line
per delivered output record.  Nothing of the reference is built or run here (no JVM / mill / Verilator offline):
  * the patch applies cleanly to the reference's Datapath_test.scala (git apply --check on a scratch copy);
  * a stream in exactly the text form the monitor prints -- the hand-derived known answers, whose expected outputs
    were typed in by hand (tests/golden/kat/make_kat.py) -- goes through vecfile.read, both oracles, and on the GPU
    through rt_dp_eval_bundles and the rt_dp_replay tool.
With a JDK and Verilator, `RTDP_DUMP_DIR=out mill raytracer.test` after `git apply` yields files of this form with
true RTL outputs, and `rt_dp_replay out/<suite>.vec` checks the GPU path against them."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from raytracer_amd import vecfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "docs", "datapath_test_dump.patch")
SAMPLE = os.path.join(ROOT, "tests", "golden", "kat", "rtl_dump_sample.vec")
REF_TEST = "/root/reference/src/test/scala/raytracer_datapath/Datapath_test.scala"
TOOL = os.path.join(ROOT, "raytracer_amd", "lib", "rt_dp_replay")
u32 = np.uint32


@pytest.mark.skipif(not os.path.exists(REF_TEST), reason="the reference tree is not present on this machine")
def test_patch_applies_to_the_reference_testbench(tmp_path):
    dst = tmp_path / "src" / "test" / "scala" / "raytracer_datapath"
    dst.mkdir(parents=True)
    shutil.copy(REF_TEST, dst / "Datapath_test.scala")
    r = subprocess.run(["git", "apply", "--check", "-p1", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["git", "apply", "-p1", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = (dst / "Datapath_test.scala").read_text()
    assert text.count("rtdpDump(dut, description") == 3          # Euclidean, angular and the unified box / triangle routine
    assert "RTDP_DUMP_DIR" in text


def test_patch_is_print_only():
    added = [l[1:] for l in open(PATCH) if l.startswith("+") and not l.startswith("+++")]
    removed = [l for l in open(PATCH) if l.startswith("-") and not l.startswith("---")]
    assert not removed, "the patch must not change the testbench's behaviour"
    assert 30 <= len(added) <= 60
    body = "".join(added)
    assert ".poke(" not in body and "enqueue" not in body and "expect(" not in body   # it only peeks and steps its own monitor thread


def test_sample_in_the_monitor_format_reads_and_matches_both_oracles(oracle):
    name, extended, ins, want = vecfile.read(SAMPLE)
    assert name == "hand_derived_kat" and extended == 1 and ins.shape == (8, 84) and want.shape == (8, 21)
    for which in ("A", "B"):
        got, _ = oracle.eval_bundles(which, ins.view(oracle.IN_BUNDLE).reshape(-1), extended=True)
        assert np.array_equal(got.view(u32).reshape(-1, 21), want), which


@pytest.mark.gpu
def test_sample_through_the_gpu_path_and_the_replay_tool(torch_cuda):
    from raytracer_amd import capi
    _, _, ins, want = vecfile.read(SAMPLE)
    ctx = capi.Context()
    got = ctx.eval_bundles(ins)
    ctx.close()
    assert np.array_equal(got, want)
    r = subprocess.run([TOOL, SAMPLE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 mismatching" in r.stdout, r.stdout + r.stderr
