"""The rt_dp_replay host driver (C++ mirror of the testbench loop over the C ABI)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "raytracer_amd", "lib", "rt_dp_replay")
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.vec")))


def test_replay_tool_is_built_and_refuses_to_run_without_a_gpu():
    import torch
    assert os.path.exists(TOOL), "build it with python -m raytracer_amd.build"
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([TOOL, GOLDEN[0]], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr  # no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_replay_matches_fixture(path, tmp_path):
    dump = tmp_path / "out.vec"
    r = subprocess.run([TOOL, path, "--dump", str(dump)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatching" in r.stdout
    # the dump round-trips: replaying it compares the GPU path with itself
    r = subprocess.run([TOOL, str(dump), "--quiet"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0


@pytest.mark.gpu
def test_replay_reports_a_corrupted_expectation(tmp_path):
    lines = open(GOLDEN[0]).read().splitlines()
    idx = next(i for i, l in enumerate(lines) if l.startswith("O "))
    words = lines[idx].split()
    words[2] = "%x" % (int(words[2], 16) ^ 1)
    lines[idx] = " ".join(words)
    bad = tmp_path / "bad.vec"
    bad.write_text("\n".join(lines) + "\n")
    r = subprocess.run([TOOL, str(bad)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "1 mismatching" in r.stdout


KAT = os.path.join(ROOT, "raytracer_amd", "lib", "kat_from_c")


def test_c_example_is_built_and_reports_a_missing_gpu():
    import torch
    assert os.path.exists(KAT), "build it with python -m raytracer_amd.build"
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([KAT], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stdout  # rt_dp_eval fails loudly, nothing is computed on the CPU


@pytest.mark.gpu
def test_known_answers_from_plain_c():
    """examples/kat_from_c.c: the four hand-derived known-answer cases through rt_dp_eval, host arrays, C only."""
    r = subprocess.run([KAT], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "KAT OK" in r.stdout, r.stdout + r.stderr
