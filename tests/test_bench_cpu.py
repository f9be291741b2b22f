"""bench.py's launch logic that needs no GPU: `--gpus N` from a plain shell starts N ranks as a child
`python -m torch.distributed.run` (never an exec, and before anything touches a GPU); inside a rank environment, or
with one GPU, nothing is started."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    return importlib.import_module("bench")


def test_plain_gpus_flag_starts_a_child_launcher(bench, monkeypatch):
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    for argv in (["--gpus", "4", "--steps", "3"], ["--steps", "3", "--gpus=4"]):
        calls.clear()
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        with pytest.raises(SystemExit) as e:
            bench.spawn_ranks()
        assert e.value.code == 7                                     # the child's exit code is relayed
        (cmd, env), = calls
        assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
        assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
        assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[-len(argv):] == argv and os.path.basename(cmd[-len(argv) - 1]) == "bench.py"
        assert "WORLD_SIZE" not in env                              # the launcher sets the rank environment itself


def test_nothing_is_started_inside_a_rank_or_for_one_gpu(bench, monkeypatch):
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("a child was started"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.spawn_ranks() is None
    monkeypatch.delenv("WORLD_SIZE")
    for argv in ([], ["--gpus", "1"], ["--steps", "5"]):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        assert bench.spawn_ranks() is None


def test_parent_does_not_load_torch_or_hip_before_it_spawns():
    """Run the real file with --gpus 2 and a stand-in `torch.distributed.run` on the path: the parent must reach the spawn
    without importing torch (so it cannot have touched a GPU)."""
    code = ("import sys, runpy, subprocess\n"
            "subprocess.call = lambda cmd, env=None: (print('SPAWN', 'torch' in sys.modules, 'raytracer_amd.capi' in sys.modules), 0)[1]\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(ROOT, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "SPAWN False False" in r.stdout, r.stdout + r.stderr
