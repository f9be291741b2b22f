"""The shard planners of the C ABI (rt_dp_shard_plan, rt_dp_shard_plan_records) and the drivers built on them.

CPU part: the planners are pure host functions, so they are checked here without a GPU -- against the Python planner
(raytracer_amd.shard) and against brute-force restatements of their contracts.
GPU part: examples/multi_gpu.c (plain C, one pthread + one context per part) and `rt_dp_replay --gpus N` (C++, one
std::thread + one UnifiedDatapath instance per part) with 1, 2 and 3 contexts; on a one-GPU box all contexts share
GPU 0 (device = part % rt_dp_device_count()), on a node with more GPUs they spread over them.
"""
import glob
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import capi, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MULTI = os.path.join(ROOT, "raytracer_amd", "lib", "multi_gpu")
REPLAY = os.path.join(ROOT, "raytracer_amd", "lib", "rt_dp_replay")
GOLDEN = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.vec")))
RESET = 1 << 16


def random_ctl(rng, n, mean_pair):
    ctl = rng.integers(1, 1 << 16, n, dtype=np.uint32)
    ctl[rng.random(n) < 1.0 / mean_pair] |= RESET
    return ctl


@pytest.mark.parametrize("world", [1, 2, 3, 8, 13])
@pytest.mark.parametrize("mean_pair", [1.0, 4.0, 300.0])
def test_c_planner_equals_python_planner(world, mean_pair):
    rng = np.random.default_rng(world * 100 + int(mean_pair))
    ectl, actl = random_ctl(rng, 5003, mean_pair), random_ctl(rng, 2999, mean_pair)
    want = shard.plan(7001, 6007, ectl, actl, world)
    got = capi.shard_plan(7001, 6007, ectl, actl, world)
    assert got == want
    # the contract itself: contiguous cover, beat cuts on pair boundaries
    for key, n, ctl in (("tri", 7001, None), ("box", 6007, None), ("euclid", 5003, ectl), ("angular", 2999, actl)):
        assert got[0][key][0] == 0 and got[-1][key][1] == n
        for r in range(world - 1):
            cut = got[r][key][1]
            assert cut == got[r + 1][key][0] and got[r][key][0] <= cut
            if ctl is not None:
                assert cut == 0 or cut == n or (ctl[cut - 1] & RESET)


def test_c_planner_edge_cases():
    none = np.zeros(1000, np.uint32)                      # one endless pair: only the first part gets beats
    got = capi.shard_plan(0, 0, none, None, 4)
    assert [p["euclid"] for p in got] == [(0, 1000), (1000, 1000), (1000, 1000), (1000, 1000)]
    assert all(p["tri"] == (0, 0) and p["angular"] == (0, 0) for p in got)
    assert capi.shard_plan(3, 2, None, None, 5) == shard.plan(3, 2, None, None, 5)      # more parts than ops
    with pytest.raises(capi.RtDpError) as e:
        capi.shard_plan(1, 1, None, None, 0)
    assert e.value.code == capi.EINVAL


@pytest.mark.parametrize("world", [1, 2, 3, 7])
@pytest.mark.parametrize("n", [0, 1, 8191, 8192, 8193, 100000])
def test_tiled_cuts_equal_python_tile_cuts(world, n):
    got = capi.shard_plan(n, n, None, None, world, box_tiled=True, tri_tiled=True)
    cuts = shard.tile_cuts(n, world)
    assert [p["box"] for p in got] == [(cuts[r], cuts[r + 1]) for r in range(world)]
    assert [p["tri"] for p in got] == [(cuts[r], cuts[r + 1]) for r in range(world)]
    assert all(p["box"][0] % 8192 == 0 for p in got if p["box"][0] < n)


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_record_planner_cuts_where_both_accumulators_are_idle(world):
    rng = np.random.default_rng(7 + world)
    n = 4000
    rec = np.zeros((n, 84), np.uint32)
    rec[:, 0] = rng.integers(0, 4, n)
    rec[:, 83] = rng.random(n) < 0.3
    cuts = capi.shard_plan_records(rec, world)
    assert cuts[0] == 0 and cuts[-1] == n and cuts == sorted(cuts)
    # brute force: idle[p] = no pair of opcode 2 or 3 under way in front of position p
    open_e = open_a = False
    idle = []
    for i in range(n):
        idle.append(not open_e and not open_a)
        if rec[i, 0] == 2:
            open_e = rec[i, 83] == 0
        elif rec[i, 0] == 3:
            open_a = rec[i, 83] == 0
    idle.append(not open_e and not open_a)
    for r in range(1, world):
        nominal = n * r // world
        want = next((p for p in range(nominal, n + 1) if idle[p]), n)
        assert cuts[r] == max(want, cuts[r - 1])
    rec[5, 0] = 9
    with pytest.raises(capi.RtDpError):
        capi.shard_plan_records(rec, 2)


def test_multi_context_driver_is_built():
    assert os.path.exists(MULTI) and os.path.exists(REPLAY), "build them with python -m raytracer_amd.build"


@pytest.mark.gpu
@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("contexts", [1, 2, 3])
def test_plain_c_driver_one_thread_and_context_per_part(contexts, tiled):
    cmd = [MULTI, "--contexts", str(contexts), "--log2-units", "18"] + (["--tiled"] if tiled else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bit-equal to the whole stream on one context" in r.stdout
    assert r.stdout.count("\npart ") + r.stdout.startswith("part ") == contexts


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 3])
def test_replay_tool_over_several_instances_matches_the_fixtures(gpus):
    for path in GOLDEN:
        r = subprocess.run([REPLAY, path, "--gpus", str(gpus)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "0 mismatching" in r.stdout, path + "\n" + r.stdout[-2000:] + r.stderr[-2000:]
        assert "parts:" in r.stdout
