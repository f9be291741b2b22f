"""bench.py's output contract: one JSON line with the agreed fields, alone and under a two-rank launch (both ranks
on the one GPU, control collectives over gloo: a rehearsal of the N > 1 path, its rates mean nothing)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
          "data", "config", "roofline", "cpu_baseline"]


SMALL = ["--strong-log2-ops", "20", "--knn-log2-points", "14", "--knn-log2-queries", "10"]   # the default line's extras at test size


def check_line(out, n_gpus, steps, warmup, scaling="weak"):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    assert [l for l in out.splitlines() if l.strip()] == lines, "standard output carries more than the result line: " + out[:500]
    d = json.loads(lines[0])
    for f in FIELDS:
        assert f in d, f
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == scaling and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 0 and d["ms_per_step"] > 0
    return d


@pytest.mark.gpu
def test_single_process_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log2-ops", "20", "--only", "box",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = check_line(r.stdout, 1, 3, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "ops/s" and "sample" in c
    # the counter figures in the roofline block belong to the kernel the library really runs for this launch
    r = d["roofline"]
    assert r["kernels"] == {"box": "rtdp::k_box_tiled<true, false>"}
    assert r["traffic"] is not None and "stale" not in r["traffic_source"]


@pytest.mark.gpu
def test_default_line_has_a_counter_pass_for_every_kernel_it_runs():
    """profiles/traffic.json is looked up by the kernel name the library resolves (rt_dp_kernel_name): after a change of
    kernel or of a default shape the committed passes must be regenerated, otherwise the line says "stale"."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = check_line(r.stdout, 1, 2, 1)
    for name in ("box", "tri", "euclid", "angular", "mixed"):
        rf = d["per_opcode"][name]["roofline"]
        assert rf["traffic"] is not None and "stale" not in rf["traffic_source"], (name, rf)


@pytest.mark.gpu
def test_stale_counter_passes_are_called_stale():
    """A kernel shape nobody measured (--vec-passes 2) must not inherit the figures of another kernel."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log2-ops", "20", "--only", "euclid",
                        "--vec-passes", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rf = check_line(r.stdout, 1, 2, 1)["roofline"]
    assert rf["traffic"] is None and rf["traffic_source"].startswith("stale") and "k_vec_chain<true, true, 2, 4>" in rf["traffic_source"]


@pytest.mark.gpu
def test_two_rank_rehearsal_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RT_DP_REHEARSE_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2-ops", "20"] + SMALL
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = check_line(r.stdout, 2, 3, 1)
    assert d["cpu_baseline"] is None
    assert set(d["per_opcode"]) == {"box", "tri", "euclid", "angular", "mixed", "box_plain", "tri_plain", "box_edge", "tri_edge",
                                    "pack_box", "pack_tri", "mixed_2p20", "mixed_2p20_plain", "record_stream", "knn"}
    assert d["per_opcode"]["record_stream"]["unit"] == "records/s" and d["per_opcode"]["record_stream"]["roofline"]["frac"] > 0
    assert d["per_opcode"]["knn"]["exhaustive"]["exchange"] == "gloo (host staging)"     # the rehearsal never touches RCCL
    assert d["per_opcode"]["box"]["layout"] == "tiled" and d["per_opcode"]["box_plain"]["layout"] == "plain"
    assert d["per_opcode"]["euclid"]["layout"] == "plain"
    assert d["per_opcode"]["mixed_2p20"]["scaling"] == "strong" and d["per_opcode"]["knn"]["leaf_major"]["scaling"] == "strong"


@pytest.mark.gpu
def test_plain_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` from a plain shell (no torchrun on the command line, no rank environment): bench.py starts its
    two ranks as a child process and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_DP_REHEARSE_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2-ops", "20"] + SMALL
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = check_line(r.stdout, 2, 3, 1)
    assert d["cpu_baseline"] is None
    assert "torch.distributed.run" not in " ".join(cmd)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["plain", "tiled"])
@pytest.mark.parametrize("gpus", [1, 2, 3])
def test_sharded_mixed_stream_reassembles_to_the_whole_stream(gpus, layout):
    """--only mixed --strong: one stream cut by shard.plan into one pair-aligned shard per rank; with --verify the shards'
    results, put together in rank order, are compared on rank 0 with the whole stream evaluated in one batch -- all through
    the HIP path (the ranks of this rehearsal share the one GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RT_DP_REHEARSE_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--only", "mixed", "--strong",
           "--strong-log2-ops", "19", "--verify", "--layout", layout]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = check_line(r.stdout, gpus, 2, 1, scaling="strong")
    assert d["verified_against_whole_stream"] is True
    c = d["config"]
    assert c["units_whole_job"] >= (1 << 19) - 64 and (gpus == 1) == (c["units_this_gpu"] == c["units_whole_job"])
