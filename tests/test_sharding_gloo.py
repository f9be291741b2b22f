"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo).

The data path has no collective: every rank evaluates its own shard.  What must
hold is that shards cut by raytracer_amd.shard are independent, i.e. that the
per-rank results, concatenated in rank order, equal the results of the whole
stream evaluated by one datapath.  The evaluator inside this test is the CPU
oracle (no GPU here); the cut logic, the control-plane helpers and the process
launch are the product's.
"""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from raytracer_amd import dist_util, shard
from oracle import gen
from oracle import dp_oracle as O

world, rank, local = dist_util.init("gloo")
assert world == 2
box = gen.box_ops(6001, seed=1, edge_frac=0.2)
tri, k = gen.tri_ops(5003, seed=2, edge_frac=0.2)
ea, eb, ectl, _ = gen.vec_beats(700, 16, seed=3, edge_frac=0.1)
qa, qb, qctl, _ = gen.vec_beats(500, 8, seed=4, edge_frac=0.1)
plan = shard.plan(k.shape[0], box[0].shape[0], ectl, qctl, world)[rank]
# the planner of the C ABI (what a C host driver with one thread per GPU calls) makes the same cuts
from raytracer_amd import capi
assert capi.shard_plan(k.shape[0], box[0].shape[0], ectl, qctl, world)[rank] == plan

s, e = plan["box"]
tmin, flags = O.box_soa([f[s:e] for f in box])
s, e = plan["tri"]
tn, td, hit = O.tri_soa([f[s:e] for f in tri], k[s:e])
s, e = plan["euclid"]
acc, acc_state = O.euclid_beats(ea[s:e], eb[s:e], ectl[s:e])   # every shard starts from +0
s, e = plan["angular"]
dot, norm, _, _ = O.angular_beats(qa[s:e], qb[s:e], qctl[s:e])

dist_util.barrier()
t = dist_util.max_over_ranks(1.0 + rank)
assert t == 2.0
assert dist_util.sum_over_ranks(flags.shape[0]) == 6001
parts = dist_util.gather_objects({"plan": plan, "flags": flags, "tmin0": tmin[0], "hit": hit, "tn": tn, "acc": acc, "dot": dot, "norm": norm})
if rank == 0:
    wt, wf = O.box_soa(box)
    wn, wd, wh = O.tri_soa(tri, k)
    wacc, _ = O.euclid_beats(ea, eb, ectl)
    wdot, wnorm, _, _ = O.angular_beats(qa, qb, qctl)
    cat = lambda key: np.concatenate([p[key] for p in parts])
    same = lambda a, b: a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()
    assert same(cat("flags"), wf) and same(cat("tmin0"), wt[0])
    assert same(cat("hit"), wh) and same(cat("tn"), wn)
    assert same(cat("acc"), wacc) and same(cat("dot"), wdot) and same(cat("norm"), wnorm)
    assert parts[0]["plan"]["euclid"][1] == parts[1]["plan"]["euclid"][0]
    print("SHARD-OK")
dist_util.shutdown()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_shards_reassemble_to_the_whole_stream(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARD-OK" in out.stdout
