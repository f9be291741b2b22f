"""k-NN on the Euclidean distance op: GPU kernel + merge against a brute-force oracle,
and the all-gather merge rehearsed with two gloo ranks on CPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32 = np.float32


def make_cloud(n, m, seed, specials=True):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1, 1, (n, 16)).astype(f32)
    qry = rng.uniform(-1, 1, (m, 16)).astype(f32)
    if specials and n >= 64:
        pts[10] = pts[3]            # duplicate points: equal distances, lower index first
        pts[11] = pts[3]
        pts[20, 5] = np.nan         # NaN distance: never listed
        pts[21, 0] = np.inf         # +Inf distance: listed last, before unused slots
        pts[22] = qry[0]            # exact hit: distance +0
        pts[23] *= f32(1e-30)       # squares underflow into the subnormal range
        qry[1] = pts[40]
    return pts, qry


def test_merge_is_partition_independent():
    import torch
    from raytracer_amd import knn
    rng = np.random.default_rng(0)
    m = 50
    d = torch.from_numpy(np.sort(rng.random((4, m, 16)).astype(f32), axis=2))
    i = torch.from_numpy(rng.integers(0, 1 << 31, (4, m, 16)).astype(np.int32))
    keys = knn.pack_keys(d, i)
    whole = knn.merge_keys(keys)
    halves = knn.merge_keys(torch.stack([knn.merge_keys(keys[:2]), knn.merge_keys(keys[2:])]))
    assert torch.equal(whole, halves)
    dd, ii = knn.unpack_keys(whole)
    assert bool((dd[:, 1:] >= dd[:, :-1]).all())
    # unsigned index order and the empty-slot key
    assert knn.pack_keys(torch.tensor([1.0]), torch.tensor([-1], dtype=torch.int32)).item() > knn.pack_keys(torch.tensor([1.0]), torch.tensor([5], dtype=torch.int32)).item()
    assert knn.pack_keys(torch.tensor([float("inf")]), torch.tensor([-1], dtype=torch.int32)).item() == knn.EMPTY_KEY


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,splits", [(1, 3, 1), (15, 5, 1), (2000, 300, 1), (2000, 300, 7), (5000, 257, None)])
def test_knn_matches_bruteforce(ctx, oracle, torch_cuda, n, m, splits):
    torch = torch_cuda
    from raytracer_amd import knn
    pts, qry = make_cloud(n, m, seed=n + m)
    want_d, want_i = oracle.knn_bruteforce(pts, qry)
    keys = knn.local_topk(ctx, torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda(), 0, splits=splits)
    d, i = knn.unpack_keys(keys)
    assert (d.cpu().numpy().view(np.uint32) == want_d.view(np.uint32)).all()
    assert (i.cpu().numpy() == want_i).all()


@pytest.mark.gpu
def test_knn_sharded_equals_unsharded(ctx, oracle, torch_cuda):
    torch = torch_cuda
    from raytracer_amd import knn
    pts, qry = make_cloud(6000, 200, seed=5)
    P, Q = torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda()
    whole = knn.local_topk(ctx, P, Q, 0)
    cuts = [0, 1000, 1001, 4096, 6000]
    parts = [knn.local_topk(ctx, P[s:e].contiguous(), Q, s) for s, e in zip(cuts[:-1], cuts[1:])]
    assert torch.equal(knn.merge_keys(torch.stack(parts)), whole)
    want_d, want_i = oracle.knn_bruteforce(pts, qry)
    d, i = knn.unpack_keys(whole)
    assert (i.cpu().numpy() == want_i).all() and (d.cpu().numpy().view(np.uint32) == want_d.view(np.uint32)).all()


def test_signed_keys_order_like_floats():
    import torch
    from raytracer_amd import knn
    v = torch.tensor([-float("inf"), -3.5, -1e-40, 0.0, 1e-40, 2.0, float("inf")])
    i = torch.tensor([7, -1, 3, 9, 0, 5, -1], dtype=torch.int32)
    keys = knn.pack_signed_keys(v, i)
    assert bool((keys[1:] > keys[:-1]).all())
    vv, ii = knn.unpack_signed_keys(keys)
    assert torch.equal(vv.view(torch.int32), v.view(torch.int32)) and ii.tolist() == [7, 0xFFFFFFFF, 3, 9, 0, 5, 0xFFFFFFFF]
    same = knn.pack_signed_keys(torch.tensor([1.5, 1.5]), torch.tensor([4, 2], dtype=torch.int32))
    assert same[1] < same[0]  # equal values: the lower index first


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,splits", [(1, 3, 1), (15, 5, 1), (2000, 200, 1), (2000, 200, 5), (5000, 130, None)])
def test_angular_knn_matches_bruteforce(ctx, oracle, torch_cuda, n, m, splits):
    torch = torch_cuda
    from raytracer_amd import knn
    pts, qry = make_cloud(n, m, seed=3 * n + m)
    if n >= 64:
        pts[30] = 0.0                       # zero norm: 0 / 0, never listed
        pts[31] = pts[32] * f32(2.0)        # same direction as point 32: equal scores up to rounding
        pts[33] = pts[34]                   # exactly equal scores: the lower index first
        pts[35] *= f32(1e-25)               # norm underflows towards the subnormal range
        qry[2] = 0.0                        # every dot product is +-0: all scores tie at 0
    want_v, want_i = oracle.knn_bruteforce_angular(pts, qry)
    keys = knn.local_topk_angular(ctx, torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda(), 0, splits=splits)
    v, i = knn.unpack_signed_keys(keys)
    assert (i.cpu().numpy() == want_i).all()
    assert (v.cpu().numpy().view(np.uint32) == want_v.view(np.uint32)).all()


def clustered_cloud(n, m, seed):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1, 1, (12, 16)).astype(f32)
    pts = (centres[rng.integers(0, 12, n)] + rng.normal(0, 0.05, (n, 16))).astype(f32)
    qry = (centres[rng.integers(0, 12, m)] + rng.normal(0, 0.08, (m, 16))).astype(f32)
    pts[n // 2:n // 2 + 40] = pts[7]      # 41 copies of one point: more ties than list slots
    qry[2] = pts[7]
    qry[3, 4] = np.nan                    # every distance NaN: an empty list
    qry[4, 0] = np.inf
    return pts, qry


@pytest.mark.gpu
@pytest.mark.parametrize("cloud,n,m,leaf,splits", [
    ("uniform", 10, 5, 4, 1), ("uniform", 10, 5, 128, 1), ("uniform", 5000, 300, 64, 1), ("uniform", 5000, 300, 64, 4),
    ("uniform", 40000, 1000, 128, None), ("clustered", 30000, 700, 32, 8), ("clustered", 30000, 700, 128, None),
    ("lowdim", 50000, 600, 64, None)])
def test_tree_search_equals_exhaustive_search(ctx, oracle, torch_cuda, cloud, n, m, leaf, splits):
    """The pruned walk must return exactly what the exhaustive kernel returns (which the test above pins to the oracle)."""
    torch = torch_cuda
    from raytracer_amd import knn, knn_tree
    if cloud == "uniform":
        pts, qry = make_cloud(n, m, seed=n + m + leaf)
    elif cloud == "clustered":
        pts, qry = clustered_cloud(n, m, seed=leaf)
    else:  # points on a 3-dimensional subspace: the tree prunes almost everything
        rng = np.random.default_rng(9)
        basis = rng.normal(0, 1, (3, 16))
        pts = (rng.uniform(-1, 1, (n, 3)) @ basis).astype(f32)
        qry = (rng.uniform(-1, 1, (m, 3)) @ basis).astype(f32)
    P, Q = torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda()
    want = knn.local_topk(ctx, P, Q, 1000)
    tree = knn_tree.KnnTree(P, leaf_size=leaf, index_base=1000)
    got = knn_tree.local_topk(ctx, tree, Q, splits=splits)
    assert torch.equal(got, want)
    for walk in (1, 16):  # the leaf-major search: same answers whatever the quality of its first bound
        stats = {}
        got = knn_tree.local_topk_grouped(ctx, tree, Q, walk_leaves=walk, stats=stats)
        bad = (got != want).any(1).nonzero().flatten().tolist()
        assert not bad, (stats, bad[:5])
    if n <= 5000:
        wd, wi = oracle.knn_bruteforce(pts, qry, 16, 1000)
        d, i = knn.unpack_keys(got)
        assert (i.cpu().numpy() == wi).all() and (d.cpu().numpy().view(np.uint32) == wd.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cloud,n,m,leaf,splits", [
    ("uniform", 10, 5, 4, 1), ("uniform", 5000, 300, 64, 1), ("uniform", 5000, 300, 64, 4), ("uniform", 40000, 1000, 128, None),
    ("clustered", 30000, 700, 32, 8), ("clustered", 30000, 700, 128, None), ("lowdim", 50000, 600, 64, None), ("scaled", 30000, 500, 64, None)])
@pytest.mark.parametrize("build", ["torch", "device"])
def test_angular_tree_search_equals_exhaustive_angular_search(ctx, oracle, torch_cuda, cloud, n, m, leaf, splits, build):
    """rt_dp_knn_tree_angular16 skips leaves on a bound of the score; its lists must be those of the exhaustive angular kernel
    (pinned to the oracle above), key for key: uniform data (zero-norm points, NaN and infinite coordinates, tiny norms, a
    zero query, ties), clustered directions (where it really prunes), a low-dimensional subspace, and clusters scaled over
    sixty orders of magnitude (norms and dot products near both ends of the float range)."""
    torch = torch_cuda
    from raytracer_amd import knn, knn_tree
    if cloud == "uniform":
        pts, qry = make_cloud(n, m, seed=n + m + leaf)
        if n >= 64:
            pts[30] = 0.0
            pts[31] = pts[32] * f32(2.0)
            pts[33] = pts[34]
            pts[35] *= f32(1e-25)
            qry[2] = 0.0
    elif cloud == "clustered":
        pts, qry = clustered_cloud(n, m, seed=leaf)
    elif cloud == "lowdim":
        rng = np.random.default_rng(9)
        basis = rng.normal(0, 1, (3, 16))
        pts = (rng.uniform(-1, 1, (n, 3)) @ basis).astype(f32)
        qry = (rng.uniform(-1, 1, (m, 3)) @ basis).astype(f32)
    else:
        pts, qry = clustered_cloud(n, m, seed=77)
        rng = np.random.default_rng(78)
        pts = (pts * (f32(10.0) ** rng.integers(-30, 18, (n, 1)).astype(f32))).astype(f32)      # squares from the subnormal range up to 1e36
        qry = (qry * (f32(10.0) ** rng.integers(-20, 18, (m, 1)).astype(f32))).astype(f32)
    P, Q = torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda()
    want = knn.local_topk_angular(ctx, P, Q, 1000)
    tree = knn_tree.KnnTree.build_device(ctx, P, leaf_size=leaf, index_base=1000) if build == "device" else knn_tree.KnnTree(P, leaf_size=leaf, index_base=1000)
    stats = {}
    got = knn_tree.local_topk_angular(ctx, tree, Q, splits=splits, stats=stats)
    bad = (got != want).any(1).nonzero().flatten().tolist()
    assert not bad, (stats, bad[:5])
    assert 0.0 < stats["leaves_scanned_fraction"] <= 1.0
    if cloud in ("clustered", "lowdim") and n >= 30000:
        assert stats["leaves_scanned_fraction"] < 0.9, stats       # the bound does prune where directions cluster
    if n <= 5000:
        wv, wi = oracle.knn_bruteforce_angular(pts, qry, 16, 1000)
        v, i = knn.unpack_signed_keys(got)
        assert (i.cpu().numpy() == wi).all() and (v.cpu().numpy().view(np.uint32) == wv.view(np.uint32)).all()


@pytest.mark.gpu
def test_tree_argument_errors(ctx, torch_cuda):
    torch = torch_cuda
    from raytracer_amd import capi, knn_tree
    P = torch.rand(1000, 16, device="cuda")
    tree = knn_tree.KnnTree(P, leaf_size=64)
    Q = torch.rand(10, 16, device="cuda")
    h = torch.zeros(10, dtype=torch.int32, device="cuda")
    b = torch.empty(10, dtype=torch.int32, device="cuda")
    d = torch.empty(3, 10, 16, device="cuda")
    i = torch.empty(3, 10, 16, dtype=torch.int32, device="cuda")
    with pytest.raises(capi.RtDpError) as e:
        ctx.knn_tree_euclid16(tree, Q, h, b, d, i, splits=3)
    assert e.value.code == capi.EINVAL
    tree.leaf_size = 6
    with pytest.raises(capi.RtDpError):
        ctx.knn_tree_euclid16(tree, Q, h, b, d[:1].contiguous(), i[:1].contiguous(), splits=1)


@pytest.mark.gpu
def test_search_argument_errors(ctx, torch_cuda):
    torch = torch_cuda
    from raytracer_amd import capi, knn_tree
    P = torch.rand(257, 16, device="cuda")
    Q = torch.rand(10, 16, device="cuda")
    d = torch.empty(1, 10, 16, device="cuda")
    i = torch.empty(1, 10, 16, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    L, h = capi.load(), ctx._h
    args = lambda p, q: (h, p, 256, q, 10, 0, 1, d.data_ptr(), i.data_ptr(), None)
    assert L.rt_dp_knn_euclid16(*args(P.data_ptr(), Q.data_ptr())) == capi.OK
    assert L.rt_dp_knn_euclid16(*args(P.data_ptr() + 4, Q.data_ptr())) == capi.EALIGN      # 8-byte alignment of the coordinates
    assert L.rt_dp_knn_angular16(*args(P.data_ptr(), Q.data_ptr() + 4)) == capi.EALIGN
    assert L.rt_dp_knn_euclid16(h, P.data_ptr(), 256, Q.data_ptr(), 10, 0, 0, d.data_ptr(), i.data_ptr(), None) == capi.EINVAL   # splits = 0
    assert L.rt_dp_knn_euclid16(h, P.data_ptr(), 256, Q.data_ptr(), 10, 0xFFFFFFF0, 1, d.data_ptr(), i.data_ptr(), None) == capi.EINVAL  # index overflow
    assert L.rt_dp_knn_euclid16(h, P.data_ptr(), 256, None, 10, 0, 1, d.data_ptr(), i.data_ptr(), None) == capi.ENULL
    base = capi.Context(support_euclidean=0)
    assert L.rt_dp_knn_angular16(base._h, P.data_ptr(), 256, Q.data_ptr(), 10, 0, 1, d.data_ptr(), i.data_ptr(), None) == capi.EINVAL  # no angular op
    base.close()
    tree = knn_tree.KnnTree(P, leaf_size=64)
    hs = torch.zeros(10, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    tree.n_leaves = 3
    with pytest.raises(capi.RtDpError) as e:
        ctx.knn_tree_grouped_euclid16(tree, Q, hs, d[0], i[0])
    assert e.value.code == capi.EINVAL
    ctx.synchronize()


def test_tree_build_is_a_balanced_partition():
    import torch
    from raytracer_amd import knn_tree
    g = torch.Generator().manual_seed(4)
    P = torch.rand(3000, 16, generator=g) * 2 - 1
    P[17, 3] = float("nan")
    tree = knn_tree.KnnTree(P, leaf_size=64, index_base=7)
    assert tree.n_leaves == 64 and tree.points.shape == (64 * 64, 16)
    idx = tree.index.to(torch.int64) & 0xFFFFFFFF
    real = idx != knn_tree.PAD_INDEX
    assert int(real.sum()) == 3000 and sorted(idx[real].tolist()) == list(range(7, 3007))  # a permutation of the points
    assert torch.equal(torch.nan_to_num(tree.points[real], nan=123.0), torch.nan_to_num(P[(idx[real] - 7)], nan=123.0))
    assert bool(torch.isinf(tree.points[~real]).all())
    # every real point lies inside the box of its leaf; the NaN coordinate opened its box
    lo = tree.leaf_lo.repeat_interleave(64, 0)[real]
    hi = tree.leaf_hi.repeat_interleave(64, 0)[real]
    pts = tree.points[real]
    ok = ((pts >= lo) & (pts <= hi)) | torch.isnan(pts)
    assert bool(ok.all())
    leaf_of_nan = int(torch.nonzero(torch.isnan(tree.points).any(1))[0]) // 64
    assert tree.leaf_lo[leaf_of_nan, 3] == float("-inf") and tree.leaf_hi[leaf_of_nan, 3] == float("inf")
    home = tree.home_leaf(P[:200].nan_to_num(0.0))
    assert int(home.min()) >= 0 and int(home.max()) < 64


WORKER = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from raytracer_amd import dist_util, knn
from oracle import dp_oracle as O
world, rank, _ = dist_util.init("gloo")
rng = np.random.default_rng(3)
pts = rng.uniform(-1, 1, (3000, 16)).astype(np.float32)
pts[100] = pts[2900]                      # a tie across the shard boundary
qry = rng.uniform(-1, 1, (64, 16)).astype(np.float32)
lo, hi = (rank * 3000) // world, ((rank + 1) * 3000) // world

def local(p, q, base):                    # CPU stand-in for the GPU kernel: brute force with the oracle's beat
    d, i = O.knn_bruteforce(p.numpy(), q.numpy(), 16, base)
    return knn.pack_keys(torch.from_numpy(d), torch.from_numpy(i.astype(np.int64)).to(torch.int32))

d, i = knn.search(None, torch.from_numpy(pts[lo:hi]), torch.from_numpy(qry), lo, local_fn=local)
wd, wi = O.knn_bruteforce(pts, qry)
assert (d.numpy().view(np.uint32) == wd.view(np.uint32)).all() and (i.numpy() == wi).all()
every = dist_util.gather_objects(i.numpy().tobytes())
assert all(e == every[0] for e in every)  # identical on every rank
if rank == 0:
    print("KNN-OK")
dist_util.shutdown()
"""


def test_all_gather_merge_two_ranks_gloo(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "KNN-OK" in out.stdout
