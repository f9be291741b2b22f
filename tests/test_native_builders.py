"""The host-side builders of the C ABI (rt_dp_knn_tree_build_host, rt_dp_bvh4_build_host): structural invariants on
the CPU, and on the GPU the searches / traversals that read their output against the exhaustive kernel / the oracle model."""
import numpy as np
import pytest

from raytracer_amd import bvh, capi

f32, u32 = np.float32, np.uint32


def test_native_kd_tree_is_a_balanced_partition_with_enclosing_boxes():
    rng = np.random.default_rng(4)
    P = rng.uniform(-1, 1, (3000, 16)).astype(f32)
    P[17, 3] = np.nan
    P[18, 0] = np.inf
    t = capi.knn_tree_build_host(P, leaf_size=64, index_base=7)
    assert t["n_leaves"] == 64 and t["points"].shape == (4096, 16)
    idx = t["index"]
    real = idx != 0xFFFFFFFF
    assert real.sum() == 3000 and sorted(idx[real].tolist()) == list(range(7, 3007))
    assert (t["points"][real].view(u32) == P[idx[real] - 7].view(u32)).all()
    assert np.isposinf(t["points"][~real]).all()
    lo = np.repeat(t["leaf_lo"], 64, axis=0)[real]
    hi = np.repeat(t["leaf_hi"], 64, axis=0)[real]
    pts = t["points"][real]
    with np.errstate(invalid="ignore"):
        assert (((pts >= lo) & (pts <= hi)) | np.isnan(pts)).all()
    leaf_of_nan = int(np.nonzero(np.isnan(t["points"]).any(1))[0][0]) // 64
    assert t["leaf_lo"][leaf_of_nan, 3] == -np.inf and t["leaf_hi"][leaf_of_nan, 3] == np.inf
    # descending the split planes with a stored point ends in the leaf that stores it (ties on a plane aside)
    finite = real & np.isfinite(t["points"]).all(1)
    home = capi.knn_home_leaf_host(t["points"][finite], t)
    at = np.nonzero(finite)[0] // 64
    assert (home == at).mean() > 0.99
    # pads collect behind the real points: the real points fill a prefix of the leaves
    per_leaf = real.reshape(64, 64).sum(1)
    assert (np.diff((per_leaf > 0).astype(int)) <= 0).all()
    tiny = capi.knn_tree_build_host(P[:3], leaf_size=4)
    assert tiny["n_leaves"] == 1 and (tiny["index"] != 0xFFFFFFFF).sum() == 3


def test_native_bvh4_is_a_partition_with_enclosing_boxes():
    tris = bvh.random_scene(3001, seed=1)
    nodes, tris12, order = capi.bvh4_build_host(tris, leaf_size=4)
    assert sorted(order.tolist()) == list(range(3001)) and (tris12[:, :9] == tris[order]).all()
    refs = nodes[:, 24:28].copy().view(u32)
    seen = np.zeros(3001, bool)
    for i in range(nodes.shape[0]):
        for c in range(4):
            r = int(refs[i, c])
            box = nodes[i, 6 * c:6 * c + 6]
            if r == bvh.NONE:
                assert np.isposinf(box).all()
                continue
            if r & bvh.LEAF:
                first, cnt = (r & 0x7FFFFFFF) >> 3, (r & 7) + 1
                assert cnt <= 4 and not seen[first:first + cnt].any()
                seen[first:first + cnt] = True
                v = tris12[first:first + cnt, :9].reshape(-1, 3)
                assert (v.min(0) == box[:3]).all() and (v.max(0) == box[3:]).all()
            else:
                assert i < r < nodes.shape[0]
                child = nodes[r, :24].reshape(4, 6)
                live = refs[r] != bvh.NONE
                assert (child[live, :3].min(0) == box[:3]).all() and (child[live, 3:].max(0) == box[3:]).all()
    assert seen.all()
    assert capi.load().rt_dp_bvh4_build_host(None, 5, 4, None, None, None, None) == capi.ENULL
    assert capi.load().rt_dp_knn_tree_build_host(None, 0, 6, 0, None, None, None, None, None, None) in (capi.ENULL, capi.EINVAL)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,leaf", [(9, 5, 4), (5000, 300, 64), (30000, 700, 128)])
def test_searches_on_the_native_index_equal_the_exhaustive_search(ctx, torch_cuda, n, m, leaf):
    torch = torch_cuda
    from raytracer_amd import knn, knn_tree
    rng = np.random.default_rng(n + m)
    pts = rng.uniform(-1, 1, (n, 16)).astype(f32)
    qry = rng.uniform(-1, 1, (m, 16)).astype(f32)
    if n > 100:
        pts[5] = pts[6]
        pts[7, 2] = np.nan
    P, Q = torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda()
    want = knn.local_topk(ctx, P, Q, 100)
    t = capi.knn_tree_build_host(pts, leaf_size=leaf, index_base=100)
    tree = knn_tree.KnnTree.__new__(knn_tree.KnnTree)
    tree.points = torch.from_numpy(t["points"]).cuda()
    tree.index = torch.from_numpy(t["index"].view(np.int32)).cuda()
    tree.leaf_lo, tree.leaf_hi = torch.from_numpy(t["leaf_lo"]).cuda(), torch.from_numpy(t["leaf_hi"]).cuda()
    tree.n_leaves, tree.leaf_size = t["n_leaves"], leaf
    home = capi.knn_home_leaf_host(qry, t)
    order = np.argsort(home, kind="stable")
    qs = torch.from_numpy(qry[order]).cuda()
    hs = torch.from_numpy(home[order].view(np.int32)).cuda()
    d = torch.empty(1, m, 16, dtype=torch.float32, device="cuda")
    i = torch.empty(1, m, 16, dtype=torch.int32, device="cuda")
    b = torch.empty(m, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.knn_tree_euclid16(tree, qs, hs, b, d, i, splits=1)
    ctx.synchronize()
    inv = torch.from_numpy(order).cuda()
    got = torch.empty_like(want)
    got[inv] = knn.pack_keys(d[0], i[0])
    assert torch.equal(got, want)
    ctx.knn_tree_grouped_euclid16(tree, qs, hs, d[0], i[0], walk_leaves=8)
    got[inv] = knn.pack_keys(d[0], i[0])
    assert torch.equal(got, want)


@pytest.mark.gpu
def test_traversal_of_the_native_hierarchy_matches_the_oracle_model(ctx, oracle, torch_cuda):
    from trace_model import trace_closest_dfs
    from test_trace import run_gpu
    tris = bvh.random_scene(4000, seed=12)
    nodes, tris12, order = capi.bvh4_build_host(tris, leaf_size=3)
    org, dirs = bvh.rays_towards(tris, 8000, seed=13, jitter=5.0)
    want_t, want_i, want_stats = trace_closest_dfs(oracle, nodes, tris12, org, dirs, 4000.0)
    got_t, got_i, stats = run_gpu(ctx, torch_cuda, nodes, tris12, org, dirs, 4000.0)
    assert (got_i == want_i).all() and (got_t.view(u32) == want_t.view(u32)).all() and stats == want_stats
    assert (got_i != 0xFFFFFFFF).mean() > 0.15
