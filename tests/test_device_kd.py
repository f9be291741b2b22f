"""rt_dp_knn_tree_build_device / rt_dp_knn_home_leaf: the index built on the GPU is a balanced partition with enclosing
boxes, its split planes lead stored points to their leaves, the device home-leaf kernel equals the host descent, and
both tree searches on it return what the exhaustive kernel returns."""
import numpy as np
import pytest

from raytracer_amd import capi

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32


def host_view(t):
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in t.items()}


@pytest.mark.parametrize("n,leaf", [(3, 4), (3000, 64), (5000, 4), (40000, 128)])
def test_device_kd_index_is_a_balanced_partition(ctx, torch_cuda, n, leaf):
    torch = torch_cuda
    rng = np.random.default_rng(n + leaf)
    P = rng.uniform(-1, 1, (n, 16)).astype(f32)
    if n >= 3000:
        P[17, 3] = np.nan
        P[18, 0] = np.inf
        P[19, 1] = -np.inf
        P[20] = P[21]                      # duplicates
        P[22, 5] = -0.0
        P[100:140, 2] = 0.25               # many equal keys in one dimension
    t = host_view(ctx.knn_tree_build_device(torch.from_numpy(P).cuda(), leaf, index_base=7))
    L = t["n_leaves"]
    assert L == capi.load().rt_dp_knn_tree_leaves(n, leaf) and t["points"].shape == (L * leaf, 16)
    idx = t["index"].view(u32)
    real = idx != 0xFFFFFFFF
    assert real.sum() == n and sorted(idx[real].tolist()) == list(range(7, n + 7))
    assert (t["points"][real].view(u32) == P[idx[real] - 7].view(u32)).all()
    assert np.isposinf(t["points"][~real]).all()
    lo = np.repeat(t["leaf_lo"], leaf, axis=0)[real]
    hi = np.repeat(t["leaf_hi"], leaf, axis=0)[real]
    pts = t["points"][real]
    with np.errstate(invalid="ignore"):
        assert (((pts >= lo) & (pts <= hi)) | np.isnan(pts)).all()
    # exact boxes: every finite bound is attained by a stored point of the leaf
    per_leaf = t["points"].reshape(L, leaf, 16)
    mask = real.reshape(L, leaf, 1)
    with np.errstate(invalid="ignore"):
        want_lo = np.where(mask, per_leaf, np.inf).min(1)
        want_hi = np.where(mask, per_leaf, -np.inf).max(1)
    has_nan = np.isnan(np.where(mask, per_leaf, 0.0)).any(1)
    assert (np.where(has_nan, -np.inf, want_lo) == t["leaf_lo"]).all() and (np.where(has_nan, np.inf, want_hi) == t["leaf_hi"]).all()
    if n >= 3000:
        leaf_of_nan = int(np.nonzero(np.isnan(t["points"]).any(1))[0][0]) // leaf
        assert t["leaf_lo"][leaf_of_nan, 3] == -np.inf and t["leaf_hi"][leaf_of_nan, 3] == np.inf
    if L > 1:
        # descending the split planes with a stored point ends in the leaf that stores it (ties on a plane aside)
        finite = real & np.isfinite(t["points"]).all(1)
        host_tree = dict(t, split_dim=t["split_dim"].view(u32), index=idx)
        home = capi.knn_home_leaf_host(np.ascontiguousarray(t["points"][finite]), host_tree)
        at = np.nonzero(finite)[0] // leaf
        assert (home == at).mean() > 0.99
        # the device descent is the host descent
        Q = rng.uniform(-1.2, 1.2, (5001, 16)).astype(f32)
        Q[5, 2] = np.nan
        dt = {k: (torch.from_numpy(np.ascontiguousarray(v)).cuda() if isinstance(v, np.ndarray) else v) for k, v in t.items()}
        got = ctx.knn_home_leaf(torch.from_numpy(Q).cuda(), dt).cpu().numpy().view(u32)
        assert (got == capi.knn_home_leaf_host(Q, host_tree)).all()


@pytest.mark.parametrize("n,m,leaf", [(9, 5, 4), (5000, 300, 64), (30000, 700, 128)])
def test_searches_on_the_device_built_index_equal_the_exhaustive_search(ctx, torch_cuda, n, m, leaf):
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
    t = ctx.knn_tree_build_device(P, leaf, index_base=100)
    tree = knn_tree.KnnTree.__new__(knn_tree.KnnTree)
    tree.points, tree.index, tree.leaf_lo, tree.leaf_hi = t["points"], t["index"], t["leaf_lo"], t["leaf_hi"]
    tree.n_leaves, tree.leaf_size = t["n_leaves"], leaf
    home = ctx.knn_home_leaf(Q, t)
    order = torch.argsort(home.to(torch.int64), stable=True)
    qs, hs = Q[order].contiguous(), home[order].contiguous()
    d = torch.empty(1, m, 16, dtype=torch.float32, device="cuda")
    i = torch.empty(1, m, 16, dtype=torch.int32, device="cuda")
    b = torch.empty(m, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.knn_tree_euclid16(tree, qs, hs, b, d, i, splits=1)
    ctx.synchronize()
    got = torch.empty_like(want)
    got[order] = knn.pack_keys(d[0], i[0])
    assert torch.equal(got, want)
    ctx.knn_tree_grouped_euclid16(tree, qs, hs, d[0], i[0], walk_leaves=8)
    got[order] = knn.pack_keys(d[0], i[0])
    assert torch.equal(got, want)
