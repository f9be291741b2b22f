#!/usr/bin/env python3
"""Parity soak of the drivers above the op kernels, with randomly drawn shapes: record streams (all three modes, whole
and in random pieces), the k-NN searches (exhaustive, tree walk, leaf-major; Euclidean and angular), the
closest-hit traversal (both walks) and the hit-mask compaction, each against the CPU oracle or a model built from
it.  Not part of the test suite (minutes of run time):

    python tests/soak/soak_drivers.py [rounds] [first seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import dp_oracle as O  # noqa: E402  (the checker)
from raytracer_amd import bvh, capi, knn, knn_tree  # noqa: E402
from oracle import gen  # noqa: E402
from test_gpu_parity import _mixed_stream  # noqa: E402
from test_knn import clustered_cloud, make_cloud  # noqa: E402
from test_box_compact import model as compact_model  # noqa: E402
from trace_model import trace_closest, trace_closest_dfs  # noqa: E402
import torch  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
u32, f32 = np.uint32, np.float32
O.build()
bad = 0
t0 = time.time()


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(u32), np.ascontiguousarray(b).view(u32))


def soak_stream(rng, seed):
    sizes = [int(rng.integers(0, hi)) for hi in (30000, 30000, 4000, 4000)]
    if sum(sizes) == 0:
        sizes[0] = 1
    stream = _mixed_stream(seed, *sizes, edge=float(rng.choice([0.0, 0.05, 0.5])))
    want, state = O.eval_bundles("B", stream.view(O.IN_BUNDLE).reshape(-1), extended=True)
    want = want.view(u32).reshape(-1, 21)
    ok = True
    for mode in (1, 2, 3):
        c = capi.Context()
        c.set_option("stream_mode", mode)
        n = stream.shape[0]
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, int(rng.integers(0, 6)))]))
        got = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            d_in = torch.from_numpy(np.ascontiguousarray(stream[a:b]).view(np.int32)).cuda()
            d_out = torch.full((b - a, 21), -1, dtype=torch.int32, device="cuda")
            c.eval_bundles_device(d_in, d_out)
            got.append(d_out.cpu().numpy().view(u32))
        ok &= same(np.concatenate(got), want) and tuple(int(x) for x in c.get_accum().view(u32)) == state
        ok &= c.get_option("stream_giveups") == 0
        # the host entry point (pinned single-synchronisation path for small streams) from the same registers
        c.reset()
        ok &= same(c.eval_bundles(stream), want) and tuple(int(x) for x in c.get_accum().view(u32)) == state
        c.close()
    return ok, "stream %s" % (sizes,)


def soak_knn(rng, seed):
    n, m = int(rng.integers(1, 30000)), int(rng.integers(1, 1500))
    kind = rng.choice(["uniform", "clustered", "grid"])
    if kind == "uniform" or n < 100 or m < 8:
        kind = "uniform"
        pts, qry = make_cloud(n, m, seed, specials=m >= 2)
    elif kind == "clustered":
        pts, qry = clustered_cloud(n, m, seed)
    else:  # few distinct coordinates: masses of exactly equal distances
        pts = rng.integers(-2, 3, (n, 16)).astype(f32)
        qry = rng.integers(-2, 3, (m, 16)).astype(f32)
    leaf = int(rng.choice([4, 16, 64, 128]))
    c = capi.Context()
    P, Q = torch.from_numpy(pts).cuda(), torch.from_numpy(qry).cuda()
    base = int(rng.integers(0, 1 << 20))
    want = knn.local_topk(c, P, Q, base, splits=int(rng.integers(1, 9)))
    ok = True
    sub = rng.choice(m, min(m, 64), replace=False)
    wd, wi = O.knn_bruteforce(pts, qry[sub], 16, base)
    d, i = knn.unpack_keys(want)
    ok &= np.array_equal(i.cpu().numpy()[sub], wi) and same(d.cpu().numpy()[sub], wd)
    tree = knn_tree.KnnTree(P, leaf_size=leaf, index_base=base)
    ok &= bool(torch.equal(knn_tree.local_topk(c, tree, Q, splits=None), want))
    ok &= bool(torch.equal(knn_tree.local_topk_grouped(c, tree, Q, walk_leaves=int(rng.choice([1, 8, 128]))), want))
    dtree = knn_tree.KnnTree.build_device(c, P, leaf_size=leaf, index_base=base)   # the index built by the HIP builder
    ok &= bool(torch.equal(knn_tree.local_topk(c, dtree, Q, splits=None), want))
    ok &= bool(torch.equal(knn_tree.local_topk_grouped(c, dtree, Q, walk_leaves=int(rng.choice([1, 8, 128]))), want))
    ga = knn.local_topk_angular(c, P, Q, base, splits=int(rng.integers(1, 5)))
    ad, ai = O.knn_bruteforce_angular(pts, qry[sub], 16, base)
    gd, gi = knn.unpack_signed_keys(ga)
    ok &= np.array_equal(gi.cpu().numpy()[sub], ai) and same(gd.cpu().numpy()[sub], ad)
    c.close()
    return ok, "knn %s n=%d m=%d leaf=%d" % (kind, n, m, leaf)


def soak_trace(rng, seed):
    n_tris, n_rays = int(rng.integers(1, 6000)), int(rng.integers(1, 12000))
    leaf = int(rng.integers(1, 9))
    tris = bvh.random_scene(n_tris, seed=seed, size=float(rng.choice([5.0, 60.0, 400.0])))
    nodes, tris12, _ = bvh.build_bvh4(tris, leaf_size=leaf)
    org, dirs = bvh.rays_towards(tris, n_rays, seed=seed + 1, jitter=float(rng.choice([0.0, 5.0, 50.0])))
    if n_rays > 8:
        dirs[1] = 0.0
        org[2] = np.nan
        dirs[3, int(rng.integers(0, 3))] = 0.0
    extent = float(rng.choice([4000.0, 700.0]))
    c = capi.Context()
    N, T = torch.from_numpy(nodes).cuda(), torch.from_numpy(tris12).cuda()
    o = [torch.from_numpy(np.ascontiguousarray(org[:, a])).cuda() for a in range(3)]
    d = [torch.from_numpy(np.ascontiguousarray(dirs[:, a])).cuda() for a in range(3)]
    ok = True
    for wavefront, model in ((False, trace_closest_dfs), (True, trace_closest)):
        wt, wi, wstats = model(O, nodes, tris12, org, dirs, extent)
        t = torch.empty(n_rays, dtype=torch.float32, device="cuda")
        i = torch.empty(n_rays, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        stats = c.trace_closest(N, T, o, d, extent, t, i, wavefront=wavefront)
        ok &= same(t.cpu().numpy(), wt) and np.array_equal(i.cpu().numpy().view(u32), wi) and stats == wstats
    # the same rays through a hierarchy built on the device: the walk equals the CPU model on those arrays
    DN, DT, _ = c.bvh4_build_device(torch.from_numpy(tris).cuda(), leaf)
    wt, wi, wstats = trace_closest_dfs(O, DN.cpu().numpy(), DT.cpu().numpy(), org, dirs, extent)
    t = torch.empty(n_rays, dtype=torch.float32, device="cuda")
    i = torch.empty(n_rays, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    stats = c.trace_closest(DN, DT, o, d, extent, t, i, wavefront=False)
    ok &= same(t.cpu().numpy(), wt) and np.array_equal(i.cpu().numpy().view(u32), wi) and stats == wstats
    c.close()
    return ok, "trace tris=%d rays=%d leaf=%d" % (n_tris, n_rays, leaf)


def soak_compact(rng, seed):
    n = int(rng.integers(1, 200000))
    fields = gen.box_ops(n, seed=seed, edge_frac=float(rng.choice([0.0, 0.3])))
    ref_t, ref_fl = O.box_soa(fields)
    ref_fl = np.asarray(ref_fl).view(u32)
    item = rng.integers(0, 2 ** 32, n, dtype=u32)
    children = rng.integers(0, 2 ** 32, (n, 4), dtype=u32)
    wi, wc, wt = compact_model(ref_t, ref_fl, item, children, True)
    c = capi.Context()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    tmin = [dev(t) for t in ref_t]
    flags = dev(ref_fl.view(np.int32))
    cap = max(1, wi.shape[0])
    oi = torch.empty(cap, dtype=torch.int32, device="cuda")
    oc = torch.empty(cap, dtype=torch.int32, device="cuda")
    ot = torch.empty(cap, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    count = c.box_compact(tmin, flags, oi, oc, ot, dev(item.view(np.int32)), dev(children.view(np.int32)))
    ok = count == wi.shape[0] and same(oi.cpu().numpy()[:count], wi) and same(oc.cpu().numpy()[:count], wc) and same(ot.cpu().numpy()[:count], wt)
    c.close()
    return ok, "compact n=%d" % n


for r in range(rounds):
    seed = seed0 + r
    rng = np.random.default_rng(50000 + seed)
    for fn in (soak_stream, soak_knn, soak_trace, soak_compact):
        ok, what = fn(rng, 70000 + 10 * seed)
        bad += 0 if ok else 1
        print("round %d %s: %s  (%.0f s)" % (seed, what, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
print("driver soak: %d rounds, %d mismatching" % (rounds, bad))
sys.exit(1 if bad else 0)
