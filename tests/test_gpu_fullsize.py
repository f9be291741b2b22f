"""Parity at the benchmark's full sizes (2^24 units per opcode, operands generated in HBM).

Each opcode is compared bit for bit with oracle B over the whole batch (the scalar
C model needs about a second per 2^24 ops), and additionally through properties
that do not depend on the oracle: the reference testbench's ordering rules
(Datapath_test.scala:548-562), consistency of the packed flag word, independence
of ops (a permuted batch gives the permuted results) and run-to-run identity.
"""
import numpy as np
import pytest

from raytracer_amd import capi, synth

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu
N = 1 << 24


def _tiled_results(ctx, torch, arrays, n, words_out, derive=False, tri=False):
    """The same ops through the tiled layout: packed on the device, evaluated, unpacked into plain int32 arrays."""
    operands = torch.empty(capi.tiled_words(n, len(arrays)), dtype=torch.int32, device="cuda")
    results = torch.empty(capi.tiled_words(n, words_out), dtype=torch.int32, device="cuda")
    out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(words_out)]
    b = capi.Batch()
    (b.set_tri_tiled if tri else b.set_box_tiled)(operands, results, n)
    torch.cuda.synchronize()
    ctx.tile_pack(arrays, operands, n)
    ctx.launch(b)
    ctx.tile_unpack(results, out, n)
    ctx.synchronize()
    return out


def _box_batch(torch, fields, n):
    tm = torch.empty(4, n, dtype=torch.float32, device="cuda")
    flags = torch.empty(n, dtype=torch.int32, device="cuda")
    b = capi.Batch()
    b.set_box(fields, [tm[k] for k in range(4)], flags, n)
    return b, tm, flags


def test_box_2p24_against_oracle_and_properties(ctx, oracle, torch_cuda):
    torch = torch_cuda
    fields = synth.box_stream(N, 20241008, "cuda")
    b, tm, flags = _box_batch(torch, fields, N)
    ctx.eval(b)
    # properties on the device, all 2^24 ops
    assert bool((tm[0] >= 0).all()) and bool((tm[1:] >= tm[:-1]).all())          # t monotone, :550
    isect = torch.stack([(flags >> k) & 1 for k in range(4)])
    assert bool((isect[:-1] >= isect[1:]).all())                                   # hits precede misses, :553-556
    assert bool((isect.bool() == torch.isfinite(tm)).all())                        # miss <=> +Inf key
    idx = torch.stack([(flags >> (4 + 2 * k)) & 3 for k in range(4)])
    assert bool((idx.sort(dim=0).values == torch.arange(4, device="cuda", dtype=idx.dtype)[:, None]).all())
    assert bool((flags >> 12 == 0).all())
    hit_rate = float(isect.float().mean())
    assert 0.05 < hit_rate < 0.6
    # full comparison with the oracle
    host = [f.cpu().numpy() for f in fields]
    wt, wf = oracle.box_soa(host)
    for k in range(4):
        assert_bits_equal(tm[k].cpu().numpy(), wt[k], "tmin_out[%d]" % k)
    assert_bits_equal(flags.cpu().numpy(), wf, "flags")
    # ops are independent: a permuted batch gives the permuted results; and a rerun is identical
    perm = torch.randperm(N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    b2, tm2, flags2 = _box_batch(torch, [f[perm].contiguous() for f in fields], N)
    ctx.eval(b2)
    assert bool((tm2.view(torch.int32) == tm[:, perm].view(torch.int32)).all()) and bool((flags2 == flags[perm]).all())
    keep = tm.clone()
    ctx.eval(b)
    assert bool((keep.view(torch.int32) == tm.view(torch.int32)).all())


def test_tri_2p24_against_oracle(ctx, oracle, torch_cuda):
    torch = torch_cuda
    fields, k = synth.tri_stream(N, 20241009, "cuda")
    tn = torch.empty(N, dtype=torch.float32, device="cuda")
    td = torch.empty(N, dtype=torch.float32, device="cuda")
    hit = torch.empty(N, dtype=torch.int32, device="cuda")
    b = capi.Batch()
    b.set_tri(fields, k, tn, td, hit, N)
    ctx.eval(b)
    assert bool(((hit == 0) | (hit == 1)).all())
    assert bool((td[hit == 1] != 0).all()) and bool((tn[hit == 1] >= 0).all())    # stage 10, :1106-1114
    wn, wd, wh = oracle.tri_soa([f.cpu().numpy() for f in fields], k.cpu().numpy().view(np.uint32))
    assert_bits_equal(tn.cpu().numpy(), wn, "t_num")
    assert_bits_equal(td.cpu().numpy(), wd, "t_denom")
    assert_bits_equal(hit.cpu().numpy(), wh, "triangle_hit")
    assert 0.3 < float(hit.float().mean()) < 0.7


def test_distance_beats_2p24_against_oracle(ctx, oracle, torch_cuda):
    torch = torch_cuda
    a, bb, ctl = synth.vec_stream(N, 16, 20241010, "cuda")
    out = torch.empty(ctl.shape[0], dtype=torch.float32, device="cuda")
    b = capi.Batch()
    b.set_euclid(a, bb, ctl, out)
    ctx.reset()
    ctx.eval(b)
    want, acc = oracle.euclid_beats(a.cpu().numpy(), bb.cpu().numpy(), ctl.cpu().numpy().view(np.uint32))
    assert_bits_equal(out.cpu().numpy(), want, "euclidean_accumulator")
    assert_bits_equal(ctx.get_accum()[0:1], np.array([acc], np.float32), "register")
    # within a pair the running sum of squares never decreases
    last = ((ctl >> 16) & 1).bool()
    assert bool((out[1:][~last[:-1]] >= out[:-1][~last[:-1]]).all())
    del a, bb, out
    q, c, ctl8 = synth.vec_stream(N, 8, 20241011, "cuda")
    dot = torch.empty(ctl8.shape[0], dtype=torch.float32, device="cuda")
    norm = torch.empty(ctl8.shape[0], dtype=torch.float32, device="cuda")
    b = capi.Batch()
    b.set_angular(q, c, ctl8, dot, norm)
    ctx.eval(b)
    wd, wn, _, _ = oracle.angular_beats(q.cpu().numpy(), c.cpu().numpy(), ctl8.cpu().numpy().view(np.uint32))
    assert_bits_equal(dot.cpu().numpy(), wd, "angular_dot_product")
    assert_bits_equal(norm.cpu().numpy(), wn, "angular_norm")


def test_mixed_batch_one_launch(ctx, oracle, torch_cuda):
    """All four opcode segments in one rt_dp_launch on a caller stream (2^22 units each)."""
    torch = torch_cuda
    n = 1 << 22
    s = torch.cuda.Stream()
    fields = synth.box_stream(n, 7, "cuda")
    b, tm, flags = _box_batch(torch, fields, n)
    tf, k = synth.tri_stream(n, 8, "cuda")
    tn, td = torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda")
    hit = torch.empty(n, dtype=torch.int32, device="cuda")
    b.set_tri(tf, k, tn, td, hit, n)
    a, bb, ctl = synth.vec_stream(n, 16, 9, "cuda")
    eo = torch.empty(ctl.shape[0], dtype=torch.float32, device="cuda")
    b.set_euclid(a, bb, ctl, eo)
    q, c, ctl8 = synth.vec_stream(n, 8, 10, "cuda")
    dot, norm = torch.empty(ctl8.shape[0], dtype=torch.float32, device="cuda"), torch.empty(ctl8.shape[0], dtype=torch.float32, device="cuda")
    b.set_angular(q, c, ctl8, dot, norm)
    torch.cuda.synchronize()
    ctx.reset()
    ctx.launch(b, s.cuda_stream)
    s.synchronize()
    wt, wf = oracle.box_soa([f.cpu().numpy() for f in fields])
    assert_bits_equal(tm[0].cpu().numpy(), wt[0], "tmin_out[0]")
    assert_bits_equal(flags.cpu().numpy(), wf, "flags")
    wn, wd, wh = oracle.tri_soa([f.cpu().numpy() for f in tf], k.cpu().numpy().view(np.uint32))
    assert_bits_equal(tn.cpu().numpy(), wn, "t_num")
    assert_bits_equal(hit.cpu().numpy(), wh, "hit")
    we, _ = oracle.euclid_beats(a.cpu().numpy(), bb.cpu().numpy(), ctl.cpu().numpy().view(np.uint32))
    assert_bits_equal(eo.cpu().numpy(), we, "euclidean_accumulator")
    wdot, wnorm, _, _ = oracle.angular_beats(q.cpu().numpy(), c.cpu().numpy(), ctl8.cpu().numpy().view(np.uint32))
    assert_bits_equal(dot.cpu().numpy(), wdot, "dot")
    assert_bits_equal(norm.cpu().numpy(), wnorm, "norm")


def test_edge_class_streams_2p24_against_oracle(ctx, oracle, torch_cuda):
    """Configs 2 and 3 with their edge classes at full size: 2^24 box ops and 2^24 triangle ops with 1 % of the ops
    in every edge class (drawn on the device, synth.BOX_EDGE_CLASSES / TRI_EDGE_CLASSES), compared with oracle B over
    the whole batch."""
    torch = torch_cuda
    fields = synth.box_stream(N, 20241008, "cuda", 0.01)
    # the classes are there: signed-zero directions, +Inf boxes, NaN operands, subnormal scenes, ties
    zero_dir = (fields[3] == 0) | (fields[4] == 0) | (fields[5] == 0)
    assert 0.008 < float(zero_dir.float().mean()) < 0.02
    assert 0.008 < float(torch.isinf(fields[10]).float().mean() * 4) < 0.03
    assert int(torch.isnan(torch.stack(fields)).sum()) > 1000
    assert 0.008 < float(((fields[10] != 0) & (fields[10].abs() < 1e-37)).float().mean()) < 0.02
    assert 0.008 < float(((fields[10] == fields[16]) & (fields[10] == fields[22]) & (fields[10] == fields[28])).float().mean()) < 0.03
    b, tm, flags = _box_batch(torch, fields, N)
    ctx.eval(b)
    wt, wf = oracle.box_soa([f.cpu().numpy() for f in fields])
    for k in range(4):
        assert_bits_equal(tm[k].cpu().numpy(), wt[k], "tmin_out[%d]" % k)
    assert_bits_equal(flags.cpu().numpy(), wf, "flags")
    assert not bool(torch.isnan(tm).any())                       # a NaN never reaches tmin_out (a hit needs tmin < tmax)
    # the tiled layout gives the same bits, all 2^24 ops with their edge classes
    tiled = _tiled_results(ctx, torch, fields, N, 5)
    for k in range(4):
        assert bool((tiled[k] == tm[k].view(torch.int32)).all()), "tiled tmin_out[%d]" % k
    assert bool((tiled[4] == flags).all()), "tiled flags"
    del fields, b, tm, flags, tiled

    tf, k = synth.tri_stream(N, 20241009, "cuda", 0.01)
    assert 0.008 < float((((k >> 4) & 3) == 3).float().mean()) < 0.02          # kz encoded as 3
    tn = torch.empty(N, dtype=torch.float32, device="cuda")
    td = torch.empty(N, dtype=torch.float32, device="cuda")
    hit = torch.empty(N, dtype=torch.int32, device="cuda")
    b = capi.Batch()
    b.set_tri(tf, k, tn, td, hit, N)
    ctx.eval(b)
    wn, wd, wh = oracle.tri_soa([f.cpu().numpy() for f in tf], k.cpu().numpy().view(np.uint32))
    assert_bits_equal(tn.cpu().numpy(), wn, "t_num")
    assert_bits_equal(td.cpu().numpy(), wd, "t_denom")
    assert_bits_equal(hit.cpu().numpy(), wh, "triangle_hit")
    assert 0.015 < float((td == 0).float().mean()) < 0.05          # degenerate triangles: t_denom == +-0, never a hit
    assert not bool(((td == 0) & (hit == 1)).any())
    nan_out = torch.isnan(tn) | torch.isnan(td)
    assert int(nan_out.sum()) > 1000 and bool((tn.view(torch.int32)[torch.isnan(tn)] == 0x7FC00000).all())   # canonical NaN
    tiled = _tiled_results(ctx, torch, list(tf) + [k], N, 3, tri=True)
    assert bool((tiled[0] == tn.view(torch.int32)).all()) and bool((tiled[1] == td.view(torch.int32)).all()) and bool((tiled[2] == hit).all())


def _windows(n, rng, count=64, size=1 << 16, ends=1 << 20):
    w = [(0, min(ends, n)), (max(0, n - ends), n)]
    w += [(int(s), int(s) + size) for s in rng.integers(0, n - size, count)]
    return w


def test_mixed_stream_2p28_properties_and_sampled_oracle(ctx, oracle, torch_cuda):
    """Config 4 at its stated size on one GPU: four segments of 2^26 units (the Euclidean operand arrays are 4 GiB each,
    so every index computation beyond 2^32 bytes is exercised), ONE launch.  Everything is checked through
    size-independent properties on the device; oracle B is run on the first and last 2^20 units of each segment and on
    64 random windows of 2^16 units (the distance windows start from the accumulator value the GPU delivered in front of
    them, or +0 behind a reset).  Measured on the MI355X box: 1.2 s for this test, 3.7 s for the edge-class streams, 4.0 s for
    the k-NN configuration."""
    torch = torch_cuda
    n = 1 << 26
    rng = np.random.default_rng(28)
    u32 = np.uint32
    fields = synth.box_stream(n, 20241008, "cuda")
    b, tm, flags = _box_batch(torch, fields, n)
    tf, k = synth.tri_stream(n, 20241009, "cuda")
    tn, td = torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda")
    hit = torch.empty(n, dtype=torch.int32, device="cuda")
    b.set_tri(tf, k, tn, td, hit, n)
    a, bb, ctl = synth.vec_stream(n, 16, 20241010, "cuda")
    ne = int(ctl.shape[0])
    assert a.numel() * 4 >= 1 << 32
    eo = torch.empty(ne, dtype=torch.float32, device="cuda")
    b.set_euclid(a, bb, ctl, eo)
    q, c, ctl8 = synth.vec_stream(n, 8, 20241011, "cuda")
    na = int(ctl8.shape[0])
    dot, norm = torch.empty(na, dtype=torch.float32, device="cuda"), torch.empty(na, dtype=torch.float32, device="cuda")
    b.set_angular(q, c, ctl8, dot, norm)
    torch.cuda.synchronize()
    ctx.reset()
    ctx.eval(b)
    assert np.array_equal(ctx.get_accum(), np.zeros(3, np.float32))        # both streams end on a reset beat

    # properties over everything
    assert bool((tm[0] >= 0).all()) and bool((tm[1:] >= tm[:-1]).all())
    isect = torch.stack([(flags >> j) & 1 for j in range(4)])
    assert bool((isect[:-1] >= isect[1:]).all()) and bool((isect.bool() == torch.isfinite(tm)).all())
    del isect
    assert bool(((hit == 0) | (hit == 1)).all()) and bool((td[hit == 1] != 0).all()) and bool((tn[hit == 1] >= 0).all())
    last = ((ctl >> 16) & 1).bool()
    assert bool((eo[1:][~last[:-1]] >= eo[:-1][~last[:-1]]).all()) and bool((eo >= 0).all())
    last8 = ((ctl8 >> 16) & 1).bool()
    assert bool((norm[1:][~last8[:-1]] >= norm[:-1][~last8[:-1]]).all())
    del last, last8
    # linearity of the last beats: the final accumulator of a pair equals a re-run of that pair alone (sampled below through the oracle)

    for s, e in _windows(n, rng):
        wt, wf = oracle.box_soa([f[s:e].cpu().numpy() for f in fields])
        for j in range(4):
            assert_bits_equal(tm[j, s:e].cpu().numpy(), wt[j], "tmin_out[%d] of ops %d..%d" % (j, s, e))
        assert_bits_equal(flags[s:e].cpu().numpy(), wf, "flags of ops %d..%d" % (s, e))
        wn, wd, wh = oracle.tri_soa([f[s:e].cpu().numpy() for f in tf], k[s:e].cpu().numpy().view(u32))
        assert_bits_equal(tn[s:e].cpu().numpy(), wn, "t_num of ops %d..%d" % (s, e))
        assert_bits_equal(td[s:e].cpu().numpy(), wd, "t_denom")
        assert_bits_equal(hit[s:e].cpu().numpy(), wh, "triangle_hit")
    for s, e in _windows(ne, rng):
        acc0 = 0.0 if s == 0 or (int(ctl[s - 1]) >> 16) & 1 else float(eo[s - 1])
        want, _ = oracle.euclid_beats(a[s:e].cpu().numpy(), bb[s:e].cpu().numpy(), ctl[s:e].cpu().numpy().view(u32), acc0)
        assert_bits_equal(eo[s:e].cpu().numpy(), want, "euclidean_accumulator of beats %d..%d" % (s, e))
    for s, e in _windows(na, rng):
        first = s == 0 or (int(ctl8[s - 1]) >> 16) & 1
        d0, n0 = (0.0, 0.0) if first else (float(dot[s - 1]), float(norm[s - 1]))
        wd, wn, _, _ = oracle.angular_beats(q[s:e].cpu().numpy(), c[s:e].cpu().numpy(), ctl8[s:e].cpu().numpy().view(u32), d0, n0)
        assert_bits_equal(dot[s:e].cpu().numpy(), wd, "angular_dot_product of beats %d..%d" % (s, e))
        assert_bits_equal(norm[s:e].cpu().numpy(), wn, "angular_norm")


def test_knn_full_configuration(ctx, oracle, torch_cuda):
    """Config 5 at its stated size: 2^20 points, 2^16 queries, U(-1,1)^16, k = 16.  The exhaustive search, the tree
    walk and the leaf-major search deliver the same lists for every query (bit for bit, distances and indices), and 256
    sampled queries equal a brute force over all points with oracle B's beat as the distance."""
    torch = torch_cuda
    from raytracer_amd import knn, knn_tree
    g = torch.Generator(device="cuda")
    g.manual_seed(20241012)
    pts = torch.rand(1 << 20, 16, generator=g, device="cuda") * 2 - 1
    qry = torch.rand(1 << 16, 16, generator=g, device="cuda") * 2 - 1
    d0, i0 = knn.unpack_keys(knn.local_topk(ctx, pts, qry, 0))
    tree = knn_tree.KnnTree.build_device(ctx, pts, leaf_size=128, index_base=0)
    d1, i1 = knn_tree.search(ctx, tree, qry, grouped=False)
    d2, i2 = knn_tree.search(ctx, tree, qry, grouped=True, walk_leaves=128)
    for d, i, what in ((d1, i1, "tree walk"), (d2, i2, "leaf-major search")):
        assert torch.equal(d.view(torch.int32), d0.view(torch.int32)), what
        assert torch.equal(i, i0), what
    assert bool((d0[:, 1:] >= d0[:, :-1]).all()) and bool(torch.isfinite(d0).all())
    P = pts.cpu().numpy()
    rng = np.random.default_rng(5)
    ctl = np.full(P.shape[0], 0x1FFFF, np.uint32)
    for qi in rng.choice(1 << 16, 256, replace=False):
        qv = qry[qi].cpu().numpy()
        dist, _ = oracle.euclid_beats(np.broadcast_to(qv, P.shape).copy(), P, ctl)
        thr = np.partition(dist, 15)[15]
        cand = np.flatnonzero(dist <= thr)
        order = cand[np.lexsort((cand, dist[cand]))][:16]
        assert np.array_equal(i0[qi].cpu().numpy(), order), "query %d" % qi
        assert_bits_equal(d0[qi].cpu().numpy(), dist[order], "distances of query %d" % qi)
