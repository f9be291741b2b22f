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
