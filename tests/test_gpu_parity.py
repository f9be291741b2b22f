"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Integer / byte outputs (flags, hit words) and every FP output are compared as
raw 32-bit patterns: no tolerance anywhere.
"""
import numpy as np
import pytest

from raytracer_amd import capi
from oracle import gen

from helpers import assert_bits_equal, hip_angular, hip_box, hip_euclid, hip_tri

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32


def check_box(ctx, oracle, fields, device, torch=None):
    tmin, flags = hip_box(ctx, fields, device, torch)
    wt, wf = oracle.box_soa(fields)
    for k in range(4):
        assert_bits_equal(tmin[k], wt[k], "tmin_out[%d]" % k)
    assert_bits_equal(flags, wf, "box flags")


def check_tri(ctx, oracle, fields, k, device, torch=None):
    tn, td, hit = hip_tri(ctx, fields, k, device, torch)
    wn, wd, wh = oracle.tri_soa(fields, k)
    assert_bits_equal(tn, wn, "t_num")
    assert_bits_equal(td, wd, "t_denom")
    assert_bits_equal(hit, wh, "triangle_hit")


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 256, 257, 1023, 1025, 4099])
@pytest.mark.parametrize("device", [False, True])
def test_box_ragged_sizes(ctx, oracle, torch_cuda, n, device):
    check_box(ctx, oracle, gen.box_ops(n, seed=100 + n, edge_frac=0.3), device, torch_cuda)


@pytest.mark.parametrize("v", [1, 2, 4])
@pytest.mark.parametrize("nt", [1, 2])
def test_box_kernel_shapes(ctx, oracle, torch_cuda, v, nt):
    ctx.set_tuning(ops_per_lane=v, nontemporal=nt)
    check_box(ctx, oracle, gen.box_ops(200003, seed=7, edge_frac=0.25), True, torch_cuda)


@pytest.mark.parametrize("wgs", [3, 4, 7, 16, 99])
def test_residency_caps_do_not_change_results(ctx, oracle, torch_cuda, wgs):
    """The dynamic-LDS residency cap of the op kernels is a scheduling knob: every legal value launches and gives the
    oracle's bits, for all four opcodes (batches above the sizes from which the default cap applies)."""
    ctx.set_option("wgs_per_cu", wgs)
    check_box(ctx, oracle, gen.box_ops((1 << 20) + 3, seed=70 + wgs, edge_frac=0.1), True, torch_cuda)
    tf, tk = gen.tri_ops((1 << 21) + 5, seed=80 + wgs, edge_frac=0.1)
    check_tri(ctx, oracle, tf, tk, True, torch_cuda)
    a, b, ctl, _ = gen.vec_beats(30000, 16, seed=90 + wgs, edge_frac=0.05)
    assert_bits_equal(hip_euclid(ctx, a, b, ctl, True, torch_cuda), oracle.euclid_beats(a, b, ctl)[0], "euclid")
    q, c, ctl8, _ = gen.vec_beats(30000, 8, seed=95 + wgs, edge_frac=0.05)
    gd, gn = hip_angular(ctx, q, c, ctl8, True, torch_cuda)
    wd, wn = oracle.angular_beats(q, c, ctl8)[:2]
    assert_bits_equal(gd, wd, "angular dot")
    assert_bits_equal(gn, wn, "angular norm")
    with pytest.raises(capi.RtDpError):
        ctx.set_option("wgs_per_cu", 2)


def test_box_random_1m(ctx, oracle, torch_cuda):
    check_box(ctx, oracle, gen.box_ops(1 << 20, seed=20241008, edge_frac=0.0), True, torch_cuda)


def test_box_edge_1m(ctx, oracle, torch_cuda):
    check_box(ctx, oracle, gen.box_ops(1 << 20, seed=11, edge_frac=0.5), True, torch_cuda)


def test_box_unaligned_views(ctx, oracle, torch_cuda):
    """Arrays that start 4 or 8 bytes into an allocation take the narrower paths."""
    torch = torch_cuda
    n = 5000
    fields = gen.box_ops(n, seed=5, edge_frac=0.2)
    wt, wf = oracle.box_soa(fields)
    for off in (1, 2, 3):
        d = []
        for f in fields:
            buf = torch.zeros(n + 8, dtype=torch.float32, device="cuda")
            buf[off:off + n] = torch.from_numpy(f).cuda()
            d.append(buf[off:off + n])
        tmin = [torch.empty(n + 8, dtype=torch.float32, device="cuda")[off:off + n] for _ in range(4)]
        flags = torch.empty(n + 8, dtype=torch.int32, device="cuda")[off:off + n]
        b = capi.Batch()
        b.set_box(d, tmin, flags, n)
        ctx.eval(b)
        for k in range(4):
            assert_bits_equal(tmin[k].cpu().numpy(), wt[k], "tmin_out[%d] off %d" % (k, off))
        assert_bits_equal(flags.cpu().numpy(), wf, "flags off %d" % off)


@pytest.mark.parametrize("n", [1, 2, 5, 256, 1027, 4098])
@pytest.mark.parametrize("device", [False, True])
def test_tri_ragged_sizes(ctx, oracle, torch_cuda, n, device):
    fields, k = gen.tri_ops(n, seed=200 + n, edge_frac=0.3)
    check_tri(ctx, oracle, fields, k, device, torch_cuda)


@pytest.mark.parametrize("v", [1, 2, 4])
@pytest.mark.parametrize("nt", [1, 2])
def test_tri_kernel_shapes(ctx, oracle, torch_cuda, v, nt):
    ctx.set_tuning(ops_per_lane=v, nontemporal=nt)
    fields, k = gen.tri_ops(200003, seed=8, edge_frac=0.25)
    check_tri(ctx, oracle, fields, k, True, torch_cuda)


def test_tri_random_and_edge_1m(ctx, oracle, torch_cuda):
    fields, k = gen.tri_ops(1 << 20, seed=20241009, edge_frac=0.0)
    check_tri(ctx, oracle, fields, k, True, torch_cuda)
    fields, k = gen.tri_ops(1 << 20, seed=12, edge_frac=0.5)
    check_tri(ctx, oracle, fields, k, True, torch_cuda)


def check_euclid(ctx, oracle, a, b, ctl, device, torch, acc0=0.0):
    ctx.set_accum([acc0, 0.0, 0.0])
    out = hip_euclid(ctx, a, b, ctl, device, torch)
    want, acc = oracle.euclid_beats(a, b, ctl, acc0)
    assert_bits_equal(out, want, "euclidean_accumulator")
    assert_bits_equal(ctx.get_accum()[0:1], np.array([acc], f32), "euclidean register after the batch")


def check_angular(ctx, oracle, q, c, ctl, device, torch, d0=0.0, n0=0.0):
    ctx.set_accum([0.0, d0, n0])
    dot, norm = hip_angular(ctx, q, c, ctl, device, torch)
    wd, wn, sd, sn = oracle.angular_beats(q, c, ctl, d0, n0)
    assert_bits_equal(dot, wd, "angular_dot_product")
    assert_bits_equal(norm, wn, "angular_norm")
    assert_bits_equal(ctx.get_accum()[1:3], np.array([sd, sn], f32), "angular registers after the batch")


@pytest.mark.parametrize("pairs", [1, 2, 17, 300, 5000])
@pytest.mark.parametrize("device", [False, True])
def test_euclid_pairs(ctx, oracle, torch_cuda, pairs, device):
    a, b, ctl, _ = gen.vec_beats(pairs, 16, seed=300 + pairs, edge_frac=0.1)
    check_euclid(ctx, oracle, a, b, ctl, device, torch_cuda)


@pytest.mark.parametrize("pairs", [1, 2, 17, 300, 5000])
@pytest.mark.parametrize("device", [False, True])
def test_angular_pairs(ctx, oracle, torch_cuda, pairs, device):
    q, c, ctl, _ = gen.vec_beats(pairs, 8, seed=400 + pairs, edge_frac=0.1)
    check_angular(ctx, oracle, q, c, ctl, device, torch_cuda)


@pytest.mark.parametrize("variant", [0, 23, 27, 21, 22, 28])
@pytest.mark.parametrize("tile,passes", [(64, 2), (240, 4), (256, 2), (512, 0), (512, 8), (2048, 4)])
def test_vec_tiles_and_long_pairs(ctx, oracle, torch_cuda, tile, passes, variant):
    """Pairs longer than a sub-tile and longer than a whole tile; batch ends mid-pair;
    batch starts with a carried-in register.  Variant 0: each opcode in its default shape; 27: fixed tiles that hand
    their running sums on, for both (the tile knob does not apply); 23 / 21 / 22: the shape in which a workgroup owns
    whole pairs, for both, with its two walks."""
    ctx.set_tuning(vec_tile=tile, vec_passes=passes, variant=variant)
    lengths = np.array([16 * 700, 5, 16 * 300 + 3, 1, 16 * 2100, 40, 16 * 256, 16 * 257, 9], np.int64)
    a, b, ctl, _ = gen.vec_beats(lengths.size, 16, seed=31, lengths=lengths)
    ctl = ctl.copy()
    ctl[-1] &= 0xFFFF  # the last pair continues into the next batch
    check_euclid(ctx, oracle, a, b, ctl, True, torch_cuda, acc0=123.5)
    lengths8 = (lengths // 2) + 1
    q, c, ctl8, _ = gen.vec_beats(lengths8.size, 8, seed=32, lengths=lengths8)
    ctl8 = ctl8.copy()
    ctl8[-1] &= 0xFFFF
    check_angular(ctx, oracle, q, c, ctl8, True, torch_cuda, d0=-3.25, n0=1e6)


def test_vec_no_reset_anywhere(ctx, oracle, torch_cuda):
    """One endless pair: every beat accumulates, no workgroup but the first has work."""
    a, b, ctl, _ = gen.vec_beats(1, 16, seed=33, lengths=np.array([16 * 3000]))
    ctl = (ctl & 0xFFFF).astype(u32)
    check_euclid(ctx, oracle, a, b, ctl, True, torch_cuda, acc0=1.0)


@pytest.mark.parametrize("width", [16, 8])
@pytest.mark.parametrize("pair_beats", [1 << 22, 1 << 18])
@pytest.mark.parametrize("variant", [23, 27, 28])
def test_vec_reset_poor_streams_are_not_a_cliff(ctx, oracle, torch_cuda, width, pair_beats, variant):
    """2^22 beats with no reset at all (one endless pair) and with a reset every 2^18 beats: the workgroups whose
    nominal range holds no pair start must leave after looking at their own tile only (the accumulation itself is
    serial by definition and belongs to the workgroup in whose range the pair starts).  Bit-exact against the
    oracle, and the launch stays under one second of wall time (measured on MI355X: tens of milliseconds)."""
    import time
    torch = torch_cuda
    nb = 1 << 22
    rng = np.random.default_rng(width + pair_beats)
    a = (rng.random((nb, width), dtype=f32) * f32(200) - f32(100)).astype(f32)
    b = (rng.random((nb, width), dtype=f32) * f32(200) - f32(100)).astype(f32)
    ctl = np.full(nb, (1 << width) - 1, u32)
    if pair_beats < nb:
        ctl[pair_beats - 1::pair_beats] |= u32(1 << 16)
        ctl[-1] &= u32(0xFFFF)                      # the last pair runs on into the next batch
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    dc = torch.from_numpy(ctl.view(np.int32)).cuda()
    out0 = torch.empty(nb, dtype=torch.float32, device="cuda")
    out1 = torch.empty(nb, dtype=torch.float32, device="cuda")
    bt = capi.Batch()
    if width == 16:
        bt.set_euclid(da, db, dc, out0, nb)
    else:
        bt.set_angular(da, db, dc, out0, out1, nb)
    ctx.set_option("variant", variant)
    ctx.set_accum([0.5, -2.0, 3.0])
    ctx.eval(bt)                                     # first call: code object load and the like
    ctx.set_accum([0.5, -2.0, 3.0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.eval(bt)
    wall = time.perf_counter() - t0
    if width == 16:
        want, acc = oracle.euclid_beats(a, b, ctl, 0.5)
        assert_bits_equal(out0.cpu().numpy(), want, "euclidean_accumulator")
        assert_bits_equal(ctx.get_accum()[0:1], np.array([acc], f32), "euclidean register after the batch")
    else:
        wd, wn, sd, sn = oracle.angular_beats(a, b, ctl, -2.0, 3.0)
        assert_bits_equal(out0.cpu().numpy(), wd, "angular_dot_product")
        assert_bits_equal(out1.cpu().numpy(), wn, "angular_norm")
        assert_bits_equal(ctx.get_accum()[1:3], np.array([sd, sn], f32), "angular registers after the batch")
    assert wall < 1.0, "reset-poor stream took %.3f s" % wall
    assert ctx.get_option("vec_chain_giveups") == 0
    print("reset-poor stream, width %d, pair of %d beats, variant %d: %.1f ms" % (width, pair_beats, variant, wall * 1e3))


def test_accumulators_run_across_batches(ctx, oracle, torch_cuda):
    """Splitting a stream into several calls gives the same results as one call."""
    a, b, ctl, _ = gen.vec_beats(400, 16, seed=34)
    want, acc = oracle.euclid_beats(a, b, ctl, 0.0)
    ctx.reset()
    cuts = [0, 1, 7, 300, 301, 900, ctl.shape[0]]
    got = np.concatenate([hip_euclid(ctx, a[s:e], b[s:e], ctl[s:e], True, torch_cuda) for s, e in zip(cuts[:-1], cuts[1:])])
    assert_bits_equal(got, want, "euclidean_accumulator over split batches")
    assert_bits_equal(ctx.get_accum()[0:1], np.array([acc], f32), "register")


def test_baseline_datapath_ignores_distance_ops(oracle, torch_cuda):
    """RaytracerParams(support_euclidean = None): opcodes 2/3 produce zeros."""
    c = capi.Context(support_euclidean=0)
    a, b, ctl, _ = gen.vec_beats(50, 16, seed=35)
    assert not hip_euclid(c, a, b, ctl, True, torch_cuda).any()
    q, cc, ctl8, _ = gen.vec_beats(50, 8, seed=36)
    d, n = hip_angular(c, q, cc, ctl8, False)
    assert not d.any() and not n.any()
    check_box(c, oracle, gen.box_ops(1000, seed=37), True, torch_cuda)
    c.close()


def test_mixed_bundle_stream_matches_rtl_oracle(ctx, oracle, torch_cuda):
    """An unsorted stream of all four opcodes through rt_dp_eval_bundles against
    oracle A (the gate-level restatement), including the accumulators interleaved
    with other opcodes and the echoed reset bits."""
    rng = np.random.default_rng(77)
    wb = gen.aos_from_box(gen.box_ops(3000, seed=41, edge_frac=0.3))
    wt = gen.aos_from_tri(*gen.tri_ops(3000, seed=42, edge_frac=0.3))
    a, b, ctl, _ = gen.vec_beats(700, 16, seed=43, edge_frac=0.1)
    we = gen.aos_from_vec(a, b, ctl, 2)
    q, c, ctl8, _ = gen.vec_beats(400, 8, seed=44, edge_frac=0.1)
    wa = gen.aos_from_vec(q, c, ctl8, 3)
    # interleave while keeping each opcode's internal order (the accumulators need it)
    tags = np.concatenate([np.full(w.shape[0], i) for i, w in enumerate((wb, wt, we, wa))])
    rng.shuffle(tags)
    srcs, pos = (wb, wt, we, wa), [0, 0, 0, 0]
    stream = np.zeros((tags.size, 84), u32)
    for i, t in enumerate(tags):
        stream[i] = srcs[t][pos[t]]
        pos[t] += 1
    # box / triangle records of the extended test carry reset = 1, mask = 0 (Datapath_test.scala:399-409)
    stream[stream[:, 0] < 2, 83] = 1
    ctx.reset()
    got = ctx.eval_bundles(stream)
    want, state = oracle.eval_bundles("A", stream.view(oracle.IN_BUNDLE).reshape(-1), extended=True)
    assert_bits_equal(got, want.view(u32).reshape(-1, 21), "output bundles")
    acc = ctx.get_accum().view(u32)
    assert tuple(int(x) for x in acc) == state


def test_error_codes(ctx, torch_cuda):
    b = capi.Batch()
    b.ob.n[capi.OP_QUADBOX] = 10  # arrays missing
    with pytest.raises(capi.RtDpError) as e:
        ctx.eval(b)
    assert e.value.code == capi.ENULL
    b = capi.Batch()
    b.ob.struct_size = 12
    with pytest.raises(capi.RtDpError) as e:
        ctx.eval(b)
    assert e.value.code == capi.EINVAL
    bad = np.zeros((3, 84), u32)
    bad[1, 0] = 4
    with pytest.raises(capi.RtDpError) as e:
        ctx.eval_bundles(bad)
    assert e.value.code == capi.EOPCODE
    with pytest.raises(capi.RtDpError) as e:
        capi.Context(support_euclidean=8)
    assert e.value.code == capi.EINVAL and "16-element" in str(e.value)
    with pytest.raises(capi.RtDpError) as e:
        capi.Context(io_recorded_float=1)
    assert "recorded-format" in str(e.value)
    ctx.eval(capi.Batch())  # empty batch is fine


def test_ray_prepare_follows_sw_ray(ctx, torch_cuda):
    """The pre-processing kernel against the float32 restatement of SW_Ray (gen.ray_derived),
    including zero, subnormal, huge and non-finite direction components and |dir| ties."""
    torch = torch_cuda
    rng = np.random.default_rng(50)
    n = 200000
    d = rng.uniform(-100, 100, (n, 3)).astype(f32)
    sp = np.array([0.0, -0.0, 1e-42, -1e-42, 3e38, -3e38, np.inf, -np.inf, 1.0, -1.0, 2.0 ** -126, 1e-20], f32)
    rows = rng.choice(n, n // 10, replace=False)
    d[rows, rng.integers(0, 3, rows.size)] = sp[rng.integers(0, sp.size, rows.size)]
    ties = rng.choice(n, 2000, replace=False)
    d[ties, 1] = d[ties, 0] * rng.choice(np.array([1, -1], f32), ties.size)   # |x| == |y|
    d[ties[:500], 2] = d[ties[:500], 0]
    inv, kx, ky, kz, shear = gen.ray_derived(d)
    D = [torch.from_numpy(np.ascontiguousarray(d[:, c])).cuda() for c in range(3)]
    oi = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    os_ = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    ok = torch.empty(n, dtype=torch.int32, device="cuda")
    oe = torch.empty(n, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.ray_prepare(D, inv=oi, k=ok, shear=os_, extent=oe)
    ctx.synchronize()

    def canon(x):
        b = np.ascontiguousarray(x, f32).view(u32).copy()
        b[np.isnan(x)] = 0x7FC00000
        return b
    for c in range(3):
        assert (oi[c].cpu().numpy().view(u32) == canon(inv[:, c])).all(), "inv[%d]" % c
    finite = np.isfinite(d).all(1)   # SW_Ray's comparisons with a NaN direction are not exercised by the reference
    assert (ok.cpu().numpy().view(u32)[finite] == gen.pack_k(kx, ky, kz)[finite]).all()
    for c in range(3):
        assert (os_[c].cpu().numpy().view(u32)[finite] == canon(shear[:, c])[finite]).all(), "shear[%d]" % c
    assert (oe.cpu().numpy() == 4000.0).all()


def _mixed_stream(seed, nb, nt, pe, pa, edge=0.3):
    rng = np.random.default_rng(seed)
    wb = gen.aos_from_box(gen.box_ops(nb, seed=seed + 1, edge_frac=edge))
    wt = gen.aos_from_tri(*gen.tri_ops(nt, seed=seed + 2, edge_frac=edge))
    a, b, ctl, _ = gen.vec_beats(pe, 16, seed=seed + 3, edge_frac=0.1)
    we = gen.aos_from_vec(a, b, ctl, 2)
    q, c, ctl8, _ = gen.vec_beats(pa, 8, seed=seed + 4, edge_frac=0.1)
    wa = gen.aos_from_vec(q, c, ctl8, 3)
    srcs = (wb, wt, we, wa)
    tags = np.concatenate([np.full(w.shape[0], i) for i, w in enumerate(srcs)])
    rng.shuffle(tags)
    stream = np.zeros((tags.size, 84), u32)
    for i in range(4):  # each opcode keeps its internal order
        stream[tags == i] = srcs[i]
    stream[stream[:, 0] < 2, 83] = 1
    return stream


@pytest.mark.parametrize("sizes", [(1, 0, 0, 0), (0, 0, 1, 0), (3, 2, 1, 1), (0, 0, 0, 30), (0, 0, 50, 0), (300, 300, 40, 40),
                                   (20000, 20000, 3000, 3000), (60000, 10, 2, 900)])
@pytest.mark.parametrize("extended", [True, False])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_device_record_stream_matches_rtl_oracle(oracle, torch_cuda, sizes, extended, mode):
    """rt_dp_eval_bundles_device (1: partition by opcode first, 2: evaluate in place then fold, 3: single pass) against oracle A."""
    torch = torch_cuda
    stream = _mixed_stream(60 + sizes[0], *sizes)
    c = capi.Context(support_euclidean=16 if extended else 0)
    c.set_option("stream_mode", mode)
    d_in = torch.from_numpy(stream.view(np.int32)).cuda()
    d_out = torch.full((stream.shape[0], 21), -1, dtype=torch.int32, device="cuda")
    c.eval_bundles_device(d_in, d_out)
    want, state = oracle.eval_bundles("A" if stream.shape[0] < 5000 else "B", stream.view(oracle.IN_BUNDLE).reshape(-1), extended=extended)
    assert_bits_equal(d_out.cpu().numpy().view(u32), want.view(u32).reshape(-1, 21), "output bundles")
    if extended:
        assert tuple(int(x) for x in c.get_accum().view(u32)) == state
    # the host-partitioned entry point agrees
    c.reset()
    assert_bits_equal(c.eval_bundles(stream), want.view(u32).reshape(-1, 21), "host-partitioned output bundles")
    bad = stream.copy()
    bad[0, 0] = 7
    with pytest.raises(capi.RtDpError) as e:
        c.eval_bundles_device(torch.from_numpy(bad.view(np.int32)).cuda(), d_out)
    assert e.value.code == capi.EOPCODE
    c.close()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_device_record_stream_carries_accumulators(oracle, torch_cuda, mode):
    """A stream cut at arbitrary records (inside vector pairs) and fed in pieces equals the stream fed whole; a stream
    of distance beats without any reset is one long pair."""
    torch = torch_cuda
    stream = _mixed_stream(7, 5000, 5000, 700, 700)
    want, state = oracle.eval_bundles("B", stream.view(oracle.IN_BUNDLE).reshape(-1), extended=True)
    c = capi.Context()
    c.set_option("stream_mode", mode)
    got = []
    cuts = [0, 1, 64, 1000, 1027, 4096, 9999, stream.shape[0]]
    for a, b in zip(cuts[:-1], cuts[1:]):
        d_in = torch.from_numpy(np.ascontiguousarray(stream[a:b]).view(np.int32)).cuda()
        d_out = torch.empty((b - a, 21), dtype=torch.int32, device="cuda")
        c.eval_bundles_device(d_in, d_out)
        got.append(d_out.cpu().numpy().view(u32))
    assert_bits_equal(np.concatenate(got), want.view(u32).reshape(-1, 21), "pieces")
    assert tuple(int(x) for x in c.get_accum().view(u32)) == state
    c.reset()
    endless = stream[stream[:, 0] >= 2].copy()
    endless[:, 83] = 0
    endless = np.ascontiguousarray(np.concatenate([endless] * 3))      # > 2048 beats per opcode, never a reset
    want, state = oracle.eval_bundles("B", endless.view(oracle.IN_BUNDLE).reshape(-1), extended=True)
    d_in = torch.from_numpy(endless.view(np.int32)).cuda()
    d_out = torch.empty((endless.shape[0], 21), dtype=torch.int32, device="cuda")
    c.eval_bundles_device(d_in, d_out)
    assert_bits_equal(d_out.cpu().numpy().view(u32), want.view(u32).reshape(-1, 21), "endless pair")
    assert tuple(int(x) for x in c.get_accum().view(u32)) == state
    c.close()


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_record_stream_into_slices_of_one_output_array(oracle, torch_cuda, mode):
    """Pieces of a stream evaluated straight into their places in ONE output array: a piece that starts at record k starts
    84 k bytes into it, which is 16-byte aligned for every fourth k only; the in-place paths take any."""
    torch = torch_cuda
    stream = _mixed_stream(17, 4000, 3000, 500, 600)
    n = stream.shape[0]
    want, state = oracle.eval_bundles("B", stream.view(oracle.IN_BUNDLE).reshape(-1), extended=True)
    d_in = torch.from_numpy(stream.view(np.int32)).cuda()
    d_out = torch.full((n + 3, 21), -1, dtype=torch.int32, device="cuda")
    c = capi.Context()
    c.set_option("stream_mode", mode)
    cuts = [0, 1, 2, 3, 67, 130, 1001, 4098, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert (d_in[a:b].data_ptr() & 15) == 0 and ((d_out[a:b].data_ptr() & 15) == 0) == (a % 4 == 0)
        c.eval_bundles_device(d_in[a:b], d_out[a:b])
    got = d_out.cpu().numpy().view(u32)
    assert_bits_equal(got[:n], want.view(u32).reshape(-1, 21), "records written into slices")
    assert (got[n:] == 0xFFFFFFFF).all(), "records behind the last slice were touched"
    assert tuple(int(x) for x in c.get_accum().view(u32)) == state
    c.close()


def _single_pass_case(oracle, torch, stream, what, chunk_log2=None, monkeypatch=None, mode=3):
    """One stream through the single-pass record-stream kernel against oracle B, registers included, twice in a row on one
    context (the second call continues the first one's registers)."""
    if chunk_log2 is not None:
        monkeypatch.setenv("RT_DP_STREAM_CHUNK_LOG2", str(chunk_log2))
    c = capi.Context()
    c.set_option("stream_mode", mode)
    two = np.ascontiguousarray(np.concatenate([stream, stream]))
    want, state = oracle.eval_bundles("B", two.view(oracle.IN_BUNDLE).reshape(-1), extended=True)
    want = want.view(u32).reshape(-1, 21)
    d_in = torch.from_numpy(stream.view(np.int32)).cuda()
    n = stream.shape[0]
    for half in range(2):
        d_out = torch.full((n, 21), -1, dtype=torch.int32, device="cuda")
        c.eval_bundles_device(d_in, d_out)
        assert_bits_equal(d_out.cpu().numpy().view(u32), want[half * n:(half + 1) * n], "%s, call %d" % (what, half))
    assert tuple(int(x) for x in c.get_accum().view(u32)) == state, what
    assert c.get_option("stream_giveups") == 0
    c.close()


@pytest.mark.parametrize("mode", [2, 3])
def test_record_stream_long_stretches_without_an_opcode(oracle, torch_cuda, mode):
    """The running sum of a vector pair has to cross thousands of records that hold no beat of its opcode: the look-back
    (mode 3) and the fold (mode 2) skip the waves in between 64 at a time."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    a, b, ctl, _ = gen.vec_beats(6, 16, seed=51, lengths=[40, 100, 16, 128, 7, 90])
    q, cnd, ctl8, _ = gen.vec_beats(5, 8, seed=52, lengths=[30, 64, 8, 100, 50])
    we, wa = gen.aos_from_vec(a, b, ctl, 2), gen.aos_from_vec(q, cnd, ctl8, 3)
    box = gen.aos_from_box(gen.box_ops(40000, seed=53, edge_frac=0.1))
    tri = gen.aos_from_tri(*gen.tri_ops(9000, seed=54, edge_frac=0.1))
    # beats of both distance opcodes sprinkled singly into long runs of box / triangle records: pairs stay open across the runs
    parts, eb, ab = [], list(we), list(wa)
    pos = 0
    while eb or ab:
        run = int(rng.integers(1, 9000))
        parts.append(box[pos % 30000:pos % 30000 + run])
        pos += run
        if eb and (not ab or rng.random() < 0.5):
            parts.append(eb.pop(0)[None, :])
        elif ab:
            parts.append(ab.pop(0)[None, :])
        parts.append(tri[:int(rng.integers(0, 200))])
    stream = np.ascontiguousarray(np.concatenate(parts)).astype(u32)
    stream[stream[:, 0] < 2, 83] = 1
    assert stream.shape[0] > 100000
    _single_pass_case(oracle, torch, stream, "sparse distance beats", mode=mode)


@pytest.mark.parametrize("mode", [2, 3])
def test_record_stream_pairs_open_across_half_a_million_records(oracle, torch_cuda, mode):
    """Distance beats at both ends of a stream and nowhere in between: the fold's bit words are crossed 64 words (2^18 records)
    at a time, in both directions; one opcode's pair stays open across the gap, the other's is closed in front of it."""
    torch = torch_cuda
    a, b, ctl, _ = gen.vec_beats(3, 16, seed=61, lengths=[70, 9, 75])
    q, cnd, ctl8, _ = gen.vec_beats(3, 8, seed=62, lengths=[66, 5, 80])
    we, wa = gen.aos_from_vec(a, b, ctl, 2), gen.aos_from_vec(q, cnd, ctl8, 3)
    box = gen.aos_from_box(gen.box_ops(1 << 15, seed=63, edge_frac=0.1))
    gap = np.concatenate([box] * 17)[:(1 << 19) + 777]
    # front: the first Euclidean pair and 4 beats of the second, the first angular pair complete; back: everything else
    stream = np.ascontiguousarray(np.concatenate([we[:74], wa[:66], gap, wa[66:], we[74:], gap[:5000], wa[:3]])).astype(u32)
    stream[stream[:, 0] < 2, 83] = 1
    _single_pass_case(oracle, torch, stream, "beats at both ends only", mode=mode)


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("kind", ["box_only", "euclid_only", "angular_only", "no_reset", "reset_everywhere", "tail_of_7", "few_resets"])
def test_record_stream_corner_streams(oracle, torch_cuda, kind, mode):
    torch = torch_cuda
    base = _mixed_stream(90, 3000, 3000, 500, 500)
    if kind == "box_only":
        stream = base[base[:, 0] == 1]
    elif kind == "euclid_only":
        stream = base[base[:, 0] == 2]
    elif kind == "angular_only":
        stream = base[base[:, 0] == 3]
    elif kind == "no_reset":
        stream = base.copy()
        stream[stream[:, 0] >= 2, 83] = 0
    elif kind == "reset_everywhere":
        stream = base.copy()
        stream[:, 83] = 1
    elif kind == "few_resets":  # pairs that run through dozens of waves, some waves with more than four head beats
        stream = np.concatenate([base, base, base])
        keep = np.random.default_rng(93).random(stream.shape[0]) < 0.02
        stream[(stream[:, 0] >= 2) & ~keep, 83] = 0
    else:
        stream = base[:7]
    _single_pass_case(oracle, torch, np.ascontiguousarray(stream), kind, mode=mode)


@pytest.mark.parametrize("mode", [2, 3])
def test_record_stream_in_pieces(oracle, torch_cuda, monkeypatch, mode):
    """Pieces of 2^11 records inside one call: the registers travel from piece to piece on the device."""
    stream = _mixed_stream(91, 4000, 4000, 600, 600)
    _single_pass_case(oracle, torch_cuda, stream, "2^11-record pieces", chunk_log2=11, monkeypatch=monkeypatch, mode=mode)


def test_record_stream_pieces_no_reset_and_bad_opcode(oracle, torch_cuda, monkeypatch):
    """Pieces of one call on a stream whose pairs never end (every piece depends on the one before), and on a stream with an
    opcode > 3 in a late piece: the error is reported and the registers keep their values."""
    torch = torch_cuda
    base = _mixed_stream(92, 3000, 3000, 700, 700)
    endless = base.copy()
    endless[endless[:, 0] >= 2, 83] = 0
    _single_pass_case(oracle, torch, np.ascontiguousarray(endless), "endless pairs in 2^11-record pieces", chunk_log2=11, monkeypatch=monkeypatch, mode=2)
    c = capi.Context()
    c.set_option("stream_mode", 2)
    c.set_accum([1.5, -2.5, 3.5])
    bad = base.copy()
    bad[-5, 0] = 6
    d_out = torch.empty((bad.shape[0], 21), dtype=torch.int32, device="cuda")
    with pytest.raises(capi.RtDpError) as e:
        c.eval_bundles_device(torch.from_numpy(bad.view(np.int32)).cuda(), d_out)
    assert e.value.code == capi.EOPCODE
    assert_bits_equal(c.get_accum(), np.array([1.5, -2.5, 3.5], f32), "registers after a refused stream")
    c.close()
