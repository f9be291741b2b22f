"""Tiled box / triangle segments (RT_DP_F_BOX_TILED, RT_DP_F_TRI_TILED): one buffer [tile][word][8192 ops] per
segment instead of one array per operand.  The layout is restated here in numpy (tile_index) and pins
rt_dp_tile_pack / rt_dp_tile_unpack; the tiled kernels must give the oracle's bits, i.e. the same bits as the plain
kernels, at every size around the tile and workgroup boundaries, with and without RT_DP_F_DERIVE_RAY, alone and next
to the other opcodes of a batch, and must leave what lies behind the last op of the last tile alone."""
import ctypes as C

import numpy as np
import pytest

from oracle import gen
from raytracer_amd import capi

from helpers import assert_bits_equal, hip_box, hip_tri

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32
T = capi.TILE_OPS
SENTINEL = 0x5EED5EED
SIZES = [1, 2, 255, 256, 257, T - 1, T, T + 1, 3 * T, 100003]


def tile_index(i, word, words):
    """RT_DP_TILED_INDEX of include/rt_dp.h."""
    i = np.asarray(i, np.int64)
    return (i // T) * T * words + word * T + i % T


def pack_numpy(arrays):
    n, words = arrays[0].shape[0], len(arrays)
    out = np.full(capi.tiled_words(n, words), SENTINEL, u32)
    idx = np.arange(n)
    for k, a in enumerate(arrays):
        out[tile_index(idx, k, words)] = np.ascontiguousarray(a).view(u32)
    return out


def unpack_numpy(tiled, words, n):
    idx = np.arange(n)
    return [tiled[tile_index(idx, k, words)] for k in range(words)]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("words", [1, 5, 34, 40])
def test_pack_and_unpack_follow_the_layout(ctx, torch_cuda, n, words):
    torch = torch_cuda
    rng = np.random.default_rng(n * 41 + words)
    arrays = [rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(u32) for _ in range(words)]
    dev = [torch.from_numpy(a.view(np.int32)).cuda() for a in arrays]
    tiled = torch.full((capi.tiled_words(n, words),), SENTINEL, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.tile_pack(dev, tiled)
    ctx.synchronize()
    assert_bits_equal(tiled.cpu().numpy().view(u32), pack_numpy(arrays), "packed buffer (padding untouched)")
    back = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(words)]
    torch.cuda.synchronize()
    ctx.tile_unpack(tiled, back)
    ctx.synchronize()
    for k in range(words):
        assert_bits_equal(back[k].cpu().numpy().view(u32), arrays[k], "array %d after the round trip" % k)


def run_box_tiled(ctx, torch, fields, derive=False, stream=None):
    n = fields[0].shape[0]
    src = [np.zeros(n, f32) if f is None else f for f in fields]
    operands = torch.from_numpy(pack_numpy(src).view(np.int32)).cuda()
    results = torch.full((capi.tiled_words(n, 5),), SENTINEL, dtype=torch.int32, device="cuda")
    b = capi.Batch().derive_ray(derive)
    b.set_box_tiled(operands, results, n)
    torch.cuda.synchronize()
    ctx.eval(b)
    raw = results.cpu().numpy().view(u32)
    out = unpack_numpy(raw, 5, n)
    # everything outside the n results is as it was
    mask = np.ones(raw.shape[0], bool)
    for k in range(5):
        mask[tile_index(np.arange(n), k, 5)] = False
    assert (raw[mask] == SENTINEL).all(), "the box kernel wrote behind op n - 1"
    return [o.view(f32) for o in out[:4]], out[4]


def run_tri_tiled(ctx, torch, fields, k, derive=False):
    n = fields[0].shape[0]
    kk = np.zeros(n, u32) if k is None else k
    operands = torch.from_numpy(pack_numpy(list(fields) + [kk]).view(np.int32)).cuda()
    results = torch.full((capi.tiled_words(n, 3),), SENTINEL, dtype=torch.int32, device="cuda")
    b = capi.Batch().derive_ray(derive)
    b.set_tri_tiled(operands, results, n)
    torch.cuda.synchronize()
    ctx.eval(b)
    raw = results.cpu().numpy().view(u32)
    out = unpack_numpy(raw, 3, n)
    mask = np.ones(raw.shape[0], bool)
    for w in range(3):
        mask[tile_index(np.arange(n), w, 3)] = False
    assert (raw[mask] == SENTINEL).all(), "the triangle kernel wrote behind op n - 1"
    return out[0].view(f32), out[1].view(f32), out[2]


@pytest.mark.parametrize("n", SIZES)
def test_tiled_box_ops_match_the_oracle(ctx, oracle, torch_cuda, n):
    fields = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=900 + n, edge_frac=0.3)]
    got_t, got_f = run_box_tiled(ctx, torch_cuda, fields)
    want_t, want_f = oracle.box_soa(fields)
    for k in range(4):
        assert_bits_equal(got_t[k], want_t[k], "tmin_out[%d]" % k)
    assert_bits_equal(got_f, want_f, "flags")


@pytest.mark.parametrize("n", SIZES)
def test_tiled_triangle_ops_match_the_oracle(ctx, oracle, torch_cuda, n):
    fields, k = gen.tri_ops(n, seed=950 + n, edge_frac=0.3)
    fields = [np.ascontiguousarray(f) for f in fields]
    gn, gd, gh = run_tri_tiled(ctx, torch_cuda, fields, k)
    wn, wd, wh = oracle.tri_soa(fields, k)
    assert_bits_equal(gn, wn, "t_num")
    assert_bits_equal(gd, wd, "t_denom")
    assert_bits_equal(gh, wh, "hit")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_aligned_shards_of_a_tiled_segment(ctx, oracle, torch_cuda, world):
    """shard.tile_cuts: every rank's part of a tiled segment is a slice of the buffer (no re-packing) and a valid tiled segment of
    its own; the parts' results, put together in rank order, are the oracle's results for the whole segment."""
    from raytracer_amd import shard
    torch = torch_cuda
    n = 5 * T + 77
    fields = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=4711, edge_frac=0.2)]
    want_t, want_f = oracle.box_soa(fields)
    operands = torch.from_numpy(pack_numpy(fields).view(np.int32)).cuda()
    results = torch.full((capi.tiled_words(n, 5),), SENTINEL, dtype=torch.int32, device="cuda")
    cuts = shard.tile_cuts(n, world)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b == a:
            continue
        batch = capi.Batch()
        batch.set_box_tiled(shard.tiled_slice(operands, 34, a, b), shard.tiled_slice(results, 5, a, b), b - a)
        torch.cuda.synchronize()
        ctx.eval(batch)
    raw = results.cpu().numpy().view(u32)
    out = unpack_numpy(raw, 5, n)
    for k in range(4):
        assert_bits_equal(out[k].view(f32), want_t[k], "tmin_out[%d]" % k)
    assert_bits_equal(out[4], want_f, "flags")


@pytest.mark.parametrize("n", [1, 257, T + 1, 50021])
def test_tiled_segments_with_derived_ray_fields(ctx, torch_cuda, n):
    """RT_DP_F_DERIVE_RAY over tiled segments: the bits of the plain derive path (which the derive tests pin against
    the two-step path and the oracle)."""
    torch = torch_cuda
    from test_derive import directions
    fields = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=300 + n, edge_frac=0.3)]
    d = directions(n, 301 + n)
    for c in range(3):
        fields[3 + c] = np.ascontiguousarray(d[:, c])
        fields[6 + c] = None
    b = capi.Batch().derive_ray()
    dev = [None if f is None else torch.from_numpy(f).cuda() for f in fields]
    tmin = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]
    flags = torch.empty(n, dtype=torch.int32, device="cuda")
    b.set_box(dev, tmin, flags, n)
    torch.cuda.synchronize()
    ctx.eval(b)
    got_t, got_f = run_box_tiled(ctx, torch, fields, derive=True)
    for k in range(4):
        assert_bits_equal(got_t[k], tmin[k].cpu().numpy(), "tmin_out[%d] against the plain derive path" % k)
    assert_bits_equal(got_f, flags.cpu().numpy().view(u32), "flags against the plain derive path")

    tf, _ = gen.tri_ops(n, seed=400 + n, edge_frac=0.3)
    tf = [np.ascontiguousarray(f) for f in tf]
    for c in range(3):
        tf[3 + c] = np.ascontiguousarray(d[:, c])
    b = capi.Batch().derive_ray()
    tn, td = torch.empty(n, dtype=torch.float32, device="cuda"), torch.empty(n, dtype=torch.float32, device="cuda")
    hit = torch.empty(n, dtype=torch.int32, device="cuda")
    b.set_tri([torch.from_numpy(f).cuda() for f in tf], None, tn, td, hit, n)
    torch.cuda.synchronize()
    ctx.eval(b)
    gn, gd, gh = run_tri_tiled(ctx, torch, tf, None, derive=True)
    assert_bits_equal(gn, tn.cpu().numpy(), "t_num against the plain derive path")
    assert_bits_equal(gd, td.cpu().numpy(), "t_denom against the plain derive path")
    assert_bits_equal(gh, hit.cpu().numpy().view(u32), "hit against the plain derive path")


@pytest.mark.parametrize("which", ["box", "tri", "both"])
def test_tiled_segments_next_to_the_other_opcodes_of_a_batch(ctx, oracle, torch_cuda, which):
    torch = torch_cuda
    nb, nt = 20011, 9001
    bf = [np.ascontiguousarray(f) for f in gen.box_ops(nb, seed=31, edge_frac=0.2)]
    tf, tk = gen.tri_ops(nt, seed=32, edge_frac=0.2)
    tf = [np.ascontiguousarray(f) for f in tf]
    a, bb, ctl, _ = gen.vec_beats(700, 16, seed=33)
    qa, qb, qctl, _ = gen.vec_beats(700, 8, seed=34)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    batch = capi.Batch()
    box_res = tri_res = None
    if which in ("box", "both"):
        box_res = torch.zeros(capi.tiled_words(nb, 5), dtype=torch.int32, device="cuda")
        batch.set_box_tiled(dev(pack_numpy(bf).view(np.int32)), box_res, nb)
    else:
        tmin = [torch.empty(nb, dtype=torch.float32, device="cuda") for _ in range(4)]
        flags = torch.empty(nb, dtype=torch.int32, device="cuda")
        batch.set_box([dev(f) for f in bf], tmin, flags, nb)
    if which in ("tri", "both"):
        tri_res = torch.zeros(capi.tiled_words(nt, 3), dtype=torch.int32, device="cuda")
        batch.set_tri_tiled(dev(pack_numpy(tf + [tk]).view(np.int32)), tri_res, nt)
    else:
        tn, td = torch.empty(nt, dtype=torch.float32, device="cuda"), torch.empty(nt, dtype=torch.float32, device="cuda")
        hit = torch.empty(nt, dtype=torch.int32, device="cuda")
        batch.set_tri([dev(f) for f in tf], dev(tk.view(np.int32)), tn, td, hit, nt)
    acc = torch.empty(ctl.shape[0], dtype=torch.float32, device="cuda")
    dot, norm = torch.empty(qctl.shape[0], dtype=torch.float32, device="cuda"), torch.empty(qctl.shape[0], dtype=torch.float32, device="cuda")
    batch.set_euclid(dev(a), dev(bb), dev(ctl.view(np.int32)), acc)
    batch.set_angular(dev(qa), dev(qb), dev(qctl.view(np.int32)), dot, norm)
    ctx.reset()
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx.launch(batch, stream.cuda_stream)
    stream.synchronize()
    want_t, want_f = oracle.box_soa(bf)
    if box_res is not None:
        out = unpack_numpy(box_res.cpu().numpy().view(u32), 5, nb)
        got_t, got_f = [o.view(f32) for o in out[:4]], out[4]
    else:
        got_t, got_f = [t.cpu().numpy() for t in tmin], flags.cpu().numpy().view(u32)
    for k in range(4):
        assert_bits_equal(got_t[k], want_t[k], "tmin_out[%d]" % k)
    assert_bits_equal(got_f, want_f, "flags")
    wn, wd, wh = oracle.tri_soa(tf, tk)
    if tri_res is not None:
        out = unpack_numpy(tri_res.cpu().numpy().view(u32), 3, nt)
        gn, gd, gh = out[0].view(f32), out[1].view(f32), out[2]
    else:
        gn, gd, gh = tn.cpu().numpy(), td.cpu().numpy(), hit.cpu().numpy().view(u32)
    assert_bits_equal(gn, wn, "t_num")
    assert_bits_equal(gd, wd, "t_denom")
    assert_bits_equal(gh, wh, "hit")
    assert_bits_equal(acc.cpu().numpy(), oracle.euclid_beats(a, bb, ctl)[0], "euclidean beats")
    wdot, wnorm, _, _ = oracle.angular_beats(qa, qb, qctl)
    assert_bits_equal(dot.cpu().numpy(), wdot, "angular dot")
    assert_bits_equal(norm.cpu().numpy(), wnorm, "angular norm")


def test_tiled_results_unpack_into_what_the_plain_kernels_write(ctx, torch_cuda):
    """pack -> tiled kernel -> unpack on the device, against the plain kernel on the same arrays, 2^20 + 3 ops (the
    size from which the residency cap applies)."""
    torch = torch_cuda
    n = (1 << 20) + 3
    fields = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=77, edge_frac=0.05)]
    want_t, want_f = hip_box(ctx, fields, True, torch)
    dev = [torch.from_numpy(f).cuda() for f in fields]
    operands = torch.empty(capi.tiled_words(n, 34), dtype=torch.float32, device="cuda")
    results = torch.empty(capi.tiled_words(n, 5), dtype=torch.int32, device="cuda")
    out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(5)]
    b = capi.Batch()
    b.set_box_tiled(operands, results, n)
    torch.cuda.synchronize()
    ctx.tile_pack(dev, operands)          # all three on the context's stream, in order
    ctx.launch(b)
    ctx.tile_unpack(results, out)
    ctx.synchronize()
    for k in range(4):
        assert_bits_equal(out[k].cpu().numpy().view(u32), want_t[k], "tmin_out[%d]" % k)
    assert_bits_equal(out[4].cpu().numpy().view(u32), want_f, "flags")


def test_tiled_argument_errors(ctx, torch_cuda):
    torch = torch_cuda
    L = capi.load()
    n = 100
    operands = torch.zeros(capi.tiled_words(n, 34), dtype=torch.float32, device="cuda")
    results = torch.zeros(capi.tiled_words(n, 5), dtype=torch.int32, device="cuda")
    b = capi.Batch()
    b.set_box_tiled(operands, results, n)
    b.finalize()
    b.rb.box.tmin_out[0] = None
    assert L.rt_dp_eval_ctx(ctx._h, C.byref(b.ob), C.byref(b.rb)) == capi.ENULL
    ptrs = (C.c_void_p * 41)(*([operands.data_ptr()] * 41))
    assert L.rt_dp_tile_pack(ctx._h, ptrs, 41, n, results.data_ptr(), None) == capi.EINVAL
    assert L.rt_dp_tile_pack(ctx._h, ptrs, 0, n, results.data_ptr(), None) == capi.EINVAL
    ptrs[1] = None
    assert L.rt_dp_tile_pack(ctx._h, ptrs, 2, n, results.data_ptr(), None) == capi.ENULL
    assert L.rt_dp_tile_pack(ctx._h, ptrs, 2, 0, None, None) == capi.OK        # nothing to do


@pytest.mark.parametrize("n", [1, 257, T - 1, T + 1, 70001])
@pytest.mark.parametrize("tables", [False, True])
def test_compaction_over_a_tiled_result_buffer(ctx, oracle, torch_cuda, n, tables):
    """rt_dp_box_compact_tiled: the visit list read off the oracle's results (ops in order, hits closest first), from
    the buffer the tiled box kernel wrote."""
    torch = torch_cuda
    from test_box_compact import model
    from test_gpu_canaries import Guarded
    g = Guarded(torch)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fields = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=100 + n)]
    ref_t, ref_fl = oracle.box_soa(fields)
    ref_fl = np.asarray(ref_fl).view(u32)
    rng = np.random.default_rng(n)
    item = rng.integers(0, 2 ** 32, n, dtype=np.uint32) if tables else None
    children = rng.integers(0, 2 ** 32, (n, 4), dtype=np.uint32) if tables else None
    want_i, want_c, want_t = model(ref_t, ref_fl, item, children, tables)
    operands = dev(pack_numpy(fields).view(np.int32))
    results = torch.full((capi.tiled_words(n, 5),), SENTINEL, dtype=torch.int32, device="cuda")
    b = capi.Batch()
    b.set_box_tiled(operands, results, n)
    torch.cuda.synchronize()
    ctx.eval(b)
    cap = max(1, want_i.shape[0])
    oi, oc = g.empty((cap,), torch.int32), g.empty((cap,), torch.int32)
    ot = g.empty((cap,), torch.float32) if tables else None
    torch.cuda.synchronize()
    count = ctx.box_compact_tiled(results, n, oi, oc, ot, None if item is None else dev(item.view(np.int32)),
                                  None if children is None else dev(children.view(np.int32)))
    torch.cuda.synchronize()
    g.check()
    assert count == want_i.shape[0]
    assert np.array_equal(oi.cpu().numpy().view(u32)[:count], want_i)
    assert np.array_equal(oc.cpu().numpy().view(u32)[:count], want_c)
    if tables:
        assert np.array_equal(ot.cpu().numpy().view(u32)[:count], want_t.view(u32))


@pytest.mark.parametrize("n", [1, 300, T, T + 1, 3 * T + 17])
def test_tiled_segments_in_host_buffers(ctx, oracle, n):
    """Without RT_DP_F_DEVICE_PTRS the library stages a tiled segment in one copy each way; the bytes behind op n - 1 of the
    caller's result buffers stay as they were."""
    bf = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=61 + n, edge_frac=0.2)]
    tf, tk = gen.tri_ops(n, seed=62 + n, edge_frac=0.2)
    tf = [np.ascontiguousarray(f) for f in tf]
    box_in, tri_in = pack_numpy(bf), pack_numpy(tf + [tk])
    box_out = np.full(capi.tiled_words(n, 5), SENTINEL, u32)
    tri_out = np.full(capi.tiled_words(n, 3), SENTINEL, u32)
    a, bb, ctl, _ = gen.vec_beats(50, 16, seed=63)
    acc = np.zeros(ctl.shape[0], f32)
    b = capi.Batch()
    b.set_box_tiled(box_in, box_out, n)
    b.set_tri_tiled(tri_in, tri_out, n)
    b.set_euclid(a, bb, ctl, acc)
    ctx.reset()
    ctx.eval(b)
    want_t, want_f = oracle.box_soa(bf)
    got = unpack_numpy(box_out, 5, n)
    for k in range(4):
        assert_bits_equal(got[k], want_t[k], "tmin_out[%d]" % k)
    assert_bits_equal(got[4], want_f, "flags")
    wn, wd, wh = oracle.tri_soa(tf, tk)
    got = unpack_numpy(tri_out, 3, n)
    assert_bits_equal(got[0], wn, "t_num")
    assert_bits_equal(got[1], wd, "t_denom")
    assert_bits_equal(got[2], wh, "hit")
    assert_bits_equal(acc, oracle.euclid_beats(a, bb, ctl)[0], "euclidean beats next to the tiled segments")
    for raw, words in ((box_out, 5), (tri_out, 3)):
        mask = np.ones(raw.shape[0], bool)
        for w in range(words):
            mask[tile_index(np.arange(n), w, words)] = False
        assert (raw[mask] == SENTINEL).all(), "bytes behind op n - 1 of a host result buffer were overwritten"
