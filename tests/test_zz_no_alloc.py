"""rt_dp_box_compact_static (the allocation-free form: offsets in a fixed device array inside the library, count
delivered to device memory) against rt_dp_box_compact on the same Quadbox results, bit for bit, on two calls in a
row of which the second has the smaller input (so stale offsets of the first call sit behind the ones the second
call uses), and its size limit."""
import ctypes as C

import numpy as np
import pytest

from oracle import gen
from raytracer_amd import capi

from test_gpu_canaries import Guarded

pytestmark = pytest.mark.gpu


def _box_results(torch, ctx, n, seed):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fields = gen.box_ops(n, seed=seed, edge_frac=0.2)
    tmin = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]
    flags = torch.empty(n, dtype=torch.int32, device="cuda")
    b = capi.Batch()
    b.set_box([dev(f) for f in fields], tmin, flags, n)
    torch.cuda.synchronize()
    ctx.eval(b)
    return tmin, flags


def _compare(torch, ctx, n, seed, tables):
    i32, f32 = torch.int32, torch.float32
    tmin, flags = _box_results(torch, ctx, n, seed)
    rng = np.random.default_rng(seed)
    item = torch.from_numpy(rng.integers(0, 2 ** 31, n, dtype=np.int32)).cuda() if tables else None
    children = torch.from_numpy(rng.integers(0, 2 ** 31, (n, 4), dtype=np.int32)).cuda() if tables else None
    cap = 4 * n
    # the existing entry point
    wi, wc, wt = torch.zeros(cap, dtype=i32, device="cuda"), torch.zeros(cap, dtype=i32, device="cuda"), torch.zeros(cap, dtype=f32, device="cuda")
    want = ctx.box_compact(tmin, flags, wi, wc, wt, item, children)
    # the allocation-free one, between guard zones, on a caller stream
    g = Guarded(torch)
    gi, gc, gt = g.empty((cap,), i32), g.empty((cap,), i32), g.empty((cap,), f32)
    d_count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx.box_compact_static(tmin, flags, gi, gc, d_count, gt, item, children, stream=stream.cuda_stream)
    stream.synchronize()
    torch.cuda.synchronize()
    g.check()
    got = int(d_count.item())
    assert got == want and 0 < got <= cap
    assert torch.equal(gi[:got], wi[:got]) and torch.equal(gc[:got], wc[:got])
    assert torch.equal(gt[:got].view(i32), wt[:got].view(i32))
    # a short list is cut, never overrun; the count still reports the whole batch
    g2 = Guarded(torch)
    si, sc = g2.empty((got // 2,), i32), g2.empty((got // 2,), i32)
    d_count.fill_(-1)
    torch.cuda.synchronize()
    ctx.box_compact_static(None, flags, si, sc, d_count, None, item, children)
    ctx.synchronize()
    torch.cuda.synchronize()
    g2.check()
    assert int(d_count.item()) == want
    assert torch.equal(si, wi[:got // 2]) and torch.equal(sc, wc[:got // 2])


def test_static_compaction_equals_the_allocating_one_twice_in_a_row(torch_cuda):
    torch = torch_cuda
    ctx = capi.Context()
    _compare(torch, ctx, 300007, 501, True)     # 1172 workgroups' worth of offsets in the static array
    _compare(torch, ctx, 4099, 502, False)      # then a smaller input: reads only what it wrote itself
    _compare(torch, ctx, 65, 503, True)
    ctx.close()


def test_static_compaction_empty_batch_and_limits(torch_cuda):
    torch = torch_cuda
    ctx = capi.Context()
    L = capi.load()
    d_count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    out = torch.zeros(8, dtype=torch.int32, device="cuda")
    flags = torch.zeros(8, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.box_compact_static(None, flags, out, out.clone(), d_count, n=0)      # empty: the count is still written
    ctx.synchronize()
    assert int(d_count.item()) == 0
    ctx.box_compact_static(None, flags, out, out.clone(), d_count)           # no hits at all
    ctx.synchronize()
    assert int(d_count.item()) == 0
    with pytest.raises(capi.RtDpError) as e:                                  # beyond the static array: rejected, nothing launched
        ctx.box_compact_static(None, flags, out, out.clone(), d_count, n=(1 << 24) + 1)
    assert e.value.code == capi.EINVAL and "static scratch" in str(e.value)
    ro = capi.BoxOut()
    ro.flags = flags.data_ptr()
    assert L.rt_dp_box_compact_static(ctx._h, C.byref(ro), 8, None, None, out.data_ptr(), out.data_ptr(), None, 8, None, None) == capi.ENULL
    ctx.close()


def test_static_compaction_allocates_nothing(torch_cuda):
    """The property the entry point exists for.  (1) Free device memory is the same before and after warmed-up calls.
    (2) The call is captured into a HIP graph under the strictest capture mode torch offers (any hipMalloc, hipFree or
    synchronisation on the capturing thread ends the capture with an error) and the replayed graph delivers the list of
    the allocating entry point for whatever the result arrays hold by then."""
    torch = torch_cuda
    i32, f32 = torch.int32, torch.float32
    ctx = capi.Context()
    n = 20011
    tmin, flags = _box_results(torch, ctx, n, 504)
    cap = 4 * n
    wi, wc, wt = torch.zeros(cap, dtype=i32, device="cuda"), torch.zeros(cap, dtype=i32, device="cuda"), torch.zeros(cap, dtype=f32, device="cuda")
    want = ctx.box_compact(tmin, flags, wi, wc, wt)
    gi, gc, gt = torch.zeros(cap, dtype=i32, device="cuda"), torch.zeros(cap, dtype=i32, device="cuda"), torch.zeros(cap, dtype=f32, device="cuda")
    d_count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    ctx.box_compact_static(tmin, flags, gi, gc, d_count, gt, stream=stream.cuda_stream)     # warm-up: code objects are loaded
    stream.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        ctx.box_compact_static(tmin, flags, gi, gc, d_count, gt, stream=stream.cuda_stream)
    stream.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "device memory moved across rt_dp_box_compact_static calls"

    gi.zero_(); gc.zero_(); gt.zero_(); d_count.fill_(-1)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
        ctx.box_compact_static(tmin, flags, gi, gc, d_count, gt, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(d_count.item()) == -1, "capture must not execute"
    graph.replay()
    torch.cuda.synchronize()
    got = int(d_count.item())
    assert got == want
    assert torch.equal(gi[:got], wi[:got]) and torch.equal(gc[:got], wc[:got]) and torch.equal(gt[:got].view(i32), wt[:got].view(i32))
    # new results in the same arrays, replay again: the graph holds no state of the first run
    tmin2, flags2 = _box_results(torch, ctx, n, 505)
    for dst, src in zip(tmin, tmin2):
        dst.copy_(src)
    flags.copy_(flags2)
    want2 = ctx.box_compact(tmin, flags, wi, wc, wt)
    graph.replay()
    torch.cuda.synchronize()
    got2 = int(d_count.item())
    assert got2 == want2 and torch.equal(gi[:got2], wi[:got2]) and torch.equal(gc[:got2], wc[:got2])
    ctx.close()
