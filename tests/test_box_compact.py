"""rt_dp_box_compact: the dense visit list behind a Quadbox batch equals the list read off the oracle's
results op by op (ops in order, hits closest first), inside guard zones, for sizes around the kernels'
wave and workgroup granularities."""
import numpy as np
import pytest

from raytracer_amd import capi
from oracle import gen

from test_gpu_canaries import Guarded

pytestmark = pytest.mark.gpu


def model(tmin, flags, item, children, with_t):
    out_i, out_c, out_t = [], [], []
    for i in range(flags.shape[0]):
        fl = int(flags[i])
        for k in range(4):
            if not (fl >> k) & 1:
                break                       # hits come first
            slot = (fl >> (4 + 2 * k)) & 3
            out_i.append(i if item is None else int(item[i]))
            out_c.append(slot if children is None else int(children[i, slot]))
            if with_t:
                out_t.append(tmin[k][i])
    return (np.array(out_i, dtype=np.uint32), np.array(out_c, dtype=np.uint32), np.array(out_t, dtype=np.float32))


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 1000, 4099, 70001])
@pytest.mark.parametrize("tables", [False, True])
def test_compaction_matches_the_oracle(torch_cuda, oracle, n, tables):
    torch = torch_cuda
    g = Guarded(torch)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fields = gen.box_ops(n, seed=100 + n)
    ref_t, ref_fl = oracle.box_soa(fields)
    ref_fl = np.asarray(ref_fl).view(np.uint32)
    rng = np.random.default_rng(n)
    item = rng.integers(0, 2 ** 32, n, dtype=np.uint32) if tables else None
    children = rng.integers(0, 2 ** 32, (n, 4), dtype=np.uint32) if tables else None
    want_i, want_c, want_t = model(ref_t, ref_fl, item, children, tables)

    c = capi.Context()
    f32, i32 = torch.float32, torch.int32
    tmin = [torch.empty(n, dtype=f32, device="cuda") for _ in range(4)]
    flags = torch.empty(n, dtype=i32, device="cuda")
    b = capi.Batch()
    b.set_box([dev(f) for f in fields], tmin, flags, n)
    torch.cuda.synchronize()
    c.eval(b)
    assert np.array_equal(flags.cpu().numpy().view(np.uint32), ref_fl)

    cap = max(1, want_i.shape[0])
    oi, oc = g.empty((cap,), i32), g.empty((cap,), i32)
    ot = g.empty((cap,), f32) if tables else None
    torch.cuda.synchronize()
    count = c.box_compact(tmin if tables else None, flags, oi, oc, ot,
                          None if item is None else dev(item.view(np.int32)), None if children is None else dev(children.view(np.int32)))
    torch.cuda.synchronize()
    g.check()
    assert count == want_i.shape[0]
    assert np.array_equal(oi.cpu().numpy().view(np.uint32)[:count], want_i)
    assert np.array_equal(oc.cpu().numpy().view(np.uint32)[:count], want_c)
    if tables:
        assert np.array_equal(ot.cpu().numpy().view(np.uint32)[:count], want_t.view(np.uint32))

    # a short list is cut, never overrun; the count still reports the whole batch
    if count > 2:
        g2 = Guarded(torch)
        si, sc = g2.empty((count // 2,), i32), g2.empty((count // 2,), i32)
        torch.cuda.synchronize()
        assert c.box_compact(None, flags, si, sc, None, None if item is None else dev(item.view(np.int32)),
                             None if children is None else dev(children.view(np.int32))) == count
        torch.cuda.synchronize()
        g2.check()
        assert np.array_equal(si.cpu().numpy().view(np.uint32), want_i[:count // 2])
        assert np.array_equal(sc.cpu().numpy().view(np.uint32), want_c[:count // 2])
    c.close()


def test_compaction_argument_errors(torch_cuda):
    torch = torch_cuda
    c = capi.Context()
    flags = torch.zeros(8, dtype=torch.int32, device="cuda")
    out = torch.zeros(8, dtype=torch.int32, device="cuda")
    ot = torch.zeros(8, dtype=torch.float32, device="cuda")
    assert c.box_compact(None, flags, out, out.clone(), None) == 0           # no hits at all
    with pytest.raises(capi.RtDpError):
        c.box_compact(None, flags, out, out.clone(), ot)                     # distances asked for, none given
    c.close()


def test_a_callers_own_traversal_loop_equals_the_library_walk(torch_cuda):
    """Level-by-level closest-hit traversal assembled by the caller from the public pieces (operand gathers in
    torch, Quadbox and triangle batches with RT_DP_F_DERIVE_RAY, rt_dp_box_compact for the next level's visits)
    against rt_dp_trace_closest's breadth-first walk: same (t, triangle) for every ray, bit for bit."""
    torch = torch_cuda
    from raytracer_amd import bvh
    f32, u32, u64 = np.float32, np.uint32, np.uint64
    tris = bvh.random_scene(6000, seed=77)
    nodes, tris12, _ = bvh.build_bvh4(tris, leaf_size=4)
    org, dirs = bvh.rays_towards(tris, 30000, seed=78, jitter=8.0)
    extent = 4000.0
    n = org.shape[0]
    c = capi.Context()
    N, T = torch.from_numpy(nodes).cuda(), torch.from_numpy(tris12).cuda()
    refs = N[:, 24:28].contiguous().view(torch.int32)
    O = torch.from_numpy(org).cuda()
    D = torch.from_numpy(dirs).cuda()

    want_t = torch.empty(n, dtype=torch.float32, device="cuda")
    want_i = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    c.trace_closest(N, T, [O[:, a].contiguous() for a in range(3)], [D[:, a].contiguous() for a in range(3)], extent, want_t, want_i, wavefront=True)

    best = np.full(n, u64(0x7F800000FFFFFFFF), u64)
    ray = torch.arange(n, dtype=torch.int64, device="cuda")
    node = torch.zeros(n, dtype=torch.int64, device="cuda")
    levels = 0
    while ray.numel():
        levels += 1
        m = int(ray.numel())
        found = torch.from_numpy((best >> u64(32)).astype(u32).view(f32)).cuda()[ray]
        ext = torch.where(found < extent, found, torch.full_like(found, extent))
        fields = [O[ray, a].contiguous() for a in range(3)] + [D[ray, a].contiguous() for a in range(3)] + [None] * 3 + [ext]
        fields += [N[node, w].contiguous() for w in range(24)]
        flags = torch.empty(m, dtype=torch.int32, device="cuda")
        tmin = [torch.empty(m, dtype=torch.float32, device="cuda") for _ in range(4)]
        b = capi.Batch().derive_ray()
        b.set_box(fields, tmin, flags, m)
        torch.cuda.synchronize()
        c.eval(b)
        item = ray.to(torch.int32).contiguous()
        children = refs[node].contiguous()
        oi = torch.empty(4 * m, dtype=torch.int32, device="cuda")
        oc = torch.empty(4 * m, dtype=torch.int32, device="cuda")
        count = c.box_compact(None, flags, oi, oc, None, item, children)
        vr, vc = oi[:count].to(torch.int64), oc[:count].to(torch.int64) & 0xFFFFFFFF
        live = vc != bvh.NONE
        leaf = live & ((vc & bvh.LEAF) != 0)
        inner = live & ~leaf
        lray, lref = vr[leaf], vc[leaf]
        if lray.numel():
            first, cnt = (lref & 0x7FFFFFFF) >> 3, (lref & 7) + 1
            tray = torch.repeat_interleave(lray, cnt)
            start = torch.cumsum(cnt, 0) - cnt
            tid = torch.repeat_interleave(first, cnt) + (torch.arange(int(cnt.sum()), device="cuda") - torch.repeat_interleave(start, cnt))
            k = int(tid.numel())
            tf = [O[tray, a].contiguous() for a in range(3)] + [D[tray, a].contiguous() for a in range(3)] + [T[tid, w].contiguous() for w in range(9)]
            num = torch.empty(k, dtype=torch.float32, device="cuda")
            den = torch.empty(k, dtype=torch.float32, device="cuda")
            hit = torch.empty(k, dtype=torch.int32, device="cuda")
            tb = capi.Batch().derive_ray()
            tb.set_tri(tf, None, num, den, hit, k)
            torch.cuda.synchronize()
            c.eval(tb)
            with np.errstate(all="ignore"):
                t = (num.cpu().numpy() / den.cpu().numpy()).astype(f32)
            ok = (hit.cpu().numpy() != 0) & ~np.isnan(t)
            t = np.where(t == 0, f32(0), t).astype(f32)
            key = (t.view(u32).astype(u64) << u64(32)) | tid.cpu().numpy().astype(u64)
            np.minimum.at(best, tray.cpu().numpy()[ok], key[ok])
        ray, node = vr[inner].contiguous(), vc[inner].contiguous()
        assert levels < 64
    got_t = (best >> u64(32)).astype(u32)
    got_i = (best & u64(0xFFFFFFFF)).astype(u32)
    assert np.array_equal(got_t, want_t.cpu().numpy().view(u32))
    assert np.array_equal(got_i, want_i.cpu().numpy().view(u32))
    assert (got_i != 0xFFFFFFFF).sum() > n // 2          # most rays do hit something
    c.close()
