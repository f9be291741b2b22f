"""rt_dp_box_compact: the dense visit list behind a Quadbox batch equals the list read off the oracle's
results op by op (ops in order, hits closest first), inside guard zones, for sizes around the kernels'
wave and workgroup granularities."""
import numpy as np
import pytest

from raytracer_amd import capi, gen

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
