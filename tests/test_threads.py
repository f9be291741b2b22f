"""Distinct contexts driven from distinct host threads at the same time (the C calls release the GIL): every
thread's results are the oracle's, the accumulator registers of one context never see the other's beats, and
an error raised on one thread leaves the other thread's error text alone."""
import threading

import numpy as np
import pytest

from raytracer_amd import capi
from oracle import gen

from helpers import assert_bits_equal, hip_box, hip_euclid, hip_tri

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32


def test_two_contexts_on_two_threads(oracle, torch_cuda):
    torch = torch_cuda
    rounds, n = 6, 100003
    work = []
    for t in range(2):
        box = gen.box_ops(n, seed=900 + t, edge_frac=0.2)
        tf, tk = gen.tri_ops(n, seed=910 + t, edge_frac=0.2)
        a, b, ctl, _ = gen.vec_beats(4000, 16, seed=920 + t, edge_frac=0.1)
        ctl = ctl.copy()
        ctl[-1] &= u32(0xFFFF)              # the stream ends inside a pair: the registers must carry over
        work.append((box, tf, tk, a, b, ctl))
    want = []
    for box, tf, tk, a, b, ctl in work:
        acc, outs = 0.0, []
        for _ in range(rounds):
            out, acc = oracle.euclid_beats(a, b, ctl, acc)
            outs.append(out)
        want.append((oracle.box_soa(box), oracle.tri_soa(tf, tk), outs))

    errors, got = [], [None, None]
    start = threading.Barrier(2)

    def worker(t):
        try:
            torch.cuda.set_device(0)
            c = capi.Context()
            box, tf, tk, a, b, ctl = work[t]
            start.wait()
            res = []
            for _ in range(rounds):
                res.append((hip_box(c, box, True, torch), hip_tri(c, tf, tk, True, torch), hip_euclid(c, a, b, ctl, True, torch)))
                if t == 1:
                    bad = capi.Batch()
                    bad.ob.n[capi.OP_QUADBOX] = 5
                    with pytest.raises(capi.RtDpError) as e:
                        c.eval(bad)
                    assert "box" in str(e.value)
            got[t] = res
            c.close()
        except BaseException as exc:  # noqa: BLE001 - reported on the main thread
            errors.append((t, exc))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=100)
    assert not errors, errors
    for t in range(2):
        (wt, wf), (wn, wd, wh), outs = want[t]
        for r in range(rounds):
            (gt, gf), (gn, gd, gh), ge = got[t][r]
            for k in range(4):
                assert_bits_equal(gt[k], wt[k], "thread %d tmin_out[%d]" % (t, k))
            assert_bits_equal(gf, wf, "thread %d flags" % t)
            assert_bits_equal(gn, wn, "thread %d t_num" % t)
            assert_bits_equal(gd, wd, "thread %d t_denom" % t)
            assert_bits_equal(gh, wh, "thread %d hit" % t)
            assert_bits_equal(ge, outs[r], "thread %d accumulator, call %d" % (t, r))
