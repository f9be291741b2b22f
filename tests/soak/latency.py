"""Per-call latency of the entry points on small batches (host and device pointers).  python tests/soak/latency.py"""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from raytracer_amd import capi
from oracle import gen
from test_gpu_parity import _mixed_stream
from helpers import hip_box
c = capi.Context()
def timeit(f, reps=300):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
for n in (100, 1000, 10000, 50000, 100000):
    s = _mixed_stream(1, n // 4, n // 4, n // 40 + 1, n // 40 + 1)
    print("eval_bundles host, %6d records: %8.1f us" % (s.shape[0], timeit(lambda: c.eval_bundles(s))))
    d_in = torch.from_numpy(s.view(np.int32)).cuda(); d_out = torch.empty((s.shape[0], 21), dtype=torch.int32, device="cuda")
    for mode in (1, 2, 3):
        c.set_option("stream_mode", mode)
        print("eval_bundles_device mode %d, %6d records: %8.1f us" % (mode, s.shape[0], timeit(lambda: c.eval_bundles_device(d_in, d_out))))
    c.set_option("stream_mode", 0)
    f = [np.ascontiguousarray(x) for x in gen.box_ops(n, seed=3)]
    tmin = [np.empty(n, np.float32) for _ in range(4)]; fl = np.empty(n, np.uint32)
    b = capi.Batch(); b.set_box(f, tmin, fl, n)
    print("eval box host SoA, %6d ops: %8.1f us" % (n, timeit(lambda: c.eval(b))))
    df = [torch.from_numpy(x).cuda() for x in f]
    dt = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]; dfl = torch.empty(n, dtype=torch.int32, device="cuda")
    b2 = capi.Batch(); b2.set_box(df, dt, dfl, n)
    print("eval box device SoA, %6d ops: %8.1f us" % (n, timeit(lambda: c.eval(b2))))


def mixed_device_batch(n):
    """A batch of all four opcodes in device arrays: n box ops, n triangle ops, about n / 4 beats of each kind."""
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    b = capi.Batch()
    keep = []
    bf = [dev(x) for x in gen.box_ops(n, seed=5)]
    bo = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)] + [torch.empty(n, dtype=torch.int32, device="cuda")]
    b.set_box(bf, bo[:4], bo[4], n)
    tf, tk = gen.tri_ops(n, seed=6)
    tfd, tkd = [dev(x) for x in tf], dev(tk.view(np.int32))
    to = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(2)] + [torch.empty(n, dtype=torch.int32, device="cuda")]
    b.set_tri(tfd, tkd, to[0], to[1], to[2], n)
    a, bb, ctl, _ = gen.vec_beats(max(n // 16, 2), 16, seed=7)
    ea, eb, ec = dev(a), dev(bb), dev(ctl.view(np.int32))
    eo = torch.empty(ctl.shape[0], dtype=torch.float32, device="cuda")
    b.set_euclid(ea, eb, ec, eo)
    qa, qb, qctl, _ = gen.vec_beats(max(n // 32, 2), 8, seed=8)
    aa, ab, ac = dev(qa), dev(qb), dev(qctl.view(np.int32))
    ao = [torch.empty(qctl.shape[0], dtype=torch.float32, device="cuda") for _ in range(2)]
    b.set_angular(aa, ab, ac, ao[0], ao[1])
    b.finalize()
    keep += [bf, bo, tfd, tkd, to, ea, eb, ec, eo, aa, ab, ac, ao]
    return b, keep


# Small batches are bound by the launches, not by HBM: the same 50 batches given one by one and as one captured graph
K = 50
stream = torch.cuda.Stream()
for n in (100, 1000, 10000):
    b, keep = mixed_device_batch(n)
    c.launch(b, stream.cuda_stream); stream.synchronize()
    def direct():
        for _ in range(K): c.launch(b, stream.cuda_stream)
        stream.synchronize()
    t_direct = timeit(direct, 40) / K
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
        for _ in range(K): c.launch(b, torch.cuda.current_stream().cuda_stream)
    def replay():
        g.replay(); torch.cuda.synchronize()
    t_graph = timeit(replay, 40) / K
    print("mixed batch of %5d box + %5d tri ops + beats, device arrays: %6.1f us per batch launched directly, %6.1f us inside a replayed graph of %d" % (n, n, t_direct, t_graph, K))
