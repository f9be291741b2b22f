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
    for mode in (1, 2):
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
