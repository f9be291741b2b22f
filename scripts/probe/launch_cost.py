"""Host cost of rt_dp_launch per opcode on small batches: time for the calls to return (enqueue) and until the stream is idle.
python3 scripts/probe/launch_cost.py [log2_ops]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from raytracer_amd import capi  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
dev = torch.device("cuda:0")
ctx = capi.Context(0)
B = bench.make_batches(1 << lg, dev, ["box", "tri", "euclid", "angular"], layout="tiled", ctx=ctx)
stream = bench.bench_stream()
N = 2000
for name in ("box", "tri", "euclid", "angular", "box", "euclid"):
    b = B[name][0]
    for _ in range(200):
        ctx.launch(b, stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.launch(b, stream.cuda_stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-8s 2^%d: enqueue %.2f us per launch, until idle %.2f us per launch" % (name, lg, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N), flush=True)
