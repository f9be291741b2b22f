"""Does a kernel's duration depend on which kernel ran before it?  Launches the four op kernels (2^LOG2 units each, resident)
in several orders on one stream, with an event pair around every launch, and prints the mean duration of each opcode per order.
python3 scripts/probe/order_probe.py [log2_ops]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from raytracer_amd import capi  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
ctx = capi.Context(0)
if variant:
    ctx.set_option("variant", variant)
print("variant", variant, flush=True)
n = 1 << lg
B = bench.make_batches(n, dev, ["box", "tri", "euclid", "angular"], layout="tiled", ctx=ctx)
B["euclid2"] = bench.make_batches(n, dev, ["euclid"], layout="tiled", ctx=ctx)["euclid"]   # the same kernel over other arrays
B["angular2"] = bench.make_batches(n, dev, ["angular"], layout="tiled", ctx=ctx)["angular"]
stream = bench.bench_stream()
names = {"B": "box", "T": "tri", "E": "euclid", "A": "angular", "F": "euclid2", "C": "angular2"}


def touch(batch, stride_bytes=65536):
    """one word per stride of every array of the batch (a page toucher: moves almost no data)"""
    acc = None
    for t in batch._keep:
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0:
            v = t.reshape(-1).view(torch.int32)[::stride_bytes // 4].sum()
            acc = v if acc is None else acc + v
    return acc


touching = os.environ.get("ORDER_PROBE_TOUCH") == "1"
print("touch before every launch" if touching else "no touching", flush=True)
torch.cuda.synchronize()
orders = sys.argv[3].split(",") if len(sys.argv) > 3 else ("EEEE", "AAAA", "EAEA", "EBAB", "ETAT", "EEAA", "EEEEAAAA", "BTEA")
for order in orders:
    seq = list(order) * 6
    with torch.cuda.stream(stream):
        for c in seq[:len(order)]:
            ctx.launch(B[names[c]][0], stream.cuda_stream)
        ev = []
        for c in seq:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if touching:
                touch(B[names[c]][0])
            e0.record(stream)
            ctx.launch(B[names[c]][0], stream.cuda_stream)
            e1.record(stream)
            ev.append((c, e0, e1))
    torch.cuda.synchronize()
    acc = {}
    for c, e0, e1 in ev:
        acc.setdefault(c, []).append(e0.elapsed_time(e1))
    print(order, " ".join("%s %.1f us" % (names[c], 1000 * sum(v) / len(v)) for c, v in acc.items()), flush=True)
