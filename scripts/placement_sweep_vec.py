#!/usr/bin/env python3
"""The beat kernels against the distance between their two operand arrays: one pool, `a` at its start, `b` at
n * width * 4 + extra bytes behind it; control words and results as from the stream generator.

    python scripts/placement_sweep_vec.py [width ...]        (GPU box; width 8 = angular, 16 = Euclidean)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from raytracer_amd import capi, synth  # noqa: E402


def main():
    widths = [int(x) for x in sys.argv[1:]] or [8, 16]
    extras_kib = [0, 1, 2, 4, 8, 16, 64, 256, 1024, 2052, 4096, 16384, 65536, 65540, 131072, 262144 + 64]
    ctx = capi.Context(device=0)
    stream = torch.cuda.Stream()
    for width in widths:
        _, _, ctl = synth.vec_stream(1 << 24, width, 20241010 if width == 16 else 20241011, "cuda")
        n = int(ctl.shape[0])
        per_unit = 136 if width == 16 else 76
        out0 = torch.empty(n, dtype=torch.float32, device="cuda")
        out1 = torch.empty(n, dtype=torch.float32, device="cuda")
        span = n * width
        pool = torch.rand(2 * span + max(extras_kib) * 256 + 64, device="cuda") * 200 - 100
        for e in extras_kib:
            off = (span + e * 256 + 3) // 4 * 4       # keep 16-byte alignment
            a = pool[:span].view(n, width)
            b = pool[off:off + span].view(n, width)
            bt = capi.Batch()
            if width == 16:
                bt.set_euclid(a, b, ctl, out0)
            else:
                bt.set_angular(a, b, ctl, out0, out1)
            bt.finalize()
            for _ in range(30):
                ctx.launch(bt, stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(60):
                ctx.launch(bt, stream.cuda_stream)
            e1.record(stream)
            stream.synchronize()
            ms = e0.elapsed_time(e1) / 60
            print("width %2d extra %7d KiB  %.4f ms  frac %.4f" % (width, e, ms, n * per_unit / ms / 1e6 / 8000.0), flush=True)
        del pool
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
