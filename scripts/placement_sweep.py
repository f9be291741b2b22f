#!/usr/bin/env python3
"""How the placement of the 39 operand / result arrays of a Quadbox batch against each other changes the kernel's
rate: one pool of random floats, array k at k * (n * 4 + extra) bytes, the same kernel launched for a list of
`extra` values.  Contents do not matter for the timing, so nothing is copied between placements.

    python scripts/placement_sweep.py [log2_ops ...]        (GPU box; prints one line per placement)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from raytracer_amd import capi  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [24, 26]
    extras_kib = sorted(set([0] + [(m << k) for k in range(0, 18) for m in (1, 3)] + [65536 + 4, 65536 + 64, 2048 + 4, 2048 + 64]))
    extras_kib = [e for e in extras_kib if e <= 196608]
    if os.environ.get("SWEEP_FEW"):
        extras_kib = [0, 1, 4, 12, 64, 512, 1024, 2052, 4096, 16384, 65536, 65540, 131072]
    ctx = capi.Context(device=0)
    ctx.set_tuning(int(os.environ.get("SWEEP_OPS_PER_LANE", "0")), 0, 0)
    ctx.set_option("wgs_per_cu", int(os.environ.get("SWEEP_WGS", "0")))
    stream = torch.cuda.Stream()
    for lg in sizes:
        n = 1 << lg if lg < 40 else lg
        max_pitch = n + max(extras_kib) * 256
        pool = torch.rand(39 * max_pitch + 64, device="cuda") * 200 - 100
        for e in extras_kib:
            pitch = n + e * 256
            arr = [pool[k * pitch:k * pitch + n] for k in range(39)]
            b = capi.Batch()
            b.set_box(arr[:34], arr[34:38], arr[38].view(torch.int32), n)
            b.finalize()
            for _ in range(10):
                ctx.launch(b, stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(40):
                ctx.launch(b, stream.cuda_stream)
            e1.record(stream)
            stream.synchronize()
            ms = e0.elapsed_time(e1) / 40
            print("n 2^%d extra %7d KiB  pitch %% 2MiB %5d KiB  %.4f ms  frac %.4f" % (lg, e, (pitch * 4 // 1024) % 2048, ms, n * 156 / ms / 1e6 / 8000.0), flush=True)
        del pool
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
