#!/usr/bin/env python3
"""Does the kind of device allocation the caller's operand arrays live in change the Quadbox kernel's rate?
One pool per kind from hipExtMallocWithFlags (default, fine-grained, uncached, contiguous), array k at
k * (4 n + extra) bytes, the same launches as scripts/placement_sweep.py.

    python scripts/alloc_kind_sweep.py [log2_ops ...]        (GPU box)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from raytracer_amd import capi  # noqa: E402

KINDS = {"default": 0x0, "finegrained": 0x1, "uncached": 0x3, "contiguous": 0x4}


class Pool:
    """A device allocation exposed to torch through the CUDA array interface."""

    def __init__(self, hip, n_floats, flags):
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(n_floats * 4), C.c_uint(flags))
        if rc != 0:
            raise RuntimeError("hipExtMallocWithFlags(flags=%d) -> %d" % (flags, rc))
        self.hip, self.ptr, self.n = hip, p.value, n_floats
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (self.ptr, False), "version": 2}

    def free(self):
        self.hip.hipFree(C.c_void_p(self.ptr))


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [24, 26]
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipFree.argtypes = [C.c_void_p]
    ctx = capi.Context(device=0)
    stream = torch.cuda.Stream()
    for lg in sizes:
        n = 1 << lg
        for extra_kib in (0, 4, 65536):
            pitch = n + extra_kib * 256
            for kind, flags in KINDS.items():
                try:
                    pool = Pool(hip, 39 * pitch + 64, flags)
                except RuntimeError as e:
                    print("n 2^%d %-12s extra %6d KiB: %s" % (lg, kind, extra_kib, e), flush=True)
                    continue
                t = torch.as_tensor(pool, device="cuda")
                step = 1 << 26
                for s in range(0, t.numel(), step):      # random operands, a piece at a time
                    t[s:s + step] = torch.rand(min(step, t.numel() - s), device="cuda") * 200 - 100
                arr = [t[k * pitch:k * pitch + n] for k in range(39)]
                b = capi.Batch()
                b.set_box(arr[:34], arr[34:38], arr[38].view(torch.int32), n)
                b.finalize()
                torch.cuda.synchronize()
                for _ in range(10):
                    ctx.launch(b, stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(40):
                    ctx.launch(b, stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
                ms = e0.elapsed_time(e1) / 40
                print("n 2^%d %-12s extra %6d KiB  %.4f ms  frac %.4f" % (lg, kind, extra_kib, ms, n * 156 / ms / 1e6 / 8000.0), flush=True)
                del arr, b, t
                torch.cuda.synchronize()
                pool.free()
    ctx.close()


if __name__ == "__main__":
    main()
