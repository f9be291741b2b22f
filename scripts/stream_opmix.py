import sys, time, torch
sys.path.insert(0, '.')
from raytracer_amd import capi
n = 1 << 24
dev = 'cuda'
g = torch.Generator(device=dev); g.manual_seed(1)
rec = torch.zeros(n, 84, dtype=torch.int32, device=dev)
vals = (torch.rand(n, 81, generator=g, device=dev) * 200 - 100)
rec[:, 1:82] = vals.view(torch.int32)
rec[:, 11:14] = torch.randint(0, 3, (n, 3), generator=g, device=dev, dtype=torch.int32)
rec[:, 82] = 0xFFFF
rec[:, 83] = torch.randint(0, 2, (n,), generator=g, device=dev, dtype=torch.int32)
out_all = torch.empty(n + 1, 21, dtype=torch.int32, device=dev)
out = out_all[:n]
ctx = capi.Context(device=0)
ctx.set_option("stream_mode", 2)
for name, mk in [("mixed", lambda: torch.randint(0, 4, (n,), generator=g, device=dev, dtype=torch.int32)),
                 ("all tri", lambda: torch.zeros(n, dtype=torch.int32, device=dev)),
                 ("all box", lambda: torch.ones(n, dtype=torch.int32, device=dev)),
                 ("all euc", lambda: torch.full((n,), 2, dtype=torch.int32, device=dev)),
                 ("all ang", lambda: torch.full((n,), 3, dtype=torch.int32, device=dev)),
                 ("box|tri", lambda: torch.randint(0, 2, (n,), generator=g, device=dev, dtype=torch.int32)),
                 ("sorted 4096", lambda: ((torch.arange(n, device=dev) >> 12) & 3).to(torch.int32))]:
    rec[:, 0] = mk()
    torch.cuda.synchronize()
    for _ in range(3): ctx.eval_bundles_device(rec, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.eval_bundles_device(rec, out)
    torch.cuda.synchronize(); per = (time.perf_counter() - t0) / 10
    print("%-12s %.4f ms  frac %.3f" % (name, per * 1e3, n * 420 / per / 8e12), flush=True)

# the same mixed stream into an output array that starts 84 bytes into an allocation (4-byte aligned only)
rec[:, 0] = torch.randint(0, 4, (n,), generator=g, device=dev, dtype=torch.int32)
for name, o in (("aligned out", out_all[:n]), ("out + 84 B", out_all[1:])):
    torch.cuda.synchronize()
    for _ in range(3): ctx.eval_bundles_device(rec, o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.eval_bundles_device(rec, o)
    torch.cuda.synchronize(); per = (time.perf_counter() - t0) / 10
    print("%-12s %.4f ms  frac %.3f" % (name, per * 1e3, n * 420 / per / 8e12), flush=True)
