"""On-device synthetic operand streams for the benchmark (torch is only the
allocator / RNG here).  Same distributions as the host generators of the test suite (oracle/gen.py), generated
directly in HBM so that a 2^24-op stream needs no host staging.

Reference generators mirrored: src/test/scala/raytracer_datapath/raytracer_gold/
SW_Data.scala:310-417, call sites Datapath_test.scala:142-225.
"""
import torch


def _next_float(shape, g, dev):
    return torch.randint(0, 1 << 24, shape, generator=g, device=dev, dtype=torch.int32).to(torch.float32) * (2.0 ** -24)


def _uniform(shape, lo, hi, g, dev):
    return _next_float(shape, g, dev) * (hi - lo) + lo


def box_stream(n, seed, dev):
    """34 float32 device arrays in capi.BOX_FIELDS order."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    fields = [_uniform((n,), -100.0, 100.0, g, dev) for _ in range(3)]            # origin
    dirs = [_uniform((n,), -100.0, 100.0, g, dev) for _ in range(3)]
    fields += dirs
    fields += [1.0 / d for d in dirs]                                              # inv = 1.0f / dir
    fields.append(torch.full((n,), 4000.0, device=dev, dtype=torch.float32))       # extent
    for _ in range(4):
        lo, hi = [], []
        for _a in range(3):
            r0 = _next_float((n,), g, dev) * 1e4 * 2 - 1e4
            r1 = _next_float((n,), g, dev) * 1e4 * 2 - 1e4
            lo.append(torch.minimum(r0, r1))
            hi.append(torch.maximum(r0, r1))
        fields += lo + hi
    return [f.contiguous() for f in fields]


def tri_stream(n, seed, dev):
    """(15 float32 device arrays in capi.TRI_FIELDS order, packed k as int32)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    vtx = [[_uniform((n,), -1000.0, 1000.0, g, dev) for _a in range(3)] for _v in range(3)]
    org = [_uniform((n,), -1000.0, 1000.0, g, dev) for _a in range(3)]
    d = torch.stack([((vtx[0][a] + vtx[1][a]) + vtx[2][a]) / 3.0 - org[a] for a in range(3)], dim=1)
    ad = d.abs()
    kz = torch.zeros(n, dtype=torch.int64, device=dev)
    m = ad[:, 0].clone()
    upd = ad[:, 1] > m
    kz[upd] = 1
    m = torch.where(upd, ad[:, 1], m)
    kz[ad[:, 2] > m] = 2
    kx = (kz + 1) % 3
    ky = (kx + 1) % 3
    dz = d.gather(1, kz[:, None])[:, 0]
    swap = dz < 0
    kx, ky = torch.where(swap, ky, kx), torch.where(swap, kx, ky)
    shear = [d.gather(1, kx[:, None])[:, 0] / dz, d.gather(1, ky[:, None])[:, 0] / dz, 1.0 / dz]
    k = (kx | (ky << 2) | (kz << 4)).to(torch.int32)
    fields = org + shear + vtx[0] + vtx[1] + vtx[2]
    return [f.contiguous() for f in fields], k.contiguous()


def vec_stream(n_beats_target, width, seed, dev, max_len=128):
    """Vector pairs of length U{1..max_len}, split into beats of `width`:
    a [beats, width], b [beats, width], ctl [beats] (int32 image of mask | reset << 16)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    mean_beats = sum((l + width - 1) // width for l in range(1, max_len + 1)) / max_len
    n_pairs = max(1, int(n_beats_target / mean_beats))
    lengths = torch.randint(0, max_len, (n_pairs,), generator=g, device=dev) + 1
    beats_per = (lengths + width - 1) // width
    last = torch.cumsum(beats_per, 0) - 1
    nb = int(last[-1].item()) + 1
    first = last - beats_per + 1
    pair_of = torch.repeat_interleave(torch.arange(n_pairs, device=dev), beats_per)
    in_pair = torch.arange(nb, device=dev) - first[pair_of]
    valid = torch.clamp(lengths[pair_of] - in_pair * width, max=width)
    on = torch.arange(width, device=dev)[None, :] < valid[:, None]
    a = _uniform((nb, width), -10000.0, 10000.0, g, dev) * on
    b = _uniform((nb, width), -10000.0, 10000.0, g, dev) * on
    ctl = ((torch.ones_like(valid) << valid) - 1)
    ctl[last] |= 1 << 16
    return a.contiguous(), b.contiguous(), ctl.to(torch.int32).contiguous()
