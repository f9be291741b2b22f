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


# special operand values injected by the "special operand" edge class: signed zeros, infinities, quiet / signalling
# NaNs with payloads, smallest / largest subnormals, smallest normals, largest finite values, +-1
_SPECIALS = [0x00000000, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00000, 0x7F800001, 0xFFFFFFFF, 0x00000001,
             0x80000001, 0x007FFFFF, 0x807FFFFF, 0x00800000, 0x80800000, 0x7F7FFFFF, 0xFF7FFFFF, 0x3F800000, 0xBF800000]

BOX_EDGE_CLASSES = ["dir component +-0 (inv = +-Inf)", "default all-+Inf box", "duplicate box (sort tie)", "box face through the ray origin",
                    "inverted (empty) box", "ray origin inside a box", "subnormal-scaled scene", "four identical boxes",
                    "NaN / Inf / subnormal operand"]
TRI_EDGE_CLASSES = ["ray through a vertex", "ray through an edge midpoint", "repeated vertex", "collinear vertices",
                    "triangle behind the origin", "subnormal-scaled scene", "kz encoded as 3", "NaN / Inf / subnormal operand"]


def _classes(n, n_classes, per_class, g, dev):
    """Class number per op: class c with probability per_class each, -1 (plain random op) otherwise."""
    u = torch.randint(0, 1 << 20, (n,), generator=g, device=dev)
    width = max(1, int(per_class * (1 << 20)))
    cls = torch.div(u, width, rounding_mode="floor")
    return torch.where(cls < n_classes, cls, torch.full_like(cls, -1))


def _inject_specials(fields, rows, g, dev):
    """One randomly chosen operand of every op in `rows` (bool mask) becomes a special value."""
    n = fields[0].shape[0]
    which = torch.randint(0, len(fields), (n,), generator=g, device=dev)
    table = torch.tensor(_SPECIALS, dtype=torch.int64, device=dev).to(torch.int32).view(torch.float32)
    vals = table[torch.randint(0, len(_SPECIALS), (n,), generator=g, device=dev)]
    return [torch.where(rows & (which == k), vals, f) for k, f in enumerate(fields)]


def box_stream(n, seed, dev, edge_per_class=0.0):
    """34 float32 device arrays in capi.BOX_FIELDS order.  edge_per_class > 0 turns that fraction of the ops into
    each of BOX_EDGE_CLASSES (the classes of the host generator of the test suite, drawn on the device)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    org = [_uniform((n,), -100.0, 100.0, g, dev) for _ in range(3)]
    dirs = [_uniform((n,), -100.0, 100.0, g, dev) for _ in range(3)]
    lo, hi = [], []
    for _ in range(4):
        l3, h3 = [], []
        for _a in range(3):
            r0 = _next_float((n,), g, dev) * 1e4 * 2 - 1e4
            r1 = _next_float((n,), g, dev) * 1e4 * 2 - 1e4
            l3.append(torch.minimum(r0, r1))
            h3.append(torch.maximum(r0, r1))
        lo.append(l3)
        hi.append(h3)
    cls = None
    if edge_per_class > 0:
        cls = _classes(n, len(BOX_EDGE_CLASSES), edge_per_class, g, dev)
        axis = torch.randint(0, 3, (n,), generator=g, device=dev)
        box = torch.randint(0, 4, (n,), generator=g, device=dev)
        zero = torch.where(torch.randint(0, 2, (n,), generator=g, device=dev) == 0, 0.0, -0.0).to(torch.float32)
        inf = torch.full((n,), float("inf"), device=dev)
        for a in range(3):
            dirs[a] = torch.where((cls == 0) & (axis == a), zero, dirs[a])                    # 0 * Inf = NaN inside the op
            lo[0][a] = torch.where((cls == 3) & (axis == a), org[a], lo[0][a])                # t = +-0 on that face
        for b in range(4):
            for a in range(3):
                m = (cls == 1) & (box == b)
                lo[b][a] = torch.where(m, inf, lo[b][a])
                hi[b][a] = torch.where(m, inf, hi[b][a])
        for b in range(1, 4):
            for a in range(3):
                m = ((cls == 2) & (box == b)) | (cls == 7)
                lo[b][a] = torch.where(m, lo[0][a], lo[b][a])
                hi[b][a] = torch.where(m, hi[0][a], hi[b][a])
        for a in range(3):
            m = cls == 4
            lo[1][a], hi[1][a] = torch.where(m, hi[1][a], lo[1][a]), torch.where(m, lo[1][a], hi[1][a])
            m = cls == 5
            lo[2][a] = torch.where(m, org[a] - 5.0, lo[2][a])
            hi[2][a] = torch.where(m, org[a] + 5.0, hi[2][a])
        scale = torch.where(cls == 6, 1e-41, 1.0).to(torch.float32)
        org = [o * scale for o in org]
        lo = [[x * scale for x in l3] for l3 in lo]
        hi = [[x * scale for x in h3] for h3 in hi]
    fields = org + dirs + [1.0 / d for d in dirs]                                          # inv = 1.0f / dir
    fields.append(torch.full((n,), 4000.0, device=dev, dtype=torch.float32))               # extent
    for b in range(4):
        fields += lo[b] + hi[b]
    if cls is not None:
        fields = _inject_specials(fields, cls == 8, g, dev)
    return [f.contiguous() for f in fields]


def tri_stream(n, seed, dev, edge_per_class=0.0):
    """(15 float32 device arrays in capi.TRI_FIELDS order, packed k as int32).  edge_per_class > 0 turns that
    fraction of the ops into each of TRI_EDGE_CLASSES."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    vtx = [[_uniform((n,), -1000.0, 1000.0, g, dev) for _a in range(3)] for _v in range(3)]
    org = [_uniform((n,), -1000.0, 1000.0, g, dev) for _a in range(3)]
    cen = [((vtx[0][a] + vtx[1][a]) + vtx[2][a]) / 3.0 for a in range(3)]
    target = list(cen)
    cls = None
    if edge_per_class > 0:
        cls = _classes(n, len(TRI_EDGE_CLASSES), edge_per_class, g, dev)
        corner = torch.randint(0, 3, (n,), generator=g, device=dev)
        for a in range(3):
            for v in range(3):
                target[a] = torch.where((cls == 0) & (corner == v), vtx[v][a], target[a])     # grazing: U / V / W at or near zero
            target[a] = torch.where(cls == 1, (vtx[0][a] + vtx[1][a]) * 0.5, target[a])
            vtx[2][a] = torch.where(cls == 2, vtx[1][a], vtx[2][a])                           # t_denom == 0
            vtx[2][a] = torch.where(cls == 3, vtx[0][a] + (vtx[1][a] - vtx[0][a]) * 2.0, vtx[2][a])
            target[a] = torch.where(cls == 4, org[a] - (cen[a] - org[a]), target[a])          # t_num < 0
    d = [target[a] - org[a] for a in range(3)]
    if cls is not None:
        scale = torch.where(cls == 5, 1e-41, 1.0).to(torch.float32)
        vtx = [[x * scale for x in v3] for v3 in vtx]
        org = [o * scale for o in org]
        d = [x * scale for x in d]
        # kz == 3 reads z like 2 (Float3.at): make z the dominant axis where the encoding is used
        big = torch.maximum(torch.maximum(d[0].abs(), d[1].abs()), d[2].abs()) + 1.0
        d[2] = torch.where(cls == 6, torch.where(d[2] >= 0, big, -big), d[2])
    d = torch.stack(d, dim=1)
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
    if cls is not None:
        kz = torch.where((cls == 6) & (kz == 2), torch.full_like(kz, 3), kz)
    k = (kx | (ky << 2) | (kz << 4)).to(torch.int32)
    fields = org + shear + vtx[0] + vtx[1] + vtx[2]
    if cls is not None:
        fields = _inject_specials(fields, cls == 7, g, dev)
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
