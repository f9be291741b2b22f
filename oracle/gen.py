"""Deterministic synthetic operand streams for the four datapath opcodes, and the float32 restatement of the
reference's ray derivation (ray_derived) that the ray pre-processing kernels are checked against.

TEST INFRASTRUCTURE, like everything under oracle/: imported by tests/, the fixture generator and the smoke
check only, never by the product package.

Distributions follow the reference's own random generators
(src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:310-417 and the
call sites in Datapath_test.scala:142-225); the reference draws from an unseeded
java.util.Random, here every stream is a pure function of its seed.  Edge-case
classes (signed zeros, NaN / Inf / subnormal operands, ties, degenerate
geometry) can be mixed in at a chosen fraction.

All arithmetic that derives ray fields is done in float32 like the Scala `Float`
code it mirrors (SW_Ray, SW_Data.scala:32-88).
"""
import numpy as np

f32 = np.float32
u32 = np.uint32

SCENE_BOUNDS = 1000.0   # SW_Data.scala:91
RAY_EXTENT = 4000.0     # SW_Data.scala:92


def _next_float(rng, shape):
    """java.util.Random.nextFloat(): a multiple of 2^-24 in [0, 1)."""
    return rng.integers(0, 1 << 24, size=shape, dtype=np.uint32).astype(f32) * f32(2.0 ** -24)


def _uniform(rng, shape, lo, hi):
    # r.nextFloat() * (upper - lower) + lower, in float32 (SW_Data.scala:361, :380, :400)
    return _next_float(rng, shape) * f32(hi - lo) + f32(lo)


def ray_derived(dirs):
    """inv, (kx, ky, kz), shear of SW_Ray from dir[n, 3] (SW_Data.scala:36-82)."""
    d = np.asarray(dirs, f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = f32(1.0) / d
        ax, ay, az = np.abs(d[:, 0]), np.abs(d[:, 1]), np.abs(d[:, 2])
        kz = np.zeros(d.shape[0], np.int64)
        mval = ax.copy()
        upd = ay > mval                      # strict >, ties keep the lower axis (:44-47)
        kz[upd] = 1
        mval = np.where(upd, ay, mval)
        kz[az > mval] = 2                    # (:48-50)
        kx = (kz + 1) % 3
        ky = (kx + 1) % 3
        idx = np.arange(d.shape[0])
        swap = d[idx, kz] < 0                # (:70-74)
        kx, ky = np.where(swap, ky, kx), np.where(swap, kx, ky)
        dz = d[idx, kz]
        shear = np.stack([d[idx, kx] / dz, d[idx, ky] / dz, f32(1.0) / dz], axis=1).astype(f32)
    return inv.astype(f32), kx.astype(u32), ky.astype(u32), kz.astype(u32), shear


def pack_k(kx, ky, kz):
    return (kx.astype(u32) | (ky.astype(u32) << 2) | (kz.astype(u32) << 4)).astype(u32)


_SPECIALS = np.array([0x00000000, 0x80000000, 0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00000, 0x7F800001,
                      0xFFFFFFFF, 0x00000001, 0x80000001, 0x007FFFFF, 0x807FFFFF, 0x00800000, 0x80800000,
                      0x7F7FFFFF, 0xFF7FFFFF, 0x3F800000, 0xBF800000], dtype=u32).view(f32)


def _inject_specials(rng, cols, rows):
    """Overwrite a random operand of each listed row with a special value."""
    if rows.size == 0:
        return
    which = rng.integers(0, len(cols), rows.size)
    vals = _SPECIALS[rng.integers(0, _SPECIALS.size, rows.size)]
    for c in range(len(cols)):
        sel = rows[which == c]
        cols[c][sel] = vals[which == c]


def box_ops(n, seed=20241008, edge_frac=0.0):
    """n Quadbox operand records as 34 float32 arrays (order: capi.BOX_FIELDS).

    genRandomRay(1e2) + 4 x genRandomBox(1e4): Datapath_test.scala:142-147,
    SW_Data.scala:315-348."""
    rng = np.random.default_rng(seed)
    org = _uniform(rng, (n, 3), -100.0, 100.0)
    dr = _uniform(rng, (n, 3), -100.0, 100.0)
    raw = _next_float(rng, (n, 4, 3, 2)) * f32(1e4) * f32(2) - f32(1e4)  # r.nextFloat() * range * 2 - range
    lo = np.minimum(raw[..., 0], raw[..., 1])
    hi = np.maximum(raw[..., 0], raw[..., 1])
    extent = np.full(n, RAY_EXTENT, f32)

    n_edge = int(n * edge_frac)
    if n_edge:
        rows = rng.choice(n, n_edge, replace=False)
        cls = rng.integers(0, 8, n_edge)
        r = rows[cls == 0]  # a direction component is +-0 (inv = +-Inf, 0 * Inf = NaN inside)
        dr[r, rng.integers(0, 3, r.size)] = np.where(rng.integers(0, 2, r.size) == 0, f32(0.0), f32(-0.0))
        r = rows[cls == 1]  # default SW_Box(): every face at +Inf (SW_Data.scala:106-113)
        b = rng.integers(0, 4, r.size)
        lo[r, b, :] = np.inf
        hi[r, b, :] = np.inf
        r = rows[cls == 2]  # duplicate boxes: ties in the sort network
        b = rng.integers(1, 4, r.size)
        lo[r, b, :] = lo[r, 0, :]
        hi[r, b, :] = hi[r, 0, :]
        r = rows[cls == 3]  # a box face through the ray origin (t = +-0)
        a = rng.integers(0, 3, r.size)
        lo[r, 0, a] = org[r, a]
        r = rows[cls == 4]  # inverted (empty) box
        lo[r, 1, :], hi[r, 1, :] = hi[r, 1, :].copy(), lo[r, 1, :].copy()
        r = rows[cls == 5]  # origin inside box 2: tmin clamps to +0
        lo[r, 2, :] = org[r, :] - f32(5.0)
        hi[r, 2, :] = org[r, :] + f32(5.0)
        r = rows[cls == 6]  # subnormal-scale scene
        s = f32(1e-41)
        org[r] *= s
        lo[r] *= s
        hi[r] *= s
        r = rows[cls == 7]  # all four boxes identical (full tie, index order 3,2,1,0 among equals)
        for b in range(1, 4):
            lo[r, b, :] = lo[r, 0, :]
            hi[r, b, :] = hi[r, 0, :]
    inv, _, _, _, _ = ray_derived(dr)

    fields = [org[:, 0], org[:, 1], org[:, 2], dr[:, 0], dr[:, 1], dr[:, 2], inv[:, 0], inv[:, 1], inv[:, 2], extent]
    for b in range(4):
        fields += [lo[:, b, 0], lo[:, b, 1], lo[:, b, 2], hi[:, b, 0], hi[:, b, 1], hi[:, b, 2]]
    fields = [np.ascontiguousarray(f, f32) for f in fields]
    if n_edge:
        # raw special operands anywhere, bypassing the derived-field relations
        rows2 = rng.choice(n, max(1, n_edge // 4), replace=False)
        _inject_specials(rng, fields, rows2)
    return fields


def tri_ops(n, seed=20241009, edge_frac=0.0):
    """n Triangle operand records: (15 float32 arrays in capi.TRI_FIELDS order, packed k).

    genRandomTriangle(-1000, 1000) + genRandomRayGivenPoint(centroid, -1000, 1000):
    Datapath_test.scala:162-176, SW_Data.scala:350-388, centroid :120-124."""
    rng = np.random.default_rng(seed)
    vtx = _uniform(rng, (n, 3, 3), -SCENE_BOUNDS, SCENE_BOUNDS)  # [op, vertex, axis]
    org = _uniform(rng, (n, 3), -SCENE_BOUNDS, SCENE_BOUNDS)
    cen = ((vtx[:, 0, :] + vtx[:, 1, :]) + vtx[:, 2, :]) / f32(3.0)
    target = cen.copy()

    n_edge = int(n * edge_frac)
    kz3 = np.zeros(n, bool)
    scale = np.ones(n, f32)
    if n_edge:
        rows = rng.choice(n, n_edge, replace=False)
        cls = rng.integers(0, 7, n_edge)
        r = rows[cls == 0]  # aim exactly at a vertex (grazing: two of U, V, W near / at zero)
        target[r] = vtx[r, rng.integers(0, 3, r.size), :]
        r = rows[cls == 1]  # aim at an edge midpoint
        target[r] = (vtx[r, 0, :] + vtx[r, 1, :]) * f32(0.5)
        r = rows[cls == 2]  # degenerate: repeated vertex
        vtx[r, 2, :] = vtx[r, 1, :]
        r = rows[cls == 3]  # degenerate: collinear
        vtx[r, 2, :] = vtx[r, 0, :] + (vtx[r, 1, :] - vtx[r, 0, :]) * f32(2.0)
        r = rows[cls == 4]  # triangle behind the origin: reverse the ray
        target[r] = org[r] - (cen[r] - org[r])
        r = rows[cls == 5]  # subnormal-scale scene
        scale[r] = f32(1e-41)
        r = rows[cls == 6]  # kz encoded as 3 (reads z, like 2)
        kz3[r] = True
    dr = (target - org).astype(f32)
    vtx = vtx * scale[:, None, None]
    org = org * scale[:, None]
    dr = dr * scale[:, None]
    if n_edge:
        r = rows[cls == 6]
        # make z the dominant axis so that kz == 2 and the 3-encoding is meaningful
        dr[r, 2] = np.where(dr[r, 2] >= 0, f32(1.0), f32(-1.0)) * (np.abs(dr[r]).max(axis=1) + f32(1.0))
    _, kx, ky, kz, shear = ray_derived(dr)
    kz = np.where(kz3 & (kz == 2), u32(3), kz).astype(u32)

    fields = [org[:, 0], org[:, 1], org[:, 2], shear[:, 0], shear[:, 1], shear[:, 2]]
    for v in range(3):
        fields += [vtx[:, v, 0], vtx[:, v, 1], vtx[:, v, 2]]
    fields = [np.ascontiguousarray(f, f32) for f in fields]
    if n_edge:
        rows2 = rng.choice(n, max(1, n_edge // 4), replace=False)
        _inject_specials(rng, fields, rows2)
    return fields, pack_k(kx, ky, kz)


def generate_mask(n):
    """Low n bits set (get_euclidean_job_seq_from_vec_pair.generate_mask, SW_Data.scala:218-231)."""
    return (1 << n) - 1


def vec_beats(n_pairs, width, seed=20241010, max_len=128, lo=-10000.0, hi=10000.0, lengths=None, edge_frac=0.0):
    """Vector pairs split into beats of `width` elements (16 Euclidean, 8 angular).

    genRandomVectorPair(-10000, 10000, 16 * 8): Datapath_test.scala:219-222,
    SW_Data.scala:405-416; beat split, zero padding, mask and last-beat reset:
    SW_Data.scala:233-268 (Euclidean), :271-308 (angular).

    Returns a [beats, width], b [beats, width], ctl [beats], lengths [n_pairs]."""
    rng = np.random.default_rng(seed)
    if lengths is None:
        lengths = rng.integers(0, max_len, n_pairs) + 1  # r.nextInt(largest_length) + 1
    lengths = np.asarray(lengths, np.int64)
    beats_per = (lengths + width - 1) // width
    nb = int(beats_per.sum())
    a = np.zeros((nb, width), f32)
    b = np.zeros((nb, width), f32)
    ctl = np.zeros(nb, u32)
    last = np.cumsum(beats_per) - 1
    first = last - beats_per + 1
    # element validity per beat
    beat_pair = np.repeat(np.arange(n_pairs), beats_per)
    beat_in_pair = np.arange(nb) - first[beat_pair]
    remaining = lengths[beat_pair] - beat_in_pair * width
    valid = np.minimum(remaining, width)
    col = np.arange(width)[None, :]
    on = col < valid[:, None]
    va = _uniform(rng, (nb, width), lo, hi)
    vb = _uniform(rng, (nb, width), lo, hi)
    a[on] = va[on]
    b[on] = vb[on]
    ctl[:] = ((1 << valid) - 1).astype(u32)
    ctl[last] |= u32(1 << 16)
    n_edge = int(nb * edge_frac)
    if n_edge:
        rows = rng.choice(nb, n_edge, replace=False)
        cls = rng.integers(0, 4, n_edge)
        r = rows[cls == 0]  # special values in live or masked-off lanes
        a[r, rng.integers(0, width, r.size)] = _SPECIALS[rng.integers(0, _SPECIALS.size, r.size)]
        r = rows[cls == 1]
        b[r, rng.integers(0, width, r.size)] = _SPECIALS[rng.integers(0, _SPECIALS.size, r.size)]
        r = rows[cls == 2]  # ragged masks (holes), which the reference's splitter never produces
        ctl[r] = (ctl[r] & u32(1 << 16)) | rng.integers(0, 1 << width, r.size).astype(u32)
        r = rows[cls == 3]  # subnormal-scale beat
        a[r] *= f32(1e-42)
        b[r] *= f32(1e-42)
    return a, b, ctl, lengths


def aos_from_box(fields):
    """Box SoA -> structured array of rt_dp_input_bundle images (uint32 words [n, 84])."""
    n = fields[0].shape[0]
    w = np.zeros((n, 84), u32)
    w[:, 0] = 1
    fb = [np.ascontiguousarray(f, f32).view(u32) for f in fields]
    # ray: origin 1..3, dir 4..6, inv 7..9, extent 10, kx ky kz 11..13, shear 14..16
    for i in range(9):
        w[:, 1 + i] = fb[i]
    w[:, 10] = fb[9]
    for b in range(4):
        for f in range(6):
            w[:, 17 + 6 * b + f] = fb[10 + 6 * b + f]
    return w


def aos_from_tri(fields, k):
    n = k.shape[0]
    w = np.zeros((n, 84), u32)
    fb = [np.ascontiguousarray(f, f32).view(u32) for f in fields]
    w[:, 0] = 0
    w[:, 1:4] = np.stack(fb[0:3], 1)
    w[:, 11] = k & 3
    w[:, 12] = (k >> 2) & 3
    w[:, 13] = (k >> 4) & 3
    w[:, 14:17] = np.stack(fb[3:6], 1)
    w[:, 41:50] = np.stack(fb[6:15], 1)
    return w


def aos_from_vec(a, b, ctl, opcode):
    n = ctl.shape[0]
    width = a.shape[1]
    w = np.zeros((n, 84), u32)
    w[:, 0] = opcode
    w[:, 50:50 + width] = np.ascontiguousarray(a, f32).view(u32)
    w[:, 66:66 + width] = np.ascontiguousarray(b, f32).view(u32)
    w[:, 82] = ctl & 0xFFFF
    w[:, 83] = (ctl >> 16) & 1
    return w
