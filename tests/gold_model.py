"""Independent float32 restatement of the reference's software gold model, used only
to re-run the properties the reference's own testbench asserts.

Follows src/test/scala/raytracer_datapath/raytracer_gold/RaytracerGold.scala:
  testIntersection (one box)  :19-65,  four boxes + stable sort by t  :68-89,
  testTriangleIntersection    :91-147,
and SW_Vector.calc_diff / calc_dot_product (SW_Data.scala:136-158).
Valid for NaN-free inputs only (scala.math.max/min propagate NaN, the RTL's
selects do not); the reference's random tests never produce NaN operands.
"""
import numpy as np

f32 = np.float32


def box_gold(fields):
    """fields: 34 arrays (capi.BOX_FIELDS order).  Returns per-box tmin [n,4] (+Inf on miss), hit [n,4],
    and the gold ordering (stable sort by t) as indices [n,4]."""
    org = np.stack(fields[0:3], 1).astype(f32)
    dr = np.stack(fields[3:6], 1).astype(f32)
    inv = np.stack(fields[6:9], 1).astype(f32)
    ext = fields[9].astype(f32)
    n = org.shape[0]
    tmin_all = np.empty((n, 4), f32)
    hit_all = np.empty((n, 4), bool)
    with np.errstate(all="ignore"):
        for b in range(4):
            lo = np.stack(fields[10 + 6 * b: 13 + 6 * b], 1).astype(f32)
            hi = np.stack(fields[13 + 6 * b: 16 + 6 * b], 1).astype(f32)
            tp_min = ((lo - org).astype(f32) * inv).astype(f32)
            tp_max = ((hi - org).astype(f32) * inv).astype(f32)
            pos = dr >= 0
            t_min = np.where(pos, tp_min, tp_max)
            t_max = np.where(pos, tp_max, tp_min)
            tmin = np.maximum(np.maximum(np.maximum(t_min[:, 0], t_min[:, 1]), t_min[:, 2]), f32(0.0))
            tmax = np.minimum(np.minimum(np.minimum(t_max[:, 0], t_max[:, 1]), t_max[:, 2]), ext)
            hit = tmin < tmax
            tmin_all[:, b] = np.where(hit, tmin, f32(np.inf))
            hit_all[:, b] = hit
    order = np.argsort(tmin_all, axis=1, kind="stable")
    return tmin_all, hit_all, order


def tri_gold(fields, k):
    """Returns (t_num, t_denom, hit) with the gold model's early-outs (fields left at 0)."""
    org = np.stack(fields[0:3], 1).astype(f32)
    sh = np.stack(fields[3:6], 1).astype(f32)
    kx, ky, kz = (k & 3).astype(np.int64), ((k >> 2) & 3).astype(np.int64), ((k >> 4) & 3).astype(np.int64)
    kz = np.minimum(kz, 2)
    idx = np.arange(org.shape[0])
    X, Y, Z = [], [], []
    with np.errstate(all="ignore"):
        for v in range(3):
            P = (np.stack(fields[6 + 3 * v: 9 + 3 * v], 1).astype(f32) - org).astype(f32)
            pz = P[idx, kz]
            X.append((P[idx, kx] - (sh[:, 0] * pz).astype(f32)).astype(f32))
            Y.append((P[idx, ky] - (sh[:, 1] * pz).astype(f32)).astype(f32))
            Z.append((sh[:, 2] * pz).astype(f32))
        A, B, C = 0, 1, 2
        U = ((X[C] * Y[B]).astype(f32) - (Y[C] * X[B]).astype(f32)).astype(f32)
        V = ((X[A] * Y[C]).astype(f32) - (Y[A] * X[C]).astype(f32)).astype(f32)
        W = ((X[B] * Y[A]).astype(f32) - (Y[B] * X[A]).astype(f32)).astype(f32)
        inside = ~((U < 0) | (V < 0) | (W < 0))
        den = ((U + V).astype(f32) + W).astype(f32)
        ok_den = inside & (den != 0)
        num = (((U * Z[A]).astype(f32) + (V * Z[B]).astype(f32)).astype(f32) + (W * Z[C]).astype(f32)).astype(f32)
        hit = ok_den & ~(num < 0)
    t_denom = np.where(inside, den, f32(0))
    t_num = np.where(ok_den, num, f32(0))
    return t_num, t_denom, hit


def calc_diff(a, b):
    """sum of (a-b)*(a-b), left fold in float32 (SW_Data.scala:136-144)."""
    acc = None
    for x, y in zip(np.asarray(a, f32), np.asarray(b, f32)):
        d = f32(x - y)
        t = f32(d * d)
        acc = t if acc is None else f32(acc + t)
    return acc


def calc_dot(a, b):
    acc = None
    for x, y in zip(np.asarray(a, f32), np.asarray(b, f32)):
        t = f32(x * y)
        acc = t if acc is None else f32(acc + t)
    return acc
