"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front-end of the CPU oracle.

Only tests/, bench.py's ``cpu_baseline`` leg and ``__graft_entry__.smoke()``
may import this module, and only as the checker.  The product
(``raytracer_amd``) never does.

Two restatements live in ``libdp_oracle.so`` (built by oracle/Makefile):
  * ``rtl_*``  -- oracle A, gate-level HardFloat transliteration (hardfloat_rtl.c)
  * ``gold_*`` -- oracle B, native-float scalar model (golden_float.c)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdp_oracle.so")

u32 = np.uint32

F3 = np.dtype([("x", u32), ("y", u32), ("z", u32)])
RAY = np.dtype([("origin", F3), ("dir", F3), ("inv", F3), ("extent", u32),
                ("kx", u32), ("ky", u32), ("kz", u32), ("shear", F3)])
AABB = np.dtype([("x_min", u32), ("y_min", u32), ("z_min", u32),
                 ("x_max", u32), ("y_max", u32), ("z_max", u32)])
TRIANGLE = np.dtype([("A", F3), ("B", F3), ("C", F3)])
IN_BUNDLE = np.dtype([("opcode", u32), ("ray", RAY), ("aabb", AABB, (4,)), ("triangle", TRIANGLE),
                      ("euclidean_a", u32, (16,)), ("euclidean_b", u32, (16,)),
                      ("euclidean_mask", u32), ("euclidean_reset_accum", u32)])
OUT_BUNDLE = np.dtype([("opcode", u32), ("tmin_out", u32, (4,)), ("isIntersect", u32, (4,)),
                       ("boxIndex", u32, (4,)), ("t_num", u32), ("t_denom", u32), ("triangle_hit", u32),
                       ("euclidean_accumulator", u32), ("euclidean_reset_accum", u32),
                       ("angular_dot_product", u32), ("angular_norm", u32), ("angular_reset_accum", u32)])
ACCUM = np.dtype([("euclidean", u32), ("angular_dot", u32), ("angular_norm", u32)])
assert IN_BUNDLE.itemsize == 84 * 4 and OUT_BUNDLE.itemsize == 21 * 4

OP_TRIANGLE, OP_QUADBOX, OP_EUCLIDEAN, OP_ANGULAR = 0, 1, 2, 3

# field order of the SoA box / triangle operand lists
BOX_FIELDS = (["org_x", "org_y", "org_z", "dir_x", "dir_y", "dir_z", "inv_x", "inv_y", "inv_z", "extent"] +
              ["b%d_%s" % (b, f) for b in range(4) for f in ("x_min", "y_min", "z_min", "x_max", "y_max", "z_max")])
TRI_FIELDS = ["org_x", "org_y", "org_z", "shear_x", "shear_y", "shear_z",
              "ax", "ay", "az", "bx", "by", "bz", "cx", "cy", "cz"]


def build(force=False):
    """Compile libdp_oracle.so if missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("hardfloat_rtl.c", "golden_float.c", "dp_bundle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        for name in ("rtl_eval_bundles", "gold_eval_bundles"):
            getattr(L, name).argtypes = [vp, vp, u64, i32, vp]
            getattr(L, name).restype = None
        for name in ("rtl_add_n", "rtl_sub_n", "rtl_mul_n", "rtl_cmp_n", "gold_add_n", "gold_sub_n", "gold_mul_n", "gold_cmp_n"):
            getattr(L, name).argtypes = [vp, vp, vp, u64]
            getattr(L, name).restype = None
        for name in ("rtl_add", "rtl_sub", "rtl_mul", "rtl_cmp"):
            getattr(L, name).argtypes = [C.c_uint32, C.c_uint32]
            getattr(L, name).restype = C.c_uint32
        for name in ("rtl_roundtrip", "rtl_flip_sign_fn", "rtl_flip_sign_via_rec"):
            getattr(L, name).argtypes = [C.c_uint32]
            getattr(L, name).restype = C.c_uint32
        L.rtl_rec_from_fn.argtypes = [C.c_uint32]
        L.rtl_rec_from_fn.restype = C.c_uint64
        L.rtl_fn_from_rec.argtypes = [C.c_uint64]
        L.rtl_fn_from_rec.restype = C.c_uint32
        L.rtl_low_mask.argtypes = [C.c_uint64, C.c_int64, C.c_int64]
        L.rtl_low_mask.restype = C.c_uint64
        L.rtl_quad_sort.argtypes = [vp, vp, vp]
        L.rtl_quad_sort.restype = None
        L.rtl_compare_select_keeps.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.rtl_compare_select_keeps.restype = C.c_uint32
        L.gold_box_soa.argtypes = [u64, vp, vp, vp]
        L.gold_box_soa.restype = None
        L.gold_tri_soa.argtypes = [u64, vp, vp, vp, vp, vp]
        L.gold_tri_soa.restype = None
        L.gold_euclid_beats.argtypes = [u64, vp, vp, vp, vp, vp]
        L.gold_euclid_beats.restype = None
        L.gold_angular_beats.argtypes = [u64, vp, vp, vp, vp, vp, vp, vp]
        L.gold_angular_beats.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _state_arg(state):
    if state is None:
        return None, None
    st = np.zeros(1, ACCUM)
    st["euclidean"], st["angular_dot"], st["angular_norm"] = [int(x) for x in state]
    return st, _p(st)


def eval_bundles(which, bundles, extended=True, state=None):
    """Run oracle ``which`` ('A' or 'B') over an IN_BUNDLE array in stream order.

    ``state`` is (euclid, dot, norm) as uint32 bit patterns or None for power-up.
    Returns (OUT_BUNDLE array, state_out tuple)."""
    bundles = np.ascontiguousarray(bundles, IN_BUNDLE)
    out = np.zeros(bundles.shape[0], OUT_BUNDLE)
    st, stp = _state_arg(state if state is not None else (0, 0, 0))
    fn = lib().rtl_eval_bundles if which == "A" else lib().gold_eval_bundles
    fn(_p(bundles), _p(out), bundles.shape[0], 1 if extended else 0, stp)
    return out, (int(st["euclidean"][0]), int(st["angular_dot"][0]), int(st["angular_norm"][0]))


def prim(which, op, a, b):
    """Element-wise primitive ('add','sub','mul','cmp') on uint32 bit patterns."""
    a = np.ascontiguousarray(a, u32)
    b = np.ascontiguousarray(b, u32)
    o = np.empty_like(a)
    getattr(lib(), ("rtl_" if which == "A" else "gold_") + op + "_n")(_p(a), _p(b), _p(o), a.size)
    return o


def _ptr_array(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def box_soa(fields):
    """Oracle B over SoA box operands. ``fields``: 34 float32 arrays in BOX_FIELDS order.
    Returns (tmin[4] float32 arrays, flags uint32)."""
    fields = [np.ascontiguousarray(f, np.float32) for f in fields]
    n = fields[0].size
    tmin = [np.empty(n, np.float32) for _ in range(4)]
    flags = np.empty(n, u32)
    lib().gold_box_soa(n, _ptr_array(fields), _ptr_array(tmin), _p(flags))
    return tmin, flags


def tri_soa(fields, k):
    """Oracle B over SoA triangle operands (15 arrays in TRI_FIELDS order + packed k)."""
    fields = [np.ascontiguousarray(f, np.float32) for f in fields]
    k = np.ascontiguousarray(k, u32)
    n = k.size
    t_num, t_denom, hit = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(n, u32)
    lib().gold_tri_soa(n, _ptr_array(fields), _p(k), _p(t_num), _p(t_denom), _p(hit))
    return t_num, t_denom, hit


def euclid_beats(a, b, ctl, acc=0.0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ctl = np.ascontiguousarray(ctl, u32)
    out = np.empty(ctl.size, np.float32)
    st = np.array([acc], np.float32)
    lib().gold_euclid_beats(ctl.size, _p(a), _p(b), _p(ctl), _p(out), _p(st))
    return out, st[0]


def angular_beats(q, c, ctl, dot=0.0, norm=0.0):
    q = np.ascontiguousarray(q, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    ctl = np.ascontiguousarray(ctl, u32)
    d, m = np.empty(ctl.size, np.float32), np.empty(ctl.size, np.float32)
    sd, sn = np.array([dot], np.float32), np.array([norm], np.float32)
    lib().gold_angular_beats(ctl.size, _p(q), _p(c), _p(ctl), _p(d), _p(m), _p(sd), _p(sn))
    return d, m, sd[0], sn[0]


def knn_bruteforce(points, queries, k=16, index_base=0):
    """Exact k-NN by brute force with oracle B's Euclidean beat as the distance
    (one beat per query/point pair, mask 0xFFFF, reset 1).  Ranked by (distance, index);
    NaN distances are excluded.  Returns (dist [m, k] float32, idx [m, k] int64), padded
    with (+Inf, 0xFFFFFFFF)."""
    points = np.ascontiguousarray(points, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    n, m = points.shape[0], queries.shape[0]
    out_d = np.full((m, k), np.inf, np.float32)
    out_i = np.full((m, k), 0xFFFFFFFF, np.int64)
    ctl = np.full(n, 0x1FFFF, u32)
    ids = np.arange(n, dtype=np.int64) + index_base
    for q in range(m):
        a = np.broadcast_to(queries[q], (n, 16)).copy()
        d, _ = euclid_beats(a, points, ctl)
        ok = ~np.isnan(d)
        order = np.lexsort((ids[ok], d[ok]))[:k]
        out_d[q, :order.size] = d[ok][order]
        out_i[q, :order.size] = ids[ok][order]
    return out_d, out_i


def knn_bruteforce_angular(points, queries, k=16, index_base=0):
    """Exact top-k by the angular op: per pair, oracle B's two 8-element beats (mask 0xFF, reset on the second)
    give dot and norm; score = dot / sqrt(norm) in float32 (numpy's sqrt and divide are correctly rounded).
    Ranked by (larger score, lower index); NaN scores are excluded.  Returns (minus score [m, k] float32 with -0
    written as +0, idx [m, k] int64), padded with (+Inf, 0xFFFFFFFF)."""
    points = np.ascontiguousarray(points, np.float32)
    queries = np.ascontiguousarray(queries, np.float32)
    n, m = points.shape[0], queries.shape[0]
    out_v = np.full((m, k), np.inf, np.float32)
    out_i = np.full((m, k), 0xFFFFFFFF, np.int64)
    ctl = np.tile(np.array([0xFF, 0x100FF], u32), n)
    ids = np.arange(n, dtype=np.int64) + index_base
    c = points.reshape(2 * n, 8)
    for q in range(m):
        qq = np.tile(queries[q].reshape(2, 8), (n, 1))
        d, nm, _, _ = angular_beats(qq, c, ctl)
        with np.errstate(all="ignore"):
            neg = -(d[1::2] / np.sqrt(nm[1::2])).astype(np.float32)
        neg = np.where(neg == 0, np.float32(0), neg).astype(np.float32)
        ok = ~np.isnan(neg)
        order = np.lexsort((ids[ok], neg[ok]))[:k]
        out_v[q, :order.size] = neg[ok][order]
        out_i[q, :order.size] = ids[ok][order]
    return out_v, out_i
