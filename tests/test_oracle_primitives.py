"""Pins of the CPU oracle's building blocks (no GPU needed).

* FNFlipSign known answers: the only fixed vectors in the reference
  (src/test/scala/raytracer_datapath/FNFlipSign_test.scala:39-56).
* HardFloat add / sub / mul / compare transliteration (oracle A) against the
  host FPU (oracle B) on an edge lattice and on random operands.
* lowMask closed form against a literal restatement of the recursive shifter
  (external/hardfloat/src/main/scala/primitives.scala:45-85).
* Select and sort semantics as the reference's unit tests state them
  (RecFNCompareSelectTest.scala:64-162, QuadSortRecFNTest.scala:93-146).
"""
import numpy as np
import pytest

u32, f32 = np.uint32, np.float32

EDGE = np.array([
    0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x00000002, 0x007FFFFF, 0x807FFFFF, 0x00400000,
    0x00800000, 0x80800000, 0x00800001, 0x00FFFFFF, 0x01000000, 0x7F7FFFFF, 0xFF7FFFFF, 0x7F000000,
    0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00000, 0x7F800001, 0x7FA00000, 0xFFFFFFFF, 0x7FFFFFFF,
    0x3F800000, 0xBF800000, 0x3F800001, 0x3F7FFFFF, 0x34000000, 0x33800000, 0x33800001, 0x4B000000,
    0x4B800000, 0x4B7FFFFF, 0x3FC00000, 0x40400000, 0x3EAAAAAB, 0x41200000, 0x0C000000, 0x73000000,
], dtype=u32)


def test_flip_sign_known_answers(oracle):
    L = oracle.lib()
    assert L.rtl_flip_sign_fn(0x00000000) == 0x80000000
    assert L.rtl_flip_sign_fn(0x80000000) == 0x00000000
    # stage 3 flips the sign of the 33-bit recoded word (UnifiedDatapath.scala:312-319)
    assert L.rtl_flip_sign_via_rec(0x00000000) == 0x80000000
    assert L.rtl_flip_sign_via_rec(0x80000000) == 0x00000000
    rng = np.random.default_rng(3)
    for x in rng.random(10000, dtype=f32):  # FNFlipSign_test.scala:29-35: expected = 0.0f - r
        want = (f32(0.0) - x).view(u32) if x != 0 else u32(0x80000000)
        assert L.rtl_flip_sign_fn(int(x.view(u32))) == int(want)


def test_recfn_roundtrip_is_lossless(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(4)
    for v in list(EDGE) + list(rng.integers(0, 1 << 32, 50000, dtype=np.uint64)):
        assert L.rtl_roundtrip(int(v)) == int(v)
    # canonical encodings of the constants the datapath uses (UnifiedDatapath.scala:25-31)
    assert L.rtl_rec_from_fn(0x7F800000) >> 29 == 0b0110  # sign 0, exponent top bits 110 = infinity
    assert (L.rtl_rec_from_fn(0x00000000) >> 29) & 0x7 == 0  # zero


@pytest.mark.parametrize("op", ["add", "sub", "mul", "cmp"])
def test_hardfloat_units_match_fpu_on_edge_lattice(oracle, op):
    a, b = np.meshgrid(EDGE, EDGE)
    a, b = a.ravel().copy(), b.ravel().copy()
    got, want = oracle.prim("A", op, a, b), oracle.prim("B", op, a, b)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(hex(a[i]), hex(b[i]), hex(got[i]), hex(want[i])) for i in bad[:5]]


def _operand_sets(n, seed):
    rng = np.random.default_rng(seed)
    bits_a = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(u32)
    bits_b = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(u32)
    yield "uniform bit patterns", bits_a, bits_b
    near = (bits_a.astype(np.int64) + rng.integers(-4, 5, n)).astype(u32) ^ (rng.integers(0, 2, n).astype(u32) << 31)
    yield "near-cancellation", bits_a, near
    fa = rng.uniform(-1e4, 1e4, n).astype(f32).view(u32)
    fb = rng.uniform(-1e4, 1e4, n).astype(f32).view(u32)
    yield "scene-range floats", fa, fb
    sa = rng.integers(0, 1 << 24, n, dtype=np.uint64).astype(u32) | (rng.integers(0, 2, n).astype(u32) << 31)
    sb = rng.integers(0, 1 << 25, n, dtype=np.uint64).astype(u32) | (rng.integers(0, 2, n).astype(u32) << 31)
    yield "subnormal neighbourhood", sa, sb
    # products that land in the subnormal range / overflow range
    ea = (rng.integers(1, 80, n).astype(u32) << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint64).astype(u32)
    eb = (rng.integers(1, 80, n).astype(u32) << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint64).astype(u32)
    yield "underflowing products", ea, eb
    ha = (rng.integers(180, 255, n).astype(u32) << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint64).astype(u32)
    hb = (rng.integers(180, 255, n).astype(u32) << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint64).astype(u32)
    yield "overflowing products", ha, hb
    # halfway cases for the adder: b = a * 2^-24 exactly (ties to even)
    ta = (rng.integers(30, 220, n).astype(u32) << 23) | rng.integers(0, 1 << 23, n, dtype=np.uint64).astype(u32)
    tb = (((ta >> 23) - 24) << 23).astype(u32)
    yield "adder ties", ta, tb


@pytest.mark.parametrize("op", ["add", "sub", "mul", "cmp"])
def test_hardfloat_units_match_fpu_on_random_operands(oracle, op):
    for name, a, b in _operand_sets(300000, 11):
        got, want = oracle.prim("A", op, a, b), oracle.prim("B", op, a, b)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (name, [(hex(a[i]), hex(b[i]), hex(got[i]), hex(want[i])) for i in bad[:5]])


def _low_mask_literal(inp, width, top, bottom):
    """lowMask exactly as primitives.scala:45-85 builds it, on Python ints."""
    assert top != bottom
    num = 1 << width
    if top < bottom:
        return _low_mask_literal(~inp & (num - 1), width, num - 1 - top, num - 1 - bottom)
    if num > 64:
        mid = num // 2
        msb = (inp >> (width - 1)) & 1
        lsbs = inp & ((1 << (width - 1)) - 1)
        if mid < top:
            if mid <= bottom:
                return _low_mask_literal(lsbs, width - 1, top - mid, bottom - mid) if msb else 0
            if msb:
                hi = _low_mask_literal(lsbs, width - 1, top - mid, 0)
                return (hi << (mid - bottom)) | ((1 << (mid - bottom)) - 1)
            return _low_mask_literal(lsbs, width - 1, mid, bottom)
        out_w = top - bottom
        inner = _low_mask_literal(lsbs, width - 1, top, bottom)
        return ((1 << out_w) - 1) & ~(0 if msb else (~inner & ((1 << out_w) - 1)))
    shift = (-1 << num) >> inp  # arithmetic shift of an SInt
    hi, lo = num - 1 - bottom, num - top
    field = (shift >> lo) & ((1 << (hi - lo + 1)) - 1)
    w = hi - lo + 1
    return int(format(field, "0%db" % w)[::-1], 2)  # Reverse


def test_low_mask_closed_form(oracle):
    L = oracle.lib()
    for v in range(512):  # the rounder's mask: 9-bit exponent, bounds (105, 130)
        assert L.rtl_low_mask(v, 105, 130) == _low_mask_literal(v, 9, 105, 130), v
    for v in range(8):    # the adder's far-path mask: 3-bit distance, bounds (7, 0)
        assert L.rtl_low_mask(v, 7, 0) == _low_mask_literal(v, 3, 7, 0), v


def test_compare_select_semantics(oracle):
    """c_out = (a > b || (a == b && option)) ? c : d  (RecFNCompareSelect.scala:41-50)."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    vals = rng.uniform(-5e7, 5e7, 20000).astype(f32)
    for a, b in zip(vals[0::2], vals[1::2]):
        ab, bb = int(a.view(u32)), int(b.view(u32))
        assert L.rtl_compare_select_keeps(0, ab, bb) == int(a > b)     # option=false: swap unless a > b
        assert L.rtl_compare_select_keeps(1, ab, bb) == int(a >= b)    # option=true: keep when a >= b
        assert L.rtl_compare_select_keeps(0, ab, ab) == 0              # a == b, option=false: swap
        assert L.rtl_compare_select_keeps(1, ab, ab) == 1              # a == b, option=true: keep
    pz, nz, nan, inf = 0x00000000, 0x80000000, 0x7FC00000, 0x7F800000
    assert L.rtl_compare_select_keeps(1, nz, pz) == 1   # -0 == +0 keeps
    assert L.rtl_compare_select_keeps(0, nz, pz) == 0
    for other in (pz, inf, 0x3F800000, nan):            # NaN on either side: neither gt nor eq -> swap
        assert L.rtl_compare_select_keeps(1, nan, other) == 0
        assert L.rtl_compare_select_keeps(1, other, nan) == 0


def test_quad_sort_orders_largest_first(oracle):
    """QuadSortRecFNTest.scala:93-146: 4-tuples in +-5e7, with +-Inf injected, out(0) largest."""
    L = oracle.lib()
    rng = np.random.default_rng(6)
    for _ in range(20000):
        v = rng.uniform(-5e7, 5e7, 4).astype(f32)
        r = rng.random(4)
        v[r < 0.05] = np.inf
        v[r > 0.95] = -np.inf
        out, idx = np.zeros(4, u32), np.zeros(4, u32)
        L.rtl_quad_sort(v.view(u32).ctypes.data, out.ctypes.data, idx.ctypes.data)
        o = out.view(f32)
        assert (o[:-1] >= o[1:]).all()
        assert sorted(idx.tolist()) == [0, 1, 2, 3]
        assert (v[idx] == o).all()
    # four equal keys: no exchange happens, indices stay 0,1,2,3 (the RTL then reverses to 3,2,1,0)
    v = np.full(4, np.inf, f32)
    out, idx = np.zeros(4, u32), np.zeros(4, u32)
    L.rtl_quad_sort(v.view(u32).ctypes.data, out.ctypes.data, idx.ctypes.data)
    assert idx.tolist() == [0, 1, 2, 3]
