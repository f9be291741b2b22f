"""Datapath-level pins of the CPU oracle (no GPU needed).

* Hand-derived known answers: every intermediate is exact in binary32, so they
  follow from the dataflow alone (UnifiedDatapath.scala stage functions).
* Oracle A (gate-level HardFloat restatement) == oracle B (native float), bit for bit.
* The properties the reference's own testbench asserts, re-run against an
  independent restatement of its software gold model
  (src/test/scala/raytracer_datapath/Datapath_test.scala:513-612, :247-382).
"""
import numpy as np
import pytest

from oracle import gen

import gold_model

u32, f32 = np.uint32, np.float32


def bits(x):
    return np.array([x], f32).view(u32)[0]


def run_a(oracle, words, **kw):
    return oracle.eval_bundles("A", words.view(oracle.IN_BUNDLE).reshape(-1), **kw)


def default_ray(w):
    """SW_Ray(origin (0,0,0), dir (1,1,1)): inv (1,1,1), extent 4000, kz=0, kx=1, ky=2, shear (1,1,1)
    (SW_Data.scala:36-82, :253)."""
    w[:, 1:4] = bits(0.0)
    w[:, 4:7] = bits(1.0)
    w[:, 7:10] = bits(1.0)
    w[:, 10] = bits(4000.0)
    w[:, 11], w[:, 12], w[:, 13] = 1, 2, 0
    w[:, 14:17] = bits(1.0)


def test_kat_box_all_default_boxes(oracle):
    w = np.zeros((1, 84), u32)
    w[0, 0] = 1
    default_ray(w)
    w[0, 17:41] = 0x7F800000  # four SW_Box(): every face +Inf (SW_Data.scala:106-113)
    for which in "AB":
        out, _ = oracle.eval_bundles(which, w.view(oracle.IN_BUNDLE).reshape(-1))
        assert out["tmin_out"][0].tolist() == [0x7F800000] * 4
        assert out["isIntersect"][0].tolist() == [0, 0, 0, 0]
        assert out["boxIndex"][0].tolist() == [3, 2, 1, 0]  # no exchange on equal keys, then reversed
        assert out["t_num"][0] == 0 and out["triangle_hit"][0] == 0 and out["euclidean_accumulator"][0] == 0


def test_kat_triangle_all_zero(oracle):
    w = np.zeros((1, 84), u32)
    default_ray(w)
    for which in "AB":
        out, _ = oracle.eval_bundles(which, w.view(oracle.IN_BUNDLE).reshape(-1))
        assert out["t_num"][0] == 0 and out["t_denom"][0] == 0 and out["triangle_hit"][0] == 0


def test_kat_euclidean_angular_and_accumulator_carry(oracle):
    w = np.zeros((3, 84), u32)
    default_ray(w)
    w[:, 50:66] = np.arange(16, dtype=f32).view(u32)               # Datapath_test.scala:132
    w[:, 66:82] = np.arange(16, 0, -1).astype(f32).view(u32)       # Datapath_test.scala:133
    for which in "AB":
        w[:, 0], w[:, 82], w[:, 83] = 2, 0xFFFF, [0, 1, 1]
        out, st = oracle.eval_bundles(which, w.view(oracle.IN_BUNDLE).reshape(-1))
        # sum (2i-16)^2 = 1376; second beat doubles it; the third starts again after the reset
        assert out["euclidean_accumulator"].tolist() == [0x44AC0000, 0x452C0000, 0x44AC0000]
        assert out["euclidean_reset_accum"].tolist() == [0, 1, 1] and st == (0, 0, 0)
        w[:, 82] = 0x00FF
        out, _ = oracle.eval_bundles(which, w[:1].view(oracle.IN_BUNDLE).reshape(-1))
        assert out["euclidean_accumulator"][0] == 0x444C0000       # first 8 elements only: 816
        w[:, 0], w[:, 82], w[:, 83] = 3, 0xFF, 1
        out, _ = oracle.eval_bundles(which, w[:1].view(oracle.IN_BUNDLE).reshape(-1))
        assert out["angular_dot_product"][0] == 0x439A0000         # sum i(16-i), i<8 = 308
        assert out["angular_norm"][0] == 0x44A18000                # sum (16-i)^2, i<8 = 1292
        assert out["angular_reset_accum"][0] == 1 and out["euclidean_accumulator"][0] == 0


def test_accumulator_starts_from_plus_zero_and_survives_other_opcodes(oracle):
    """-0 beat sums become +0 through 0 + x; box / triangle records in between leave the registers alone."""
    w = np.zeros((4, 84), u32)
    default_ray(w)
    w[:, 0] = [2, 1, 0, 2]
    w[0, 50], w[0, 66], w[0, 82], w[0, 83] = bits(3.0), bits(1.0), 1, 0     # (3-1)^2 = 4, no reset
    w[3, 50], w[3, 66], w[3, 82], w[3, 83] = bits(5.0), bits(2.0), 1, 1     # + 9 = 13
    w[1, 17:41] = 0x7F800000
    for which in "AB":
        out, st = oracle.eval_bundles(which, w.view(oracle.IN_BUNDLE).reshape(-1))
        assert out["euclidean_accumulator"].tolist() == [bits(4.0), 0, 0, bits(13.0)]
        assert st == (0, 0, 0)


@pytest.mark.parametrize("extended", [True, False])
def test_oracle_a_equals_oracle_b(oracle, extended):
    streams = {
        "box": gen.aos_from_box(gen.box_ops(60000, 1, 0.25)),
        "tri": gen.aos_from_tri(*gen.tri_ops(60000, 2, 0.25)),
        "euc": gen.aos_from_vec(*gen.vec_beats(6000, 16, 3, edge_frac=0.2)[:3], 2),
        "ang": gen.aos_from_vec(*gen.vec_beats(6000, 8, 4, edge_frac=0.2)[:3], 3),
    }
    for name, words in streams.items():
        b = words.view(oracle.IN_BUNDLE).reshape(-1)
        oa, sa = oracle.eval_bundles("A", b, extended=extended)
        ob, sb = oracle.eval_bundles("B", b, extended=extended)
        eq = oa.view(u32).reshape(-1, 21) == ob.view(u32).reshape(-1, 21)
        assert eq.all(), (name, np.nonzero(~eq.all(1))[0][:5])
        assert sa == sb
        if not extended and name in ("euc", "ang"):
            assert not oa.view(u32).reshape(-1, 21)[:, 1:].any()  # baseline datapath: no-ops


def test_soa_and_aos_entry_points_of_oracle_b_agree(oracle):
    fields = gen.box_ops(5000, 5, 0.3)
    tmin, flags = oracle.box_soa(fields)
    out, _ = oracle.eval_bundles("B", gen.aos_from_box(fields).view(oracle.IN_BUNDLE).reshape(-1))
    for k in range(4):
        assert (tmin[k].view(u32) == out["tmin_out"][:, k]).all()
        assert (((flags >> k) & 1) == out["isIntersect"][:, k]).all()
        assert (((flags >> (4 + 2 * k)) & 3) == out["boxIndex"][:, k]).all()
    tf, k = gen.tri_ops(5000, 6, 0.3)
    tn, td, hit = oracle.tri_soa(tf, k)
    out, _ = oracle.eval_bundles("B", gen.aos_from_tri(tf, k).view(oracle.IN_BUNDLE).reshape(-1))
    assert (tn.view(u32) == out["t_num"]).all() and (td.view(u32) == out["t_denom"]).all() and (hit == out["triangle_hit"]).all()
    a, b, ctl, _ = gen.vec_beats(500, 16, 7, edge_frac=0.1)
    acc, _ = oracle.euclid_beats(a, b, ctl)
    out, _ = oracle.eval_bundles("B", gen.aos_from_vec(a, b, ctl, 2).view(oracle.IN_BUNDLE).reshape(-1))
    assert (acc.view(u32) == out["euclidean_accumulator"]).all()
    q, c, ctl8, _ = gen.vec_beats(500, 8, 8, edge_frac=0.1)
    d, m, _, _ = oracle.angular_beats(q, c, ctl8)
    out, _ = oracle.eval_bundles("B", gen.aos_from_vec(q, c, ctl8, 3).view(oracle.IN_BUNDLE).reshape(-1))
    assert (d.view(u32) == out["angular_dot_product"]).all() and (m.view(u32) == out["angular_norm"]).all()


def test_box_properties_of_the_reference_testbench(oracle):
    """check_raybox_result, Datapath_test.scala:513-585, on a seeded random stream."""
    fields = gen.box_ops(20000, seed=20241008)
    out, _ = run_a(oracle, gen.aos_from_box(fields))
    g_tmin, g_hit, _ = gold_model.box_gold(fields)
    t = out["tmin_out"].view(f32)
    isect = out["isIntersect"].astype(bool)
    bidx = out["boxIndex"].astype(np.int64)
    rows = np.arange(t.shape[0])[:, None]
    assert (t[:, 0] >= 0).all() and (t[:, 1:] >= t[:, :-1]).all()             # :550 monotone t
    assert (isect[:, :-1] | ~isect[:, 1:]).all()                                # :553-556 hits precede misses
    assert (g_hit[rows, bidx] == isect).all()                                   # :559-562 same opinion per box
    gt = g_tmin[rows, bidx]
    with np.errstate(all="ignore"):
        err = np.where(t != 0, np.abs(gt - t) / t, gt)
    assert (err[isect] <= 1e-3).all()                                           # :566-576
    # stronger than the reference asks: max / select agree exactly when no NaN is around
    assert (gt[isect] == t[isect]).all()
    assert np.isinf(t[~isect]).all()
    assert (np.sort(bidx, axis=1) == np.arange(4)).all()


def test_triangle_properties_of_the_reference_testbench(oracle):
    """check_raytriangle_result, Datapath_test.scala:587-612; the gold model uses the RTL's own
    operation order (RaytracerGold.scala:115-137), so on hits t_num / t_denom agree bit for bit."""
    fields, k = gen.tri_ops(20000, seed=20241009)
    out, _ = run_a(oracle, gen.aos_from_tri(fields, k))
    g_num, g_den, g_hit = gold_model.tri_gold(fields, k)
    assert (out["triangle_hit"].astype(bool) == g_hit).all()                    # :604
    h = g_hit
    assert h.sum() > 1000
    assert (out["t_num"][h] == g_num[h].view(u32)).all()
    assert (out["t_denom"][h] == g_den[h].view(u32)).all()


def test_distance_ops_against_the_software_sums(oracle):
    """The reference only prints the worst normalised difference (Datapath_test.scala:292-304,
    :358-377); here it is bounded: tree-vs-left-fold reassociation of <= 128 float32 terms."""
    rng = np.random.default_rng(9)
    lengths = rng.integers(1, 129, 200)
    for width, op in ((16, 2), (8, 3)):
        a, b, ctl, _ = gen.vec_beats(lengths.size, width, seed=10 + width, lengths=lengths)
        out, _ = run_a(oracle, gen.aos_from_vec(a, b, ctl, op))
        last = np.nonzero((ctl >> 16) & 1)[0]
        first = np.concatenate([[0], last[:-1] + 1])
        for p, (s, e) in enumerate(zip(first, last)):
            va = a[s:e + 1].ravel()[:lengths[p]]
            vb = b[s:e + 1].ravel()[:lengths[p]]
            if op == 2:
                want = gold_model.calc_diff(va, vb)
                got = out["euclidean_accumulator"][e:e + 1].view(f32)[0]
                assert abs(got - want) <= 1e-4 * abs(want) + 1e-30
            else:
                wd, wn = gold_model.calc_dot(va, vb), gold_model.calc_dot(vb, vb)
                gd = out["angular_dot_product"][e:e + 1].view(f32)[0]
                gn = out["angular_norm"][e:e + 1].view(f32)[0]
                assert abs(gn - wn) <= 1e-4 * abs(wn)
                scale = np.sqrt(float(gold_model.calc_dot(va, va)) * float(wn))  # cancellation-safe bound
                assert abs(gd - wd) <= 1e-4 * scale
            assert out["euclidean_reset_accum"][e] == 1


def test_generated_masks_follow_the_reference_splitter():
    """get_euclidean_job_seq_from_vec_pair (SW_Data.scala:213-269): full beats carry 0xffff, the last
    beat of a pair carries the low-n-bits mask and the reset flag."""
    a, b, ctl, lengths = gen.vec_beats(4, 16, seed=1, lengths=np.array([17, 16, 3, 33]))
    assert ctl.tolist() == [0xFFFF, 0x10001, 0x1FFFF, 0x10007, 0xFFFF, 0xFFFF, 0x10001]
    assert (a[1, 1:] == 0).all() and (b[3, 3:] == 0).all()
    assert gen.generate_mask(3) == 0x0007 and gen.generate_mask(16) == 0xFFFF
    _, _, ctl8, _ = gen.vec_beats(2, 8, seed=1, lengths=np.array([9, 8]))
    assert ctl8.tolist() == [0xFF, 0x10001, 0x100FF]


def test_ray_derivation_follows_sw_ray():
    """SW_Ray (SW_Data.scala:36-82): strict-> tie rule of max_dim, kx/ky swap on negative dir[kz]."""
    d = np.array([[1, 1, 1], [1, 2, 2], [-3, 1, 2], [1, -5, 2], [0.5, 0.25, -4], [2, -2, 1]], f32)
    inv, kx, ky, kz, shear = gen.ray_derived(d)
    assert kz.tolist() == [0, 1, 0, 1, 2, 0]
    assert kx.tolist() == [1, 2, 2, 0, 1, 1]
    assert ky.tolist() == [2, 0, 1, 2, 0, 2]
    assert np.allclose(inv, 1.0 / d)
    assert shear[4].tolist() == [f32(0.25) / f32(-4), f32(0.5) / f32(-4), f32(1) / f32(-4)]
