"""No RT_DP_OK with untrusted bits: a hand-over between workgroups (chained beat kernels) or between waves (record
streams, stream_mode 3) that is given up on must either be repaired or reported.

The give-up is forced with the library's test hook: option "debug_chain_skip" = k makes workgroup (wave) k withhold the
running sums it should publish and cuts the successors' bounded wait to 2^12 polls.  Synchronous entry points then
evaluate the beats again without hand-overs and deliver the oracle's bits ("chain_fallbacks" counts it); behind the
asynchronous rt_dp_launch the failure surfaces at rt_dp_synchronize / rt_dp_get_accum as RT_DP_EHIP.  Every case runs
once: nothing here loops over a failing launch.
"""
import numpy as np
import pytest

from raytracer_amd import capi
from oracle import gen

from helpers import assert_bits_equal, hip_angular, hip_euclid

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32


def long_pairs(width, seed):
    """Vector pairs that span several 256-beat tiles, so that the withheld sums are really needed."""
    lengths = np.array([width * 1500, 7, width * 900 + 3, width * 40], np.int64)
    return gen.vec_beats(lengths.size, width, seed=seed, lengths=lengths)[:3]


@pytest.mark.parametrize("device", [True, False])
def test_euclidean_giveup_is_repaired_by_the_synchronous_entry_points(ctx, oracle, torch_cuda, device):
    a, b, ctl = long_pairs(16, 71)
    ctx.set_accum([2.5, 0.0, 0.0])
    ctx.set_option("debug_chain_skip", 1)
    out = hip_euclid(ctx, a, b, ctl, device, torch_cuda)
    want, acc = oracle.euclid_beats(a, b, ctl, 2.5)
    assert_bits_equal(out, want, "euclidean_accumulator behind a forced give-up")
    assert ctx.get_option("vec_chain_giveups") >= 1 and ctx.get_option("chain_fallbacks") == 1
    assert_bits_equal(ctx.get_accum()[0:1], np.array([acc], f32), "euclidean register")
    # with the hook off the same context is healthy again: no new give-up, no new fallback
    ctx.set_option("debug_chain_skip", -1)
    before = ctx.get_option("vec_chain_giveups")
    ctx.set_accum([2.5, 0.0, 0.0])
    assert_bits_equal(hip_euclid(ctx, a, b, ctl, device, torch_cuda), want, "euclidean_accumulator, hook off")
    assert ctx.get_option("vec_chain_giveups") == before and ctx.get_option("chain_fallbacks") == 1


def test_angular_giveup_is_repaired(ctx, oracle, torch_cuda):
    q, c, ctl = long_pairs(8, 72)
    ctx.set_accum([0.0, -1.5, 4.0])
    ctx.set_option("debug_chain_skip", 2)
    dot, norm = hip_angular(ctx, q, c, ctl, True, torch_cuda)
    wd, wn, sd, sn = oracle.angular_beats(q, c, ctl, -1.5, 4.0)
    assert_bits_equal(dot, wd, "angular_dot_product behind a forced give-up")
    assert_bits_equal(norm, wn, "angular_norm behind a forced give-up")
    assert ctx.get_option("chain_fallbacks") == 1
    assert_bits_equal(ctx.get_accum()[1:3], np.array([sd, sn], f32), "angular registers")


def test_asynchronous_launch_reports_the_giveup_at_synchronize(ctx, oracle, torch_cuda):
    torch = torch_cuda
    a, b, ctl = long_pairs(16, 73)
    n = ctl.shape[0]
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    dc = torch.from_numpy(ctl.view(np.int32)).cuda()
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    bt = capi.Batch()
    bt.set_euclid(da, db, dc, out, n)
    torch.cuda.synchronize()
    ctx.set_option("debug_chain_skip", 1)
    ctx.launch(bt)
    with pytest.raises(capi.RtDpError) as e:
        ctx.synchronize()
    assert e.value.code == capi.EHIP and "gave up" in str(e.value)
    ctx.synchronize()                      # reported once
    # the same batch through the shape without hand-overs is the documented way out
    ctx.set_option("debug_chain_skip", -1)
    ctx.set_option("variant", 23)
    ctx.reset()
    ctx.launch(bt)
    ctx.synchronize()
    assert_bits_equal(out.cpu().numpy(), oracle.euclid_beats(a, b, ctl, 0.0)[0], "euclidean_accumulator, owning shape")


@pytest.mark.parametrize("mode", [1, 3])
def test_record_stream_giveups_are_repaired(ctx, oracle, torch_cuda, mode):
    """Mode 1 runs the chained beat kernels on its partitioned segments, mode 3 hands sums from wave to wave."""
    torch = torch_cuda
    rng = np.random.default_rng(75)
    n = 6000
    rec = np.zeros((n, 84), u32)
    rec[:, 0] = rng.integers(2, 4, n)                                     # distance beats only: long chains
    rec[:, 50:82] = (rng.random((n, 32), dtype=f32) * f32(20) - f32(10)).view(u32)
    rec[:, 82] = 0xFFFF
    rec[:, 83] = rng.random(n) < 0.002                                    # a reset every ~500 records
    want = oracle.eval_bundles("B", rec.view(oracle.IN_BUNDLE).reshape(-1), extended=True)[0].view(u32).reshape(-1, 21)
    d_in = torch.from_numpy(rec.view(np.int32)).cuda()
    d_out = torch.empty(n, 21, dtype=torch.int32, device="cuda")
    ref = capi.Context()
    ref.set_option("stream_mode", 2)
    ref.eval_bundles_device(d_in, d_out)
    healthy = d_out.cpu().numpy().view(u32).copy()
    assert_bits_equal(healthy, want, "record stream, mode 2, against the oracle")
    ctx.set_option("stream_mode", mode)
    ctx.set_option("debug_chain_skip", 3)
    d_out.zero_()
    ctx.eval_bundles_device(d_in, d_out)
    assert_bits_equal(d_out.cpu().numpy().view(u32), healthy, "record stream behind a forced give-up, mode %d" % mode)
    assert ctx.get_option("chain_fallbacks") == 1
    assert_bits_equal(ctx.get_accum(), ref.get_accum(), "registers behind the stream")
    ref.close()
