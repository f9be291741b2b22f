"""Library behaviours around the op kernels: the residency cap as the runtime's occupancy query sees it, the
accumulator calls against batches launched on a caller's stream, argument errors reported before anything runs,
and the per-thread default context of rt_dp_eval."""
import ctypes as C
import threading

import numpy as np
import pytest

from oracle import gen
from raytracer_amd import capi

from helpers import assert_bits_equal

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32


def test_resolved_residency_is_the_default_table(torch_cuda, ctx):
    """Box batches run with 3 resident workgroups per CU, triangle batches with 5 (the measured optima); the cap is
    requested through dynamic LDS, and the occupancy query of the runtime must agree that it took."""
    assert ctx.get_option("box_wgs_per_cu") == 3
    assert ctx.get_option("tri_wgs_per_cu") == 5
    assert ctx.get_option("stream_waves_per_cu") == 8      # the record-stream evaluation kernel's cap took as well
    for w in (3, 4, 6, 8):
        ctx.set_option("wgs_per_cu", w)
        assert ctx.get_option("wgs_per_cu") == w
        assert ctx.get_option("box_wgs_per_cu") == w, "cap %d did not take" % w
    ctx.set_option("wgs_per_cu", 99)
    assert ctx.get_option("box_wgs_per_cu") == 0 and ctx.get_option("tri_wgs_per_cu") == 0
    ctx.set_option("wgs_per_cu", 0)
    assert ctx.get_option("ops_per_lane") == 1 and ctx.get_option("nontemporal") == 1 and ctx.get_option("vec_tile") == 512
    with pytest.raises(capi.RtDpError):
        ctx.get_option("no_such_knob")


def test_accumulator_calls_wait_for_a_callers_stream(ctx, oracle, torch_cuda):
    """rt_dp_launch on a caller stream, then rt_dp_get_accum / rt_dp_set_accum / a batch on the context's own stream
    with no synchronisation in between: each sees the registers the launched batch left."""
    torch = torch_cuda
    a, b, ctl, _ = gen.vec_beats(60000, 16, seed=71)
    ctl = ctl.copy()
    ctl[-1] &= u32(0xFFFF)                       # ends inside a pair: the register is live afterwards
    want, acc = oracle.euclid_beats(a, b, ctl, 0.0)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    dc = torch.from_numpy(ctl.view(np.int32)).cuda()
    out = torch.empty(ctl.shape[0], dtype=torch.float32, device="cuda")
    bt = capi.Batch()
    bt.set_euclid(da, db, dc, out)
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.reset()
        # something slow in front of the batch on the caller's stream, so that an unordered read would come too early
        with torch.cuda.stream(stream):
            x = torch.ones(1 << 26, device="cuda")
            for _i in range(10):
                x = x * 1.0001
        ctx.launch(bt, stream.cuda_stream)
        got = ctx.get_accum()                    # no synchronisation by the caller
        assert_bits_equal(got[0:1], np.array([acc], f32), "register read right behind a launch on a caller stream")
    # a second batch through the synchronous entry point continues from the launched one
    ctx.reset()
    ctx.launch(bt, stream.cuda_stream)
    tail_a, tail_b, tail_ctl, _ = gen.vec_beats(10, 16, seed=72)
    want2, _ = oracle.euclid_beats(tail_a, tail_b, tail_ctl, acc)
    out2 = np.zeros(tail_ctl.shape[0], f32)
    b2 = capi.Batch()
    b2.set_euclid(tail_a, tail_b, tail_ctl, out2)
    ctx.eval(b2)                                 # host arrays, the context's own stream
    assert_bits_equal(out2, want2, "batch on the context stream behind a batch on a caller stream")
    assert_bits_equal(out.cpu().numpy(), want, "the launched batch itself")


def test_misaligned_beat_arrays_are_refused_before_anything_runs(ctx, torch_cuda):
    torch = torch_cuda
    n = 1000
    fields = [torch.from_numpy(np.ascontiguousarray(f)).cuda() for f in gen.box_ops(n, seed=5)]
    tmin = [torch.full((n,), -7.0, dtype=torch.float32, device="cuda") for _ in range(4)]
    flags = torch.full((n,), 0x5A5A, dtype=torch.int32, device="cuda")
    pool = torch.zeros(16 * 64 + 16, dtype=torch.float32, device="cuda")
    b = capi.Batch()
    b.set_box(fields, tmin, flags)
    b.set_euclid(pool[1:1 + 16 * 64].view(64, 16), pool[0:16 * 64].view(64, 16), torch.zeros(64, dtype=torch.int32, device="cuda"),
                 torch.zeros(64, dtype=torch.float32, device="cuda"))     # euc.a is 4 bytes off a 16-byte boundary
    torch.cuda.synchronize()
    with pytest.raises(capi.RtDpError) as e:
        ctx.launch(b)
    assert e.value.code == capi.EALIGN
    ctx.synchronize()
    torch.cuda.synchronize()
    assert bool((flags == 0x5A5A).all()) and bool((tmin[0] == -7.0).all()), "the box segment ran although the batch was refused"


def test_default_context_of_a_thread_is_released_when_the_thread_ends(torch_cuda):
    """rt_dp_eval keeps a per-thread context (8 MiB of device staging once host arrays went through it); threads that
    come and go must not pile these up."""
    torch = torch_cuda
    L = capi.load()
    fields = gen.box_ops(500, seed=9)

    def one_call():
        torch.cuda.set_device(0)
        n = 500
        tmin = [np.zeros(n, f32) for _ in range(4)]
        flags = np.zeros(n, u32)
        b = capi.Batch()
        b.set_box([np.ascontiguousarray(f, f32) for f in fields], tmin, flags)
        b.finalize()
        assert L.rt_dp_eval(C.byref(b.ob), C.byref(b.rb)) == 0

    def run_threads(k):
        for _ in range(k):
            t = threading.Thread(target=one_call)
            t.start()
            t.join()

    run_threads(2)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    run_threads(24)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "24 short-lived threads cost %d MiB of device memory" % ((free0 - free1) >> 20)


def test_a_batch_captured_into_a_graph_replays_with_new_operands_and_carries_the_registers(ctx, oracle, torch_cuda):
    """rt_dp_launch on a stream under capture: the replays read whatever the operand arrays hold by then, and the
    accumulator registers carry replay -> replay -> direct launch -> replay as they would over plain batches."""
    torch = torch_cuda
    n_box, n_beats = 3000, 4000
    rounds = []
    for seed in (81, 82, 83, 84):
        fields = gen.box_ops(n_box, seed=seed)
        a, b, ctl, _ = gen.vec_beats(1500, 16, seed=seed)
        qa, qb, qctl, _ = gen.vec_beats(1500, 8, seed=seed + 100)
        a, b, ctl = a[:n_beats], b[:n_beats], ctl[:n_beats].copy()      # a graph replays one fixed size
        qa, qb, qctl = qa[:n_beats], qb[:n_beats], qctl[:n_beats].copy()
        assert ctl.shape[0] == n_beats and qctl.shape[0] == n_beats
        ctl[-1] &= u32(0xFFFF)                   # both streams end inside a pair: the registers are live afterwards
        qctl[-1] &= u32(0xFFFF)
        ctl[0] &= u32(0xFFFF)                    # ... and start inside one
        qctl[0] &= u32(0xFFFF)
        rounds.append((fields, a, b, ctl, qa, qb, qctl))

    def dev(x):
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()

    d_fields = [dev(f) for f in rounds[0][0]]
    d_a, d_b, d_ctl = dev(rounds[0][1]), dev(rounds[0][2]), dev(rounds[0][3].view(np.int32))
    d_qa, d_qb, d_qctl = dev(rounds[0][4]), dev(rounds[0][5]), dev(rounds[0][6].view(np.int32))
    tmin = [torch.empty(n_box, dtype=torch.float32, device="cuda") for _ in range(4)]
    flags = torch.empty(n_box, dtype=torch.int32, device="cuda")
    acc = torch.empty(n_beats, dtype=torch.float32, device="cuda")
    dot, norm = torch.empty(n_beats, dtype=torch.float32, device="cuda"), torch.empty(n_beats, dtype=torch.float32, device="cuda")
    # the box segment in tiled form (one operand and one result buffer), the beats as plain arrays
    box_in = torch.empty(capi.tiled_words(n_box, 34), dtype=torch.float32, device="cuda")
    box_out = torch.empty(capi.tiled_words(n_box, 5), dtype=torch.int32, device="cuda")
    bt = capi.Batch()
    bt.set_box_tiled(box_in, box_out, n_box)
    bt.set_euclid(d_a, d_b, d_ctl, acc)
    bt.set_angular(d_qa, d_qb, d_qctl, dot, norm)
    bt.finalize()

    def load(r):
        fields, a, b, ctl, qa, qb, qctl = rounds[r]
        for d, f in zip(d_fields, fields):
            d.copy_(dev(f))
        d_a.copy_(dev(a)); d_b.copy_(dev(b)); d_ctl.copy_(dev(ctl.view(np.int32)))
        d_qa.copy_(dev(qa)); d_qb.copy_(dev(qb)); d_qctl.copy_(dev(qctl.view(np.int32)))
        torch.cuda.synchronize()
        ctx.tile_pack(d_fields, box_in, n_box)
        ctx.synchronize()

    state = [f32(0), f32(0), f32(0)]

    def check(r, what):
        fields, a, b, ctl, qa, qb, qctl = rounds[r]
        ctx.tile_unpack(box_out, [t.view(torch.int32) for t in tmin] + [flags], n_box)
        ctx.synchronize()
        want_t, want_f = oracle.box_soa(fields)
        want_acc, state[0] = oracle.euclid_beats(a, b, ctl, state[0])
        want_dot, want_norm, state[1], state[2] = oracle.angular_beats(qa, qb, qctl, state[1], state[2])
        assert_bits_equal(acc.cpu().numpy(), want_acc, what + ": euclidean beats")
        assert_bits_equal(dot.cpu().numpy(), want_dot, what + ": angular dot")
        assert_bits_equal(norm.cpu().numpy(), want_norm, what + ": angular norm")
        assert_bits_equal(flags.cpu().numpy().view(u32), want_f, what + ": box flags")
        for k in range(4):
            assert_bits_equal(tmin[k].cpu().numpy(), want_t[k], what + ": box tmin %d" % k)

    stream = torch.cuda.Stream()
    ctx.reset()
    load(0)
    ctx.launch(bt, stream.cuda_stream)           # direct, also the kernels' one-time set-up
    stream.synchronize()
    check(0, "direct launch before the capture")

    graph = torch.cuda.CUDAGraph()
    load(1)
    with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
        ctx.launch(bt, torch.cuda.current_stream().cuda_stream)
    graph.replay()
    torch.cuda.synchronize()
    check(1, "first replay")
    load(2)
    graph.replay()
    torch.cuda.synchronize()
    check(2, "second replay, new operands")
    load(3)
    ctx.launch(bt, stream.cuda_stream)           # a direct launch between replays
    stream.synchronize()
    check(3, "direct launch between replays")
    got = ctx.get_accum()
    assert_bits_equal(got, np.array(state, f32), "registers after the direct launch")
    load(0)
    graph.replay()
    torch.cuda.synchronize()
    check(0, "replay after a direct launch")


def test_registers_of_an_opcode_survive_batches_without_it(ctx, oracle, torch_cuda):
    """The Euclidean and the angular registers are kept in alternating device copies per opcode: batches with beats of one
    opcode, of the other, of both, and a record stream through rt_dp_eval_bundles_device in between must each continue
    from where the opcode's previous beats stopped."""
    torch = torch_cuda
    from test_gpu_parity import _mixed_stream

    def beats(width, seed, pairs=40):
        a, b, ctl, _ = gen.vec_beats(pairs, width, seed=seed)
        ctl = ctl.copy()
        ctl[-1] &= u32(0xFFFF)          # stops inside a pair: the register stays live
        return a, b, ctl

    state = {"e": f32(0), "d": f32(0), "n": f32(0)}

    def run(kind, seed):
        b = capi.Batch()
        outs = {}
        if "e" in kind:
            a, bb, ctl = beats(16, seed)
            outs["e"] = (np.zeros(ctl.shape[0], f32), a, bb, ctl)
            b.set_euclid(a, bb, ctl, outs["e"][0])
        if "a" in kind:
            q, c, ctl8 = beats(8, seed + 1)
            outs["a"] = (np.zeros(ctl8.shape[0], f32), np.zeros(ctl8.shape[0], f32), q, c, ctl8)
            b.set_angular(q, c, ctl8, outs["a"][0], outs["a"][1])
        ctx.eval(b)
        if "e" in outs:
            got, a, bb, ctl = outs["e"]
            want, state["e"] = oracle.euclid_beats(a, bb, ctl, state["e"])
            assert_bits_equal(got, want, "euclidean beats of batch %r" % kind)
        if "a" in outs:
            gd, gn, q, c, ctl8 = outs["a"]
            wd, wn, state["d"], state["n"] = oracle.angular_beats(q, c, ctl8, state["d"], state["n"])
            assert_bits_equal(gd, wd, "angular dot of batch %r" % kind)
            assert_bits_equal(gn, wn, "angular norm of batch %r" % kind)
        assert_bits_equal(ctx.get_accum(), np.array([state["e"], state["d"], state["n"]], f32), "registers after batch %r" % kind)

    ctx.reset()
    for step, kind in enumerate(["e", "a", "e", "e", "ea", "a", "a", "e"]):
        run(kind, 500 + 10 * step)
    # a record stream in HBM (the fused path folds all three registers in one kernel), then single opcodes again
    stream = _mixed_stream(7, 50, 50, 12, 12)
    stream[np.nonzero(stream[:, 0] == 2)[0][-1], 83] = 0      # both distance opcodes stop inside a pair
    stream[np.nonzero(stream[:, 0] == 3)[0][-1], 83] = 0
    d_in = torch.from_numpy(stream.view(np.int32)).cuda()
    d_out = torch.empty((stream.shape[0], 21), dtype=torch.int32, device="cuda")
    ctx.eval_bundles_device(d_in, d_out)
    c2 = capi.Context()
    c2.set_accum([state["e"], state["d"], state["n"]])
    want = c2.eval_bundles(stream)                             # host-partitioned path of a second context from the same registers
    assert_bits_equal(d_out.cpu().numpy().view(u32), want, "record stream continuing the registers")
    after = c2.get_accum()
    c2.close()
    assert_bits_equal(ctx.get_accum(), after, "registers after the record stream")
    state["e"], state["d"], state["n"] = after
    for step, kind in enumerate(["a", "e", "ea"]):
        run(kind, 700 + 10 * step)
