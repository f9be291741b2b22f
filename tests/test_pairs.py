"""Vector pairs <-> beats: rt_dp_beat_offsets_host against the splitter's beat counts on CPU; on the GPU the split
equals the Python restatement of the reference's job splitters (oracle.gen.vec_beats, SW_Data.scala:213-308)
word for word, and pairs -> beats -> distance op -> pair values equals the oracle run on that restatement."""
import numpy as np
import pytest

from raytracer_amd import capi
from oracle import gen

f32, u32 = np.float32, np.uint32
LENGTHS = np.array([1, 16, 17, 0, 8, 9, 128, 31, 32, 33, 0, 0, 5, 100, 64, 2], np.int64)


def ragged(lengths, seed):
    rng = np.random.default_rng(seed)
    total = int(lengths.sum())
    a = rng.uniform(-1e4, 1e4, total).astype(f32)
    b = rng.uniform(-1e4, 1e4, total).astype(f32)
    off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    return a, b, off


def reference_split(a, b, off, width):
    """Beat rows as the reference's splitter lays them out (via gen.vec_beats' layout rules)."""
    rows_a, rows_b, ctl = [], [], []
    for p in range(off.shape[0] - 1):
        va, vb = a[int(off[p]):int(off[p + 1])], b[int(off[p]):int(off[p + 1])]
        for s in range(0, va.shape[0], width):
            ca, cb = va[s:s + width], vb[s:s + width]
            rows_a.append(np.pad(ca, (0, width - ca.shape[0])))
            rows_b.append(np.pad(cb, (0, width - cb.shape[0])))
            ctl.append(gen.generate_mask(ca.shape[0]) | ((1 << 16) if s + width >= va.shape[0] else 0))
    shape = (len(ctl), width)
    return (np.array(rows_a, f32).reshape(shape), np.array(rows_b, f32).reshape(shape), np.array(ctl, u32))


@pytest.mark.parametrize("width", [16, 8])
def test_beat_offsets_follow_the_splitter(width):
    L = capi.load()
    off = np.concatenate([[0], np.cumsum(LENGTHS)]).astype(np.uint64)
    bo = np.zeros(off.shape[0], np.uint64)
    n = L.rt_dp_beat_offsets_host(off.ctypes.data, LENGTHS.shape[0], width, bo.ctypes.data)
    per = (LENGTHS + width - 1) // width
    assert n == per.sum() and (bo == np.concatenate([[0], np.cumsum(per)])).all()
    assert L.rt_dp_beat_offsets_host(off.ctypes.data, LENGTHS.shape[0], width, None) == n
    assert L.rt_dp_beat_offsets_host(off.ctypes.data, LENGTHS.shape[0], 12, None) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("width", [16, 8])
@pytest.mark.parametrize("n_pairs", [16, 3000])
def test_split_and_pair_values_on_the_gpu(ctx, oracle, torch_cuda, width, n_pairs):
    torch = torch_cuda
    rng = np.random.default_rng(n_pairs + width)
    lengths = LENGTHS if n_pairs == 16 else rng.integers(0, 130, n_pairs)
    a, b, off = ragged(lengths, seed=n_pairs)
    ga, gb, gctl, bo = ctx.pairs_to_beats(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), off, width)
    wa, wb, wctl = reference_split(a, b, off, width)
    assert np.array_equal(ga.cpu().numpy().view(u32), wa.view(u32)) and np.array_equal(gb.cpu().numpy().view(u32), wb.view(u32))
    assert np.array_equal(gctl.cpu().numpy().view(u32), wctl)

    n = int(gctl.shape[0])
    batch = capi.Batch()
    if width == 16:
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        batch.set_euclid(ga, gb, gctl, out, n)
        ctx.eval(batch)
        got = ctx.beats_to_pairs(out, bo).cpu().numpy()
        want_beats = oracle.euclid_beats(wa, wb, wctl)[0]
    else:
        dot = torch.empty(n, dtype=torch.float32, device="cuda")
        norm = torch.empty(n, dtype=torch.float32, device="cuda")
        batch.set_angular(ga, gb, gctl, dot, norm, n)
        ctx.eval(batch)
        got = ctx.beats_to_pairs(dot, bo).cpu().numpy()
        want_beats = oracle.angular_beats(wa, wb, wctl)[0]
    host_bo = bo.cpu().numpy()
    want = np.array([want_beats[int(host_bo[p + 1]) - 1] if host_bo[p + 1] > host_bo[p] else 0.0 for p in range(len(lengths))], f32)
    assert np.array_equal(got.view(u32), want.view(u32))
