"""RT_DP_F_DERIVE_RAY: box and triangle ops whose inv / shear / k are computed inside the kernels from dir give the
bits of (a) the two-step GPU path, rt_dp_ray_prepare followed by a plain batch, on every row, and (b) the oracle
fed with the float32 restatement of SW_Ray (gen.ray_derived, SW_Data.scala:36-82) on rows with a finite direction
(the reference never exercises SW_Ray's comparisons on a NaN direction)."""
import numpy as np
import pytest

from raytracer_amd import capi
from oracle import gen

from helpers import assert_bits_equal, hip_box, hip_tri

pytestmark = pytest.mark.gpu
f32, u32 = np.float32, np.uint32
SPECIAL = np.array([0.0, -0.0, 1e-42, -1e-42, 3e38, -3e38, np.inf, -np.inf, np.nan, 1.0, -1.0, 2.0 ** -126, 1e-20], f32)


def directions(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.uniform(-100, 100, (n, 3)).astype(f32)
    rows = rng.choice(n, max(1, n // 8), replace=False)
    d[rows, rng.integers(0, 3, rows.size)] = SPECIAL[rng.integers(0, SPECIAL.size, rows.size)]
    ties = rng.choice(n, max(1, n // 50), replace=False)
    d[ties, 1] = d[ties, 0] * rng.choice(np.array([1, -1], f32), ties.size)
    d[ties[::4], 2] = d[ties[::4], 0]
    return d


def prepared(ctx, torch, d):
    n = d.shape[0]
    D = [torch.from_numpy(np.ascontiguousarray(d[:, c])).cuda() for c in range(3)]
    inv = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    shear = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    k = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.ray_prepare(D, inv=inv, k=k, shear=shear)
    ctx.synchronize()
    return [t.cpu().numpy() for t in inv], [t.cpu().numpy() for t in shear], k.cpu().numpy().view(u32)


@pytest.mark.parametrize("n", [1, 2, 63, 257, 4099, 300001])
@pytest.mark.parametrize("device", [False, True])
def test_box_with_derived_reciprocals(ctx, oracle, torch_cuda, n, device):
    torch = torch_cuda
    fields = [np.ascontiguousarray(f) for f in gen.box_ops(n, seed=300 + n, edge_frac=0.3)]
    d = directions(n, 301 + n)
    for c in range(3):
        fields[3 + c] = np.ascontiguousarray(d[:, c])
    inv_gpu, _, _ = prepared(ctx, torch, d)
    two_step = list(fields)
    for c in range(3):
        two_step[6 + c] = inv_gpu[c]
    want_t, want_f = hip_box(ctx, two_step, True, torch)

    fused = list(fields)
    for c in range(3):
        fused[6 + c] = None
    b = capi.Batch().derive_ray()
    if device:
        dev = [None if f is None else torch.from_numpy(f).cuda() for f in fused]
        tmin = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]
        flags = torch.empty(n, dtype=torch.int32, device="cuda")
        b.set_box(dev, tmin, flags, n)
        torch.cuda.synchronize()
        ctx.eval(b)
        got_t, got_f = [t.cpu().numpy() for t in tmin], flags.cpu().numpy().view(u32)
    else:
        got_t, got_f = [np.full(n, 7.0, f32) for _ in range(4)], np.full(n, 0xDEAD, u32)
        b.set_box(fused, got_t, got_f, n)
        ctx.eval(b)
    for k in range(4):
        assert_bits_equal(got_t[k], want_t[k], "tmin_out[%d] against the two-step path" % k)
    assert_bits_equal(got_f, want_f, "flags against the two-step path")

    ref = list(fields)
    inv = gen.ray_derived(d)[0]
    for c in range(3):
        ref[6 + c] = np.ascontiguousarray(inv[:, c])
    ot, of = oracle.box_soa(ref)
    for k in range(4):
        assert_bits_equal(got_t[k], ot[k], "tmin_out[%d] against the oracle" % k)
    assert_bits_equal(got_f, of, "flags against the oracle")


@pytest.mark.parametrize("n", [1, 2, 63, 257, 4099, 300001])
@pytest.mark.parametrize("device", [False, True])
def test_triangle_with_derived_shear(ctx, oracle, torch_cuda, n, device):
    torch = torch_cuda
    fields, _ = gen.tri_ops(n, seed=400 + n, edge_frac=0.3)
    fields = [np.ascontiguousarray(f) for f in fields]
    d = directions(n, 401 + n)
    # half of the rays aim at their triangle, like the reference's generator (SW_Data.scala:350-388)
    with np.errstate(all="ignore"):
        cen = np.stack([(fields[6 + c] + fields[9 + c] + fields[12 + c]) / f32(3) for c in range(3)], 1)
        org = np.stack(fields[0:3], 1)
        aim = np.arange(n) % 2 == 0
        d[aim] = (cen - org)[aim]
    _, shear_gpu, k_gpu = prepared(ctx, torch, d)
    two_step = list(fields)
    for c in range(3):
        two_step[3 + c] = shear_gpu[c]
    wn, wd, wh = hip_tri(ctx, two_step, k_gpu, True, torch)

    fused = list(fields)
    for c in range(3):
        fused[3 + c] = np.ascontiguousarray(d[:, c])
    b = capi.Batch().derive_ray()
    if device:
        dev = [torch.from_numpy(f).cuda() for f in fused]
        tn = torch.empty(n, dtype=torch.float32, device="cuda")
        td = torch.empty(n, dtype=torch.float32, device="cuda")
        hit = torch.empty(n, dtype=torch.int32, device="cuda")
        b.set_tri(dev, None, tn, td, hit, n)
        torch.cuda.synchronize()
        ctx.eval(b)
        gn, gd, gh = tn.cpu().numpy(), td.cpu().numpy(), hit.cpu().numpy().view(u32)
    else:
        gn, gd, gh = np.full(n, 7.0, f32), np.full(n, 7.0, f32), np.full(n, 9, u32)
        b.set_tri(fused, None, gn, gd, gh, n)
        ctx.eval(b)
    assert_bits_equal(gn, wn, "t_num against the two-step path")
    assert_bits_equal(gd, wd, "t_denom against the two-step path")
    assert_bits_equal(gh, wh, "hit against the two-step path")

    finite = np.isfinite(d).all(1)
    _, kx, ky, kz, shear = gen.ray_derived(d)
    ref = list(fields)
    for c in range(3):
        ref[3 + c] = np.ascontiguousarray(shear[:, c])
    on, od, oh = oracle.tri_soa(ref, gen.pack_k(kx, ky, kz))
    assert_bits_equal(gn[finite], np.asarray(on)[finite], "t_num against the oracle")
    assert_bits_equal(gd[finite], np.asarray(od)[finite], "t_denom against the oracle")
    assert_bits_equal(gh[finite], np.asarray(oh)[finite], "hit against the oracle")


def test_derive_needs_directions(ctx, torch_cuda):
    n = 8
    z = [np.zeros(n, f32) for _ in range(15)]
    b = capi.Batch()                      # plain mode, then the flag without tri.dir
    b.set_tri(z, np.zeros(n, u32), np.zeros(n, f32), np.zeros(n, f32), np.zeros(n, u32), n)
    b.finalize()
    b.ob.flags |= capi.F_DERIVE_RAY
    L = capi.load()
    import ctypes as C
    assert L.rt_dp_eval_ctx(ctx._h, C.byref(b.ob), C.byref(b.rb)) == capi.ENULL
