"""Shared plumbing of the parity tests: run a segment through the C ABI, either from
host arrays (library-staged) or from torch CUDA tensors (in-place HBM)."""
import numpy as np

from raytracer_amd import capi

f32, u32 = np.float32, np.uint32


def bits(x):
    return np.ascontiguousarray(x).view(u32)


def assert_bits_equal(got, want, what):
    g, w = bits(got), bits(want)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, "%s: %d of %d differ, first at %d: got 0x%08x want 0x%08x" % (
        what, bad.size, g.size, bad[0], g.ravel()[bad[0]], w.ravel()[bad[0]])


def _to_dev(torch, arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


def hip_box(ctx, fields, device=False, torch=None):
    n = fields[0].shape[0]
    b = capi.Batch()
    if device:
        d = _to_dev(torch, fields)
        tmin = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]
        flags = torch.empty(n, dtype=torch.int32, device="cuda")
        b.set_box(d, tmin, flags, n)
        ctx.eval(b)
        return [t.cpu().numpy() for t in tmin], flags.cpu().numpy().view(u32)
    tmin = [np.full(n, 7.0, f32) for _ in range(4)]
    flags = np.full(n, 0xDEAD, u32)
    b.set_box([np.ascontiguousarray(f, f32) for f in fields], tmin, flags, n)
    ctx.eval(b)
    return tmin, flags


def hip_tri(ctx, fields, k, device=False, torch=None):
    n = k.shape[0]
    b = capi.Batch()
    if device:
        d = _to_dev(torch, fields)
        dk = torch.from_numpy(k.view(np.int32)).cuda()
        tn = torch.empty(n, dtype=torch.float32, device="cuda")
        td = torch.empty(n, dtype=torch.float32, device="cuda")
        hit = torch.empty(n, dtype=torch.int32, device="cuda")
        b.set_tri(d, dk, tn, td, hit, n)
        ctx.eval(b)
        return tn.cpu().numpy(), td.cpu().numpy(), hit.cpu().numpy().view(u32)
    tn, td, hit = np.full(n, 7.0, f32), np.full(n, 7.0, f32), np.full(n, 9, u32)
    b.set_tri([np.ascontiguousarray(f, f32) for f in fields], np.ascontiguousarray(k, u32), tn, td, hit, n)
    ctx.eval(b)
    return tn, td, hit


def hip_euclid(ctx, a, b_, ctl, device=False, torch=None):
    n = ctl.shape[0]
    bt = capi.Batch()
    if device:
        da, db = _to_dev(torch, [a, b_])
        dc = torch.from_numpy(ctl.view(np.int32)).cuda()
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        bt.set_euclid(da, db, dc, out, n)
        ctx.eval(bt)
        return out.cpu().numpy()
    out = np.full(n, 7.0, f32)
    bt.set_euclid(np.ascontiguousarray(a, f32), np.ascontiguousarray(b_, f32), np.ascontiguousarray(ctl, u32), out, n)
    ctx.eval(bt)
    return out


def hip_angular(ctx, q, c, ctl, device=False, torch=None):
    n = ctl.shape[0]
    bt = capi.Batch()
    if device:
        dq, dc_ = _to_dev(torch, [q, c])
        dctl = torch.from_numpy(ctl.view(np.int32)).cuda()
        dot = torch.empty(n, dtype=torch.float32, device="cuda")
        norm = torch.empty(n, dtype=torch.float32, device="cuda")
        bt.set_angular(dq, dc_, dctl, dot, norm, n)
        ctx.eval(bt)
        return dot.cpu().numpy(), norm.cpu().numpy()
    dot, norm = np.full(n, 7.0, f32), np.full(n, 7.0, f32)
    bt.set_angular(np.ascontiguousarray(q, f32), np.ascontiguousarray(c, f32), np.ascontiguousarray(ctl, u32), dot, norm, n)
    ctx.eval(bt)
    return dot, norm


def hip_tiled(ctx, arrays, n, words_out, tri, torch):
    """Box (34 arrays) or triangle (15 arrays + k) ops through the tiled layout: packed on the device, evaluated, unpacked.
    Returns `words_out` uint32 numpy arrays."""
    dev = _to_dev(torch, [np.ascontiguousarray(a).view(np.int32) for a in arrays])
    operands = torch.empty(capi.tiled_words(n, len(arrays)), dtype=torch.int32, device="cuda")
    results = torch.empty(capi.tiled_words(n, words_out), dtype=torch.int32, device="cuda")
    out = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(words_out)]
    b = capi.Batch()
    (b.set_tri_tiled if tri else b.set_box_tiled)(operands, results, n)
    torch.cuda.synchronize()
    ctx.tile_pack(dev, operands, n)
    ctx.launch(b)
    ctx.tile_unpack(results, out, n)
    ctx.synchronize()
    return [o.cpu().numpy().view(u32) for o in out]
