#!/usr/bin/env python3
"""Parity soak: many seeded random + edge-case streams of every opcode through the C ABI against the
scalar golden model (oracle B), bit for bit.  Not part of the test suite (minutes of run time):

    python tests/soak/soak.py [seeds] [ops per stream]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import dp_oracle as O  # noqa: E402  (the checker)
from raytracer_amd import capi  # noqa: E402
from oracle import gen  # noqa: E402
from helpers import hip_angular, hip_box, hip_euclid, hip_tiled, hip_tri  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 19
import torch  # noqa: E402

O.build()
ctx = capi.Context()
bad = 0
t0 = time.time()
for s in range(seeds):
    edge = [0.0, 0.01, 0.3, 0.9][s % 4]
    f = gen.box_ops(n + s, seed=1000 + s, edge_frac=edge)
    wt, wf = O.box_soa(f)
    gt, gf = hip_box(ctx, f, device=True, torch=torch)
    ok = all((a.view(np.uint32) == b.view(np.uint32)).all() for a, b in zip(gt, wt)) and (gf == wf).all()
    tf, tk = gen.tri_ops(n + s, seed=2000 + s, edge_frac=edge)
    wn, wd, wh = O.tri_soa(tf, tk)
    gn, gd, gh = hip_tri(ctx, tf, tk, device=True, torch=torch)
    ok &= (gn.view(np.uint32) == wn.view(np.uint32)).all() and (gd.view(np.uint32) == wd.view(np.uint32)).all() and (gh == wh).all()
    # the same ops through the tiled layout
    tb = hip_tiled(ctx, f, n + s, 5, False, torch)
    ok &= all((tb[k] == np.asarray(wt[k]).view(np.uint32)).all() for k in range(4)) and (tb[4] == np.asarray(wf).view(np.uint32)).all()
    tt = hip_tiled(ctx, list(tf) + [tk], n + s, 3, True, torch)
    ok &= (tt[0] == wn.view(np.uint32)).all() and (tt[1] == wd.view(np.uint32)).all() and (tt[2] == np.asarray(wh).view(np.uint32)).all()
    a, b, ctl, _ = gen.vec_beats(n // 40, 16, seed=3000 + s, edge_frac=edge * 0.3)
    we, _ = O.euclid_beats(a, b, ctl)
    ctx.reset()
    ge = hip_euclid(ctx, a, b, ctl, device=True, torch=torch)
    ok &= (np.asarray(ge).view(np.uint32) == we.view(np.uint32)).all()
    q, c, ctl8, _ = gen.vec_beats(n // 40, 8, seed=4000 + s, edge_frac=edge * 0.3)
    wdot, wnorm, _, _ = O.angular_beats(q, c, ctl8)
    ctx.reset()
    gdot, gnorm = hip_angular(ctx, q, c, ctl8, device=True, torch=torch)
    ok &= (np.asarray(gdot).view(np.uint32) == wdot.view(np.uint32)).all() and (np.asarray(gnorm).view(np.uint32) == wnorm.view(np.uint32)).all()
    bad += 0 if ok else 1
    print("seed %d edge %.2f: %s  (%.0f s)" % (s, edge, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
print("soak: %d streams of %d ops per opcode, %d mismatching" % (seeds, n, bad))
sys.exit(1 if bad else 0)
