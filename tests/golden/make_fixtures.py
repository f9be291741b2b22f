#!/usr/bin/env python3
"""Generates tests/golden/*.vec: the six Datapath_test suites with fixed seeds, plus
edge-case suites, as input records and expected output records.

    python tests/golden/make_fixtures.py

IMPORTANT: the expected outputs come from this repository's gate-level
restatement of the RTL (oracle A, oracle/hardfloat_rtl.c), NOT from a Verilator
run of the reference: its Chisel tool-chain (JVM, mill, Verilator) is not
available offline, and its own test draws unseeded random inputs and holds no
vectors.  The suites mirror src/test/scala/raytracer_datapath/Datapath_test.scala:
  baseline / extended box      :142-160, :191-202   (100 x genRandomRay(1e2) + 4 x genRandomBox(1e4))
  baseline / extended triangle :162-189, :204-217   (100 x genRandomTriangle + ray through the centroid)
  euclidean, angular           :219-241             (100 vector pairs, length U{1..128})
Extended box / triangle records carry vec_a = 0..15, vec_b = 16..1, reset = 1, mask = 0 (:393-410).

File format (text): '#' comment lines, one 'S <name> extended=<0|1>' line, then per record
  I <84 hex words of rt_dp_input_bundle>
  O <21 hex words of rt_dp_output_bundle>
A fresh datapath (accumulators +0) is assumed at the start of each file.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import dp_oracle as O  # noqa: E402
from oracle import gen  # noqa: E402

u32, f32 = np.uint32, np.float32
N = 100


def with_default_vectors(w):
    w = w.copy()
    w[:, 50:66] = np.arange(16, dtype=f32).view(u32)
    w[:, 66:82] = np.arange(16, 0, -1).astype(f32).view(u32)
    w[:, 82] = 0
    w[:, 83] = 1
    return w


def write(name, words, extended):
    out, _ = O.eval_bundles("A", words.view(O.IN_BUNDLE).reshape(-1), extended=bool(extended))
    chk, _ = O.eval_bundles("B", words.view(O.IN_BUNDLE).reshape(-1), extended=bool(extended))
    ow = out.view(u32).reshape(-1, 21)
    assert (ow == chk.view(u32).reshape(-1, 21)).all(), name
    path = os.path.join(HERE, name + ".vec")
    with open(path, "w") as f:
        f.write("# expected outputs: oracle A (this repository's restatement of the RTL), not a Verilator run\n")
        f.write("S %s extended=%d\n" % (name, extended))
        for i in range(words.shape[0]):
            f.write("I " + " ".join("%x" % x for x in words[i]) + "\n")
            f.write("O " + " ".join("%x" % x for x in ow[i]) + "\n")
    print(path, words.shape[0], "records")


def main():
    box = gen.aos_from_box(gen.box_ops(N, seed=0x5EED0001))
    tri = gen.aos_from_tri(*gen.tri_ops(N, seed=0x5EED0002))
    write("baseline_ray_box_random_100", box, 0)
    write("baseline_ray_triangle_random_100", tri, 0)
    write("extended_ray_box_random_100", with_default_vectors(box), 1)
    write("extended_ray_triangle_random_100", with_default_vectors(tri), 1)
    # the reference feeds the SAME vector pairs to the euclidean and to the angular test
    rng = np.random.default_rng(0x5EED0005)
    lengths = rng.integers(0, 128, N) + 1
    a16, b16, ctl16, _ = gen.vec_beats(N, 16, seed=0x5EED0005, lengths=lengths)
    write("extended_euclidean_100", gen.aos_from_vec(a16, b16, ctl16, 2), 1)
    a8, b8, ctl8, _ = gen.vec_beats(N, 8, seed=0x5EED0005, lengths=lengths)
    write("extended_angular_100", gen.aos_from_vec(a8, b8, ctl8, 3), 1)
    # edge-case suites (no counterpart in the reference's tests)
    write("edge_ray_box_200", with_default_vectors(gen.aos_from_box(gen.box_ops(200, seed=0x5EED0007, edge_frac=1.0))), 1)
    write("edge_ray_triangle_200", with_default_vectors(gen.aos_from_tri(*gen.tri_ops(200, seed=0x5EED0008, edge_frac=1.0))), 1)
    ea, eb, ectl, _ = gen.vec_beats(40, 16, seed=0x5EED0009, edge_frac=0.5)
    qa, qb, qctl, _ = gen.vec_beats(40, 8, seed=0x5EED000A, edge_frac=0.5)
    we, wa = gen.aos_from_vec(ea, eb, ectl, 2), gen.aos_from_vec(qa, qb, qctl, 3)
    wb = with_default_vectors(gen.aos_from_box(gen.box_ops(60, seed=0x5EED000B, edge_frac=0.3)))
    wt = with_default_vectors(gen.aos_from_tri(*gen.tri_ops(60, seed=0x5EED000C, edge_frac=0.3)))
    # interleave the four opcodes, each keeping its own order
    rng = np.random.default_rng(0x5EED000D)
    srcs = [wt, wb, we, wa]
    tags = np.concatenate([np.full(s.shape[0], i) for i, s in enumerate(srcs)])
    rng.shuffle(tags)
    pos = [0, 0, 0, 0]
    mixed = np.zeros((tags.size, 84), u32)
    for i, t in enumerate(tags):
        mixed[i] = srcs[t][pos[t]]
        pos[t] += 1
    write("edge_mixed_stream", mixed, 1)


if __name__ == "__main__":
    main()
