#!/usr/bin/env python3
"""Writes rtl_dump_sample.vec: the hand-derived known-answer cases as ONE record stream, in exactly the text form
that the print-only monitor of docs/datapath_test_dump.patch emits from the reference's testbench (BigInt.toString(16):
lower-case hex, no leading zeros, one "I" line per accepted record, one "O" line per delivered record).

The expected outputs below are typed in by hand, not computed: every intermediate of these cases is exact in binary32,
so they follow from the dataflow of src/main/scala/raytracer_datapath/UnifiedDatapath.scala alone
  K1  four default boxes (all +Inf, raytracer_gold/SW_Data.scala:106-113), ray (0,0,0) -> (1,1,1), extent 4000:
      no hit, tmin_out = +Inf x 4, boxIndex = 3,2,1,0 (no swap on equal keys, then reversed, :1120-1130)
  K2  all-zero triangle: t_num = t_denom = +0, triangle_hit = 0 (t_denom == 0, :1087-1116)
  K3  a = 0..15, b = 16..1 (Datapath_test.scala:132-133): sum (2i-16)^2 = 1376; first 8 elements only: 816
  K4  query 0..7, candidate 16..9: dot = 308, norm = 1292
  K6  the K3 beat twice without / with reset, then once more: 1376, 2752, 1376
"""
import os
import struct


def f(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


ONE, EXTENT, INF = f(1.0), f(4000.0), 0x7F800000
RAY = [0, 0, 0, ONE, ONE, ONE, ONE, ONE, ONE, EXTENT, 1, 2, 0, ONE, ONE, ONE]   # origin, dir, inv, extent, kx ky kz, shear
A16 = [f(float(i)) for i in range(16)]
B16 = [f(float(16 - i)) for i in range(16)]
Q8 = A16[:8] + [0] * 8
C8 = B16[:8] + [0] * 8


def rec(opcode, ray, aabb, tri, a, b, mask, reset):
    w = [opcode] + ray + aabb + tri + a + b + [mask, reset]
    assert len(w) == 84
    return w


def out(opcode, tmin=(0, 0, 0, 0), isect=(0, 0, 0, 0), index=(0, 0, 0, 0), t_num=0, t_denom=0, hit=0, acc=0, dot=0, norm=0, reset=0):
    # both *_reset_accum outputs echo the input bit for every opcode (UnifiedDatapath.scala:1254, :1267-1268)
    return [opcode] + list(tmin) + list(isect) + list(index) + [t_num, t_denom, hit, acc, reset, dot, norm, reset]


Z16, Z24, Z9, ZRAY = [0] * 16, [0] * 24, [0] * 9, [0] * 16
STREAM = [
    (rec(1, RAY, [INF] * 24, Z9, Z16, Z16, 0, 0), out(1, tmin=(INF,) * 4, index=(3, 2, 1, 0))),            # K1
    (rec(0, RAY, Z24, Z9, Z16, Z16, 0, 0), out(0)),                                                        # K2
    (rec(2, ZRAY, Z24, Z9, A16, B16, 0xFFFF, 1), out(2, acc=0x44AC0000, reset=1)),                         # K3
    (rec(2, ZRAY, Z24, Z9, A16, B16, 0x00FF, 1), out(2, acc=0x444C0000, reset=1)),                         # K3, masked
    (rec(3, ZRAY, Z24, Z9, Q8, C8, 0xFF, 1), out(3, dot=0x439A0000, norm=0x44A18000, reset=1)),            # K4
    (rec(2, ZRAY, Z24, Z9, A16, B16, 0xFFFF, 0), out(2, acc=0x44AC0000, reset=0)),                         # K6 beat 1
    (rec(2, ZRAY, Z24, Z9, A16, B16, 0xFFFF, 1), out(2, acc=0x452C0000, reset=1)),                         # K6 beat 2
    (rec(2, ZRAY, Z24, Z9, A16, B16, 0xFFFF, 1), out(2, acc=0x44AC0000, reset=1)),                         # K6 restart
]

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rtl_dump_sample.vec")
with open(path, "w") as fh:
    fh.write("# hand-derived known answers in the output format of docs/datapath_test_dump.patch; written by make_kat.py\n")
    fh.write("S hand_derived_kat extended=1\n")
    for i, o in STREAM:
        fh.write("I " + " ".join("%x" % x for x in i) + "\n")
        fh.write("O " + " ".join("%x" % x for x in o) + "\n")
print(path)
