"""The tiled segment layout as the header defines it (RT_DP_TILE_OPS, RT_DP_TILED_WORDS, RT_DP_TILED_INDEX), compiled from plain C
and compared with the Python helpers the GPU tests use (capi.tiled_words, the numpy index model)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = capi.TILE_OPS

PROBE = r"""
#include <stdio.h>
#include "rt_dp.h"
int main(void) {
    const unsigned long long n[] = {0, 1, 8191, 8192, 8193, 100003, 1ull << 26};
    const unsigned words[] = {RT_DP_BOX_TILE_IN_WORDS, RT_DP_BOX_TILE_OUT_WORDS, RT_DP_TRI_TILE_IN_WORDS, RT_DP_TRI_TILE_OUT_WORDS};
    printf("%u %u %u %u %u\n", RT_DP_TILE_OPS, words[0], words[1], words[2], words[3]);
    for (int k = 0; k < 7; ++k)
        for (int w = 0; w < 4; ++w) printf("W %llu %u %llu\n", n[k], words[w], (unsigned long long)RT_DP_TILED_WORDS(n[k], words[w]));
    const unsigned long long i[] = {0, 1, 255, 8191, 8192, 8193, 24576 + 5, 67108863};
    for (int k = 0; k < 8; ++k)
        for (unsigned f = 0; f < 34; f += 11) printf("I %llu %u %llu\n", i[k], f, (unsigned long long)RT_DP_TILED_INDEX(i[k], f, 34u));
    printf("F %u %u %u\n", RT_DP_F_BOX_TILED, RT_DP_F_TRI_TILED, RT_DP_F_ALL_FLAGS);
    return 0;
}
"""


def tile_index(i, word, words):
    return (i // T) * T * words + word * T + i % T


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not present")
def test_header_macros_agree_with_the_python_helpers():
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "p.c"), os.path.join(tmp, "p")
        with open(src, "w") as f:
            f.write(PROBE)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True, capture_output=True)
        lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    head = [int(x) for x in lines[0].split()]
    assert head == [T, capi.BOX_TILE_IN_WORDS, capi.BOX_TILE_OUT_WORDS, capi.TRI_TILE_IN_WORDS, capi.TRI_TILE_OUT_WORDS]
    seen = {"W": 0, "I": 0}
    for line in lines[1:]:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "W":
            n, words, got = int(tok[1]), int(tok[2]), int(tok[3])
            assert got == capi.tiled_words(n, words), line
            assert got % T == 0 and got >= n * words and got < (n + T) * words
        elif tok[0] == "I":
            i, word, got = int(tok[1]), int(tok[2]), int(tok[3])
            assert got == tile_index(i, word, 34), line
        elif tok[0] == "F":
            assert [int(x) for x in tok[1:]] == [capi.F_BOX_TILED, capi.F_TRI_TILED, 0xF]
            continue
        seen[tok[0]] += 1
    assert seen == {"W": 28, "I": 32}


def test_the_layout_is_a_bijection_onto_whole_tiles():
    """Every (op, word) of a segment has its own place, all places lie inside the buffer, and inside a tile the ops of one word are contiguous."""
    for n, words in ((1, 5), (T, 3), (T + 7, 16), (3 * T - 1, 34)):
        i = np.repeat(np.arange(n), words)
        w = np.tile(np.arange(words), n)
        idx = tile_index(i, w, words)
        assert idx.max() < capi.tiled_words(n, words)
        assert np.unique(idx).size == n * words
        first = tile_index(np.arange(min(n, T)), 2 % words, words)
        assert (np.diff(first) == 1).all()


def test_tile_aligned_shards_are_contiguous_slices_of_the_buffer():
    """shard.tile_cuts cuts a tiled segment on tile boundaries, so each rank's ops are one slice of the buffer and the slices
    re-tile to exactly what packing the rank's own ops would give."""
    from raytracer_amd import shard
    words = 5
    for n in (0, 1, T - 1, T, 5 * T + 77, 16 * T):
        plain = np.arange(n * words, dtype=np.uint32).reshape(words, n)          # word k of op i
        buf = np.zeros(capi.tiled_words(n, words), dtype=np.uint32)
        i = np.arange(n)
        for k in range(words):
            buf[tile_index(i, k, words)] = plain[k]
        for world in (1, 2, 3, 8):
            cuts = shard.tile_cuts(n, world)
            assert cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts[:-1], cuts[1:]))
            assert all(c % T == 0 for c in cuts[:-1])
            sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
            assert max(sizes) - min(sizes) <= T
            for a, b in zip(cuts[:-1], cuts[1:]):
                part = shard.tiled_slice(buf, words, a, b)
                assert part.shape[0] == capi.tiled_words(b - a, words)
                j = np.arange(b - a)
                for k in range(words):
                    assert (part[tile_index(j, k, words)] == plain[k, a:b]).all()
    with pytest.raises(ValueError):
        shard.tiled_slice(np.zeros(4 * T * words, dtype=np.uint32), words, 5, T)
