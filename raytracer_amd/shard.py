"""Partitioning of an opcode-sorted batch over the GPUs of a node.

Box and triangle ops carry no state, so a segment is cut into contiguous,
near-equal ranges.  Distance beats carry the accumulator from beat to beat
inside one vector pair (UnifiedDatapath.scala:1133-1154, :1028-1064), so a cut
may only fall on a pair boundary: directly after a beat whose reset bit is set.
With cuts on pair boundaries every shard starts from a +0 accumulator, the
shards are independent and no collective is needed on the data path.
"""
import numpy as np

RESET_BIT = 1 << 16


def even_cuts(n, world):
    """world+1 cut points of [0, n) into contiguous near-equal ranges."""
    return [(n * r) // world for r in range(world + 1)]


def tile_cuts(n, world, tile_ops=8192):
    """world+1 cut points of [0, n) that fall on whole tiles of a tiled segment.

    A tiled segment stores tile t of every word contiguously, so a range that
    starts and stops on a tile boundary is one contiguous slice of the buffer:
    words [start * W, stop * W) for W words per op (the last range also takes
    the partial tile).  Ranges differ by at most one tile.
    """
    tiles = -(-int(n) // tile_ops)
    return [min(int(n), ((tiles * r) // world) * tile_ops) for r in range(world)] + [int(n)]


def tiled_slice(buf, words_per_op, start, stop, tile_ops=8192):
    """The words of ops [start, stop) of a tiled buffer; start must be tile-aligned."""
    if start % tile_ops:
        raise ValueError("a tiled segment can only be cut on a tile boundary")
    end_tiles = -(-int(stop) // tile_ops)
    return buf[start * words_per_op: end_tiles * tile_ops * words_per_op]


def snap_to_pair_boundary(ctl, pos):
    """Smallest boundary >= pos: an index i with i == 0, i == n or reset(ctl[i-1])."""
    n = int(ctl.shape[0])
    pos = int(pos)
    if pos <= 0:
        return 0
    if pos >= n:
        return n
    rest = np.asarray(ctl[pos - 1:]) & RESET_BIT
    hits = np.flatnonzero(rest)
    return n if hits.size == 0 else pos + int(hits[0])


def vec_cuts(ctl, world):
    """Cut points for a beat segment: even cuts moved forward to the next pair boundary."""
    n = int(ctl.shape[0])
    cuts = [snap_to_pair_boundary(ctl, c) for c in even_cuts(n, world)]
    cuts[0], cuts[-1] = 0, n
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def plan(n_tri, n_box, euc_ctl, ang_ctl, world):
    """Per-rank (start, stop) ranges for the four segments of a batch."""
    tri, box = even_cuts(n_tri, world), even_cuts(n_box, world)
    euc = vec_cuts(euc_ctl, world) if euc_ctl is not None else [0] * (world + 1)
    ang = vec_cuts(ang_ctl, world) if ang_ctl is not None else [0] * (world + 1)
    return [{"tri": (tri[r], tri[r + 1]), "box": (box[r], box[r + 1]),
             "euclid": (euc[r], euc[r + 1]), "angular": (ang[r], ang[r + 1])} for r in range(world)]
