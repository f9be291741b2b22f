"""Tree index for the exact k-NN search on the Euclidean distance op.

A balanced k-d partition of the points: level by level every segment is split at the median
of its widest dimension, until 2^D leaves of `leaf_size` slots remain (the point count is
padded with +Inf slots, which collect in the last leaves).  The HIP kernel behind
rt_dp_knn_tree_euclid16 walks the leaves of this index per wave of 64 queries and skips a
leaf when a lower bound on the distance to its bounding box proves that it cannot
contribute; what it does scan goes through the same distance and list code as the
exhaustive kernel, so the answers are bit-identical to rt_dp_knn_euclid16.

The build runs on the GPU with torch (sort / gather per level); it is index construction,
outside the timed query path.
"""
import math

import torch

from . import knn

PAD_INDEX = 0xFFFFFFFF


class KnnTree:
    def __init__(self, points, leaf_size=128, index_base=0):
        assert points.dim() == 2 and points.shape[1] == 16 and points.dtype == torch.float32
        assert leaf_size >= 4 and leaf_size % 4 == 0
        dev = points.device
        n = int(points.shape[0])
        leaves = max(1, -(-n // leaf_size))
        depth = max(0, math.ceil(math.log2(leaves)))
        leaves = 1 << depth
        total = leaves * leaf_size
        pts = torch.full((total, 16), float("inf"), dtype=torch.float32, device=dev)
        pts[:n] = points
        idx = torch.full((total,), PAD_INDEX, dtype=torch.int64, device=dev)
        idx[:n] = torch.arange(n, device=dev, dtype=torch.int64) + index_base
        self.split_dim, self.split_val = [], []
        for level in range(depth):
            segs, seglen = 1 << level, total >> level
            v = pts.view(segs, seglen, 16)
            finite = torch.isfinite(v)
            hi = torch.where(finite, v, torch.full_like(v, float("-inf"))).amax(dim=1)
            lo = torch.where(finite, v, torch.full_like(v, float("inf"))).amin(dim=1)
            dim = torch.nan_to_num(hi - lo, nan=-1.0, posinf=-1.0, neginf=-1.0).argmax(dim=1)  # widest finite extent
            key = v.gather(2, dim.view(segs, 1, 1).expand(segs, seglen, 1)).squeeze(2)
            skey, order = key.sort(dim=1, stable=True)  # NaN and the +Inf pads sort last
            pts = v.gather(1, order.unsqueeze(2).expand(segs, seglen, 16)).reshape(total, 16)
            idx = idx.view(segs, seglen).gather(1, order).reshape(total)
            self.split_dim.append(dim)
            self.split_val.append(skey[:, seglen // 2].contiguous())  # first key of the right half
        v = pts.view(leaves, leaf_size, 16)
        pad = (idx == PAD_INDEX).view(leaves, leaf_size, 1)
        lo = torch.where(pad, torch.full_like(v, float("inf")), v).amin(dim=1)
        hi = torch.where(pad, torch.full_like(v, float("-inf")), v).amax(dim=1)
        # a NaN coordinate in a leaf makes min / max NaN: the box is unbounded in that dimension
        self.leaf_lo = torch.where(torch.isnan(lo), torch.full_like(lo, float("-inf")), lo).contiguous()
        self.leaf_hi = torch.where(torch.isnan(hi), torch.full_like(hi, float("inf")), hi).contiguous()
        self.points = pts.contiguous()
        self.index = torch.where(idx >= 2 ** 31, idx - 2 ** 32, idx).to(torch.int32).contiguous()  # bit image of the uint32 index
        self.n_leaves, self.leaf_size, self.depth, self.n_points = leaves, leaf_size, depth, n

    @classmethod
    def build_device(cls, ctx, points, leaf_size=128, index_base=0):
        """The same index through rt_dp_knn_tree_build_device (HIP kernels + one radix sort per level)."""
        t = ctx.knn_tree_build_device(points, leaf_size, index_base)
        self = cls.__new__(cls)
        self.points, self.index, self.leaf_lo, self.leaf_hi = t["points"], t["index"], t["leaf_lo"], t["leaf_hi"]
        self.n_leaves, self.leaf_size, self.n_points = t["n_leaves"], leaf_size, int(points.shape[0])
        self.depth = max(0, self.n_leaves.bit_length() - 1)
        # heap order -> one tensor per level, as home_leaf reads them
        self.split_dim = [t["split_dim"][(1 << l) - 1:(2 << l) - 1].to(torch.int64) for l in range(self.depth)]
        self.split_val = [t["split_val"][(1 << l) - 1:(2 << l) - 1] for l in range(self.depth)]
        return self

    def home_leaf(self, queries):
        """The leaf each query falls into when it descends the split planes (int32 [m])."""
        seg = torch.zeros(queries.shape[0], dtype=torch.int64, device=queries.device)
        for level in range(self.depth):
            d = self.split_dim[level][seg]
            x = queries.gather(1, d.view(-1, 1)).squeeze(1)
            seg = seg * 2 + (x >= self.split_val[level][seg]).to(torch.int64)
        return seg.to(torch.int32)


def default_splits(tree, m):
    """Enough workgroups for 256 CUs x a few blocks: 256 queries per block."""
    blocks = (m + 255) // 256
    s = 1
    while s * 2 <= min(tree.n_leaves, 64) and blocks * s < 4096:
        s *= 2
    return s


def local_topk(ctx, tree, queries, splits=None, k=knn.TOPK, stream=None):
    """Top-k of every query over the tree's points, as int64 keys [m, k] (same format as knn.local_topk)."""
    m = int(queries.shape[0])
    home = tree.home_leaf(queries)
    order = home.argsort(stable=True)
    qs = queries[order].contiguous()
    hs = home[order].contiguous()
    splits = splits or default_splits(tree, m)
    d = torch.empty(splits, m, knn.TOPK, dtype=torch.float32, device=queries.device)
    i = torch.empty(splits, m, knn.TOPK, dtype=torch.int32, device=queries.device)
    bound = torch.empty(m, dtype=torch.int32, device=queries.device)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    if s == 0:  # the library reads a NULL stream as "context stream"; order after torch's work instead
        torch.cuda.current_stream().synchronize()
    ctx.knn_tree_euclid16(tree, qs, hs, bound, d, i, splits=splits, stream=s)
    if s == 0:
        ctx.synchronize()
    keys = knn.merge_keys(knn.pack_keys(d, i), k)
    out = torch.empty_like(keys)
    out[order] = keys
    return out


def local_topk_angular(ctx, tree, queries, splits=None, k=knn.TOPK, stream=None, stats=None):
    """Top-k of every query by the angular op's score over the tree's points, as signed keys [m, k] of (minus score, index):
    the same keys as knn.local_topk_angular on the same points (rt_dp_knn_tree_angular16 skips only leaves that provably
    cannot contribute).  stats (a dict, or None) receives the share of (wave, leaf) pairs that were scanned."""
    m = int(queries.shape[0])
    home = tree.home_leaf(queries)
    order = home.argsort(stable=True)
    qs = queries[order].contiguous()
    hs = home[order].contiguous()
    splits = splits or default_splits(tree, m)
    v = torch.empty(splits, m, knn.TOPK, dtype=torch.float32, device=queries.device)
    i = torch.empty(splits, m, knn.TOPK, dtype=torch.int32, device=queries.device)
    bound = torch.empty(m, dtype=torch.int32, device=queries.device)
    scanned = torch.zeros(1, dtype=torch.int64, device=queries.device) if stats is not None else None
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    if s == 0:
        torch.cuda.current_stream().synchronize()
    ctx.knn_tree_angular16(tree, qs, hs, bound, v, i, splits=splits, leaves_scanned=scanned, stream=s)
    if s == 0:
        ctx.synchronize()
    if stats is not None:
        waves = ((m + 255) // 256) * 4
        stats["leaves_scanned_fraction"] = float(scanned.item()) / float(waves * tree.n_leaves)
    keys = knn.merge_keys(knn.pack_signed_keys(v, i), k)
    out = torch.empty_like(keys)
    out[order] = keys
    return out


def local_topk_grouped(ctx, tree, queries, k=knn.TOPK, walk_leaves=128, stats=None):
    """The same keys as local_topk through the leaf-major search (for data on which the wave walk cannot prune)."""
    m = int(queries.shape[0])
    home = tree.home_leaf(queries)
    order = home.argsort(stable=True)
    qs = queries[order].contiguous()
    hs = home[order].contiguous()
    d = torch.empty(m, knn.TOPK, dtype=torch.float32, device=queries.device)
    i = torch.empty(m, knn.TOPK, dtype=torch.int32, device=queries.device)
    torch.cuda.current_stream().synchronize()
    st = ctx.knn_tree_grouped_euclid16(tree, qs, hs, d, i, walk_leaves=walk_leaves)
    if stats is not None:
        stats.update(st)
    keys = knn.pack_keys(d, i)[:, :k].contiguous()
    out = torch.empty_like(keys)
    out[order] = keys
    return out


def search(ctx, tree, queries, k=knn.TOPK, group=None, splits=None, grouped=False, walk_leaves=128):
    """Sharded exact k-NN through per-GPU trees: every rank holds the tree of its shard of the points
    (built with that shard's index_base); one all-gather merges the partial lists."""
    keys = local_topk_grouped(ctx, tree, queries, k=k, walk_leaves=walk_leaves) if grouped else local_topk(ctx, tree, queries, splits=splits, k=k)
    return knn.unpack_keys(knn.all_gather_merge(keys, k, group))
