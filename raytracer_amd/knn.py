"""k-nearest-neighbour search on the datapath's Euclidean distance op, sharded over GPUs.

Points are sharded (contiguous ranges, global index = shard base + position), queries
are replicated.  Every GPU computes, per query, its local top-k with the HIP kernel
behind rt_dp_knn_euclid16; the partial lists are exchanged with ONE all-gather (RCCL
over xGMI when the backend is "nccl": k * 8 bytes per query and GPU, 8 MiB for 64 Ki
queries at k = 16) and merged locally.  Candidates are ranked by (distance, index),
distances are never recomputed after the kernel, so the result is bit-identical for any
number of GPUs or splits.

A list entry travels as one int64 key: distance bits << 32 | index.  Distances are sums
of squares, i.e. non-negative floats or +Inf (a NaN distance never enters a list), for
which the IEEE bit pattern is order-preserving, so sorting keys sorts (distance, index).
"""
import torch
import torch.distributed as dist

TOPK = 16
EMPTY_KEY = (0x7F800000 << 32) | 0xFFFFFFFF  # (+Inf, 0xFFFFFFFF): an unused slot


def pack_keys(d, i):
    """float32 distances + int32 index images -> int64 keys."""
    return (d.contiguous().view(torch.int32).to(torch.int64) << 32) | (i.to(torch.int64) & 0xFFFFFFFF)


def unpack_keys(keys):
    """int64 keys -> (float32 distances, int64 indices)."""
    d = (keys >> 32).to(torch.int32).view(torch.float32)
    return d, keys & 0xFFFFFFFF


def pack_signed_keys(v, i):
    """Like pack_keys for values of either sign (the angular search's negated scores): the float's bits are mapped to
    an integer that orders like the float (negative: all bits flipped; non-negative: sign bit set) and shifted into
    the signed int64 range, so that sorting keys still sorts (value, index).  -0 must not occur (the kernel writes +0)."""
    b = v.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    u = torch.where(b >= 0x80000000, 0xFFFFFFFF - b, b + 0x80000000)
    return ((u - 0x80000000) << 32) | (i.to(torch.int64) & 0xFFFFFFFF)


def unpack_signed_keys(keys):
    u = (keys >> 32) + 0x80000000
    b = torch.where(u >= 0x80000000, u - 0x80000000, 0xFFFFFFFF - u)
    b = torch.where(b >= 0x80000000, b - (1 << 32), b)
    return b.to(torch.int32).view(torch.float32), keys & 0xFFFFFFFF


def merge_keys(keys, k=TOPK):
    """keys [..., m, n] stacked partial lists -> [m, k] smallest keys per query, ascending."""
    if keys.dim() == 3:
        keys = keys.permute(1, 0, 2).reshape(keys.shape[1], -1)
    return keys.sort(dim=1).values[:, :k].contiguous()


def local_topk(ctx, points, queries, index_base=0, splits=None, k=TOPK, stream=None):
    """Local top-k of every query over this GPU's points, as int64 keys [m, k]."""
    assert points.shape[1] == 16 and queries.shape[1] == 16 and 1 <= k <= TOPK
    m = int(queries.shape[0])
    if splits is None:
        # enough workgroups for the chip: 256 CUs x a few blocks, 256 queries per block
        blocks = (m + 255) // 256
        splits = max(1, min(64, (2048 + blocks - 1) // blocks, max(1, int(points.shape[0]) // 1024)))
    d = torch.empty(splits, m, TOPK, dtype=torch.float32, device=queries.device)
    i = torch.empty(splits, m, TOPK, dtype=torch.int32, device=queries.device)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    if s == 0:  # the library reads a NULL stream as "context stream"; order after torch's work instead
        torch.cuda.current_stream().synchronize()
    ctx.knn_euclid16(points, queries, d, i, index_base=index_base, splits=splits, stream=s)
    if s == 0:
        ctx.synchronize()
    return merge_keys(pack_keys(d, i), k)


def local_topk_angular(ctx, points, queries, index_base=0, splits=None, k=TOPK, stream=None):
    """Local top-k of every query by the angular op's score dot / sqrt(norm), largest first, as signed keys [m, k]
    of (minus score, index); unpack_signed_keys gives the negated scores back."""
    assert points.shape[1] == 16 and queries.shape[1] == 16 and 1 <= k <= TOPK
    m = int(queries.shape[0])
    if splits is None:
        blocks = (m + 255) // 256
        splits = max(1, min(64, (2048 + blocks - 1) // blocks, max(1, int(points.shape[0]) // 1024)))
    v = torch.empty(splits, m, TOPK, dtype=torch.float32, device=queries.device)
    i = torch.empty(splits, m, TOPK, dtype=torch.int32, device=queries.device)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    if s == 0:
        torch.cuda.current_stream().synchronize()
    ctx.knn_angular16(points, queries, v, i, index_base=index_base, splits=splits, stream=s)
    if s == 0:
        ctx.synchronize()
    return merge_keys(pack_signed_keys(v, i), k)


def all_gather_merge(keys, k=TOPK, group=None):
    """The one exchange of the k-NN path: all-gather of the per-GPU partial lists, then a local merge."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return keys
    world = dist.get_world_size(group)
    m, kk = keys.shape
    if keys.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the exchange without RCCL (e.g. several ranks on one GPU): gloo moves host memory
        host = keys.contiguous().cpu()
        gathered = torch.empty((world * m, kk), dtype=keys.dtype)
        dist.all_gather_into_tensor(gathered, host, group=group)
        return merge_keys(gathered.to(keys.device).view(world, m, kk), k)
    gathered = torch.empty((world * m, kk), dtype=keys.dtype, device=keys.device)  # rank-major concatenation
    dist.all_gather_into_tensor(gathered, keys.contiguous(), group=group)
    return merge_keys(gathered.view(world, m, kk), k)


def search(ctx, points_shard, queries, index_base, k=TOPK, group=None, local_fn=None, splits=None):
    """Sharded exact k-NN.  Returns (distances [m, k] float32, indices [m, k] int64), identical on every rank.

    local_fn replaces the GPU kernel in CPU rehearsals of the exchange (tests only)."""
    fn = local_fn if local_fn is not None else (lambda p, q, base: local_topk(ctx, p, q, base, splits=splits, k=k))
    keys = fn(points_shard, queries, index_base)
    return unpack_keys(all_gather_merge(keys, k, group))
