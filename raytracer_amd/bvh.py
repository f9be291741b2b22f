"""4-wide bounding volume hierarchy over triangles, in the layout rt_dp_trace_closest reads.

Host-side construction (numpy): every node quarters its triangle range by two rounds of
median splits along the widest axis of the centroids; ranges of at most `leaf_size`
triangles become leaves.  Children are always created after their parent, so an inner
child's index is larger than its parent's, which is what the traversal's validation pass
demands.  An absent child carries the all-+Inf box of the reference's default box
(src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:106-113), which no ray hits.
"""
import numpy as np

NONE = 0xFFFFFFFF
LEAF = 0x80000000
NODE_WORDS = 32
TRI_WORDS = 12
f32, u32 = np.float32, np.uint32


def leaf_ref(first, count):
    assert 1 <= count <= 8 and first < (1 << 28)
    return LEAF | (first << 3) | (count - 1)


def build_bvh4(tris, leaf_size=4):
    """tris [n, 9] float32 (A xyz, B xyz, C xyz).  Returns (nodes [m, 32] float32 with the child references
    as uint32 bit images in words 24..27, tris12 [n, 12] float32 in leaf order, order [n]: the caller's index
    of each stored triangle)."""
    tris = np.ascontiguousarray(tris, f32).reshape(-1, 9)
    n = tris.shape[0]
    assert 1 <= leaf_size <= 8
    v = tris.reshape(n, 3, 3)
    lo, hi = v.min(axis=1), v.max(axis=1)
    cen = (lo.astype(np.float64) + hi) * 0.5
    order = np.arange(n)

    def halve(a, b):  # sorts order[a:b] along the widest centroid axis; returns the midpoint
        if b - a > 1:
            c = cen[order[a:b]]
            axis = int(np.argmax(np.nan_to_num(c.max(axis=0) - c.min(axis=0), nan=0.0, posinf=0.0)))
            order[a:b] = order[a:b][np.argsort(c[:, axis], kind="stable")]
        return (a + b + 1) // 2

    boxes, refs = [], []

    def new_node():
        boxes.append(np.full(24, np.inf, f32))
        refs.append(np.full(4, NONE, np.uint32))
        return len(boxes) - 1

    stack = [(0, n, new_node())]
    while stack:
        a, b, nid = stack.pop()
        m = halve(a, b)
        cuts = [a, halve(a, m), m, halve(m, b), b]
        for c in range(4):
            s, e = cuts[c], cuts[c + 1]
            if e <= s:
                continue
            ids = order[s:e]
            boxes[nid][6 * c:6 * c + 3] = lo[ids].min(axis=0)
            boxes[nid][6 * c + 3:6 * c + 6] = hi[ids].max(axis=0)
            if e - s <= leaf_size:
                refs[nid][c] = leaf_ref(s, e - s)
            else:
                child = new_node()
                refs[nid][c] = child
                stack.append((s, e, child))
    nodes = np.zeros((len(boxes), NODE_WORDS), f32)
    nodes[:, :24] = np.stack(boxes)
    nodes[:, 24:28] = np.stack(refs).astype(u32).view(f32)
    tris12 = np.zeros((n, TRI_WORDS), f32)
    tris12[:, :9] = tris[order]
    return nodes, tris12, order


def random_scene(n_tris, seed=20241014, lo=-1000.0, hi=1000.0, size=60.0):
    """Small triangles scattered in a cube: vertex A uniform in [lo, hi]^3, B and C within `size` of A."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(lo, hi, (n_tris, 1, 3))
    v = a + np.concatenate([np.zeros((n_tris, 1, 3)), rng.uniform(-size, size, (n_tris, 2, 3))], axis=1)
    return v.reshape(n_tris, 9).astype(f32)


def rays_towards(tris, n_rays, seed=20241015, lo=-1000.0, hi=1000.0, jitter=0.0):
    """Origins uniform in the cube, directions towards the centroid of a random triangle (the reference's
    genRandomRayGivenPoint pattern, SW_Data.scala:375-388), optionally jittered."""
    rng = np.random.default_rng(seed)
    org = rng.uniform(lo, hi, (n_rays, 3)).astype(f32)
    pick = rng.integers(0, tris.shape[0], n_rays)
    cen = tris.reshape(-1, 3, 3).mean(axis=1, dtype=np.float64)[pick]
    d = (cen + rng.normal(0, jitter, (n_rays, 3)) if jitter else cen) - org
    return org, d.astype(f32)
