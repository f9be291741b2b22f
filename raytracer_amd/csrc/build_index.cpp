// Host-side construction of the two search structures the traversal / search entry points read, for callers
// that have no other builder at hand (no counterpart in the reference, which stops at the datapath):
//   rt_dp_knn_tree_build_host   balanced k-d partition of 16-dimensional points into 2^D leaves
//   rt_dp_bvh4_build_host       4-wide bounding volume hierarchy over triangles
// Plain host arrays in, plain host arrays out; the caller uploads the results.  Both are index construction,
// outside the timed query paths, and deliberately simple: median splits along the widest dimension.
#include "../../include/rt_dp.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

namespace {

constexpr uint32_t kPad = 0xFFFFFFFFu;
constexpr float kInf = std::numeric_limits<float>::infinity();

uint32_t leaves_for(uint32_t n, uint32_t leaf_size) {
    uint64_t want = (n + (uint64_t)leaf_size - 1) / leaf_size;
    uint32_t leaves = 1;
    while (leaves < want) leaves <<= 1;
    return leaves;
}

// NaN sorts last, like the +Inf pads: a total order for the partial sorts
inline bool key_less(float a, float b) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return !an && bn;
    return a < b;
}

}  // namespace

extern "C" {

uint32_t rt_dp_knn_tree_leaves(uint32_t n_points, uint32_t leaf_size) {
    if (leaf_size == 0) return 0;
    return leaves_for(n_points, leaf_size);
}

int rt_dp_knn_tree_build_host(const float *points, uint32_t n_points, uint32_t leaf_size, uint32_t index_base, float *out_points,
                              uint32_t *out_index, float *out_lo, float *out_hi, uint32_t *out_split_dim, float *out_split_val) {
    if (!points && n_points) return RT_DP_ENULL;
    if (!out_points || !out_index || !out_lo || !out_hi) return RT_DP_ENULL;
    if (leaf_size == 0 || (leaf_size & 3u)) return RT_DP_EINVAL;
    if ((uint64_t)index_base + n_points > 0xFFFFFFFFull) return RT_DP_EINVAL;
    const uint32_t leaves = leaves_for(n_points, leaf_size);
    const uint64_t total = (uint64_t)leaves * leaf_size;
    if (total > 0xFFFFFFFFull) return RT_DP_EINVAL;
    // slot -> point id (kPad for a pad slot); pads sit behind the real points of every segment
    std::vector<uint32_t> slot(total, kPad);
    std::iota(slot.begin(), slot.begin() + n_points, 0u);
    auto coord = [&](uint32_t id, int d) { return id == kPad ? kInf : points[(size_t)id * 16 + d]; };
    uint32_t depth = 0;
    while ((1u << depth) < leaves) ++depth;
    for (uint32_t level = 0; level < depth; ++level) {
        const uint32_t segs = 1u << level;
        const uint64_t len = total >> level;
        for (uint32_t sgi = 0; sgi < segs; ++sgi) {
            uint32_t *s0 = slot.data() + (uint64_t)sgi * len, *s1 = s0 + len;
            // widest finite extent
            int dim = 0;
            float best = -1.0f;
            for (int d = 0; d < 16; ++d) {
                float lo = kInf, hi = -kInf;
                for (uint32_t *p = s0; p < s1; ++p) {
                    if (*p == kPad) continue;
                    const float v = points[(size_t)*p * 16 + d];
                    if (std::isfinite(v)) { lo = std::min(lo, v); hi = std::max(hi, v); }
                }
                const float ext = hi - lo;  // -Inf for an empty segment
                if (std::isfinite(ext) && ext > best) { best = ext; dim = d; }
            }
            uint32_t *mid = s0 + len / 2;
            std::nth_element(s0, mid, s1, [&](uint32_t a, uint32_t b) {
                const float fa = coord(a, dim), fb = coord(b, dim);
                if (key_less(fa, fb)) return true;
                if (key_less(fb, fa)) return false;
                return a < b;  // ties (and pads, whose id is the largest) by id: deterministic
            });
            if (out_split_dim && out_split_val) {
                // nodes in heap order: level l occupies [2^l - 1, 2^(l+1) - 1); the first key of the right half
                float first = kInf;
                bool any = false;
                for (uint32_t *p = mid; p < s1; ++p) {
                    const float v = coord(*p, dim);
                    if (!any || key_less(v, first)) { first = v; any = true; }
                }
                out_split_dim[(segs - 1) + sgi] = (uint32_t)dim;
                out_split_val[(segs - 1) + sgi] = first;
            }
        }
    }
    for (uint32_t leaf = 0; leaf < leaves; ++leaf) {
        float lo[16], hi[16];
        bool nan_dim[16];
        for (int d = 0; d < 16; ++d) { lo[d] = kInf; hi[d] = -kInf; nan_dim[d] = false; }
        for (uint32_t k = 0; k < leaf_size; ++k) {
            const uint64_t s = (uint64_t)leaf * leaf_size + k;
            const uint32_t id = slot[s];
            out_index[s] = id == kPad ? kPad : index_base + id;
            for (int d = 0; d < 16; ++d) {
                const float v = coord(id, d);
                out_points[s * 16 + d] = v;
                if (id == kPad) continue;
                if (v != v) nan_dim[d] = true;
                else { lo[d] = std::min(lo[d], v); hi[d] = std::max(hi[d], v); }
            }
        }
        for (int d = 0; d < 16; ++d) {  // a NaN coordinate opens the box in that dimension
            out_lo[(size_t)leaf * 16 + d] = nan_dim[d] ? -kInf : lo[d];
            out_hi[(size_t)leaf * 16 + d] = nan_dim[d] ? kInf : hi[d];
        }
    }
    return RT_DP_OK;
}

int rt_dp_knn_home_leaf_host(const float *queries, uint32_t n_queries, uint32_t n_leaves, const uint32_t *split_dim, const float *split_val,
                             uint32_t *out_home) {
    if ((!queries || !out_home) && n_queries) return RT_DP_ENULL;
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return RT_DP_EINVAL;
    if (n_leaves > 1 && (!split_dim || !split_val)) return RT_DP_ENULL;
    for (uint32_t q = 0; q < n_queries; ++q) {
        uint32_t seg = 0;
        for (uint32_t segs = 1; segs < n_leaves; segs <<= 1) {
            const uint32_t node = (segs - 1) + seg;
            const float x = queries[(size_t)q * 16 + (split_dim[node] & 15u)];
            seg = seg * 2 + ((x >= split_val[node]) ? 1u : 0u);  // NaN goes left
        }
        out_home[q] = seg;
    }
    return RT_DP_OK;
}

uint32_t rt_dp_bvh4_max_nodes(uint32_t n_tris) {
    // every node has at least two non-empty children unless it holds a single leaf: fewer than n_tris + 1 nodes
    return n_tris + 1u;
}

int rt_dp_bvh4_build_host(const float *tris, uint32_t n_tris, uint32_t leaf_size, float *out_nodes, uint32_t *out_n_nodes, float *out_tris,
                          uint32_t *out_order) {
    if ((!tris && n_tris) || !out_nodes || !out_n_nodes || (n_tris && (!out_tris || !out_order))) return RT_DP_ENULL;
    if (leaf_size == 0 || leaf_size > 8 || n_tris > (1u << 28)) return RT_DP_EINVAL;
    std::vector<uint32_t> order(n_tris);
    std::iota(order.begin(), order.end(), 0u);
    std::vector<float> lo((size_t)n_tris * 3), hi((size_t)n_tris * 3), cen((size_t)n_tris * 3);
    for (uint32_t t = 0; t < n_tris; ++t)
        for (int a = 0; a < 3; ++a) {
            const float x = tris[(size_t)t * 9 + a], y = tris[(size_t)t * 9 + 3 + a], z = tris[(size_t)t * 9 + 6 + a];
            lo[(size_t)t * 3 + a] = std::min(x, std::min(y, z));
            hi[(size_t)t * 3 + a] = std::max(x, std::max(y, z));
            cen[(size_t)t * 3 + a] = 0.5f * lo[(size_t)t * 3 + a] + 0.5f * hi[(size_t)t * 3 + a];
        }
    auto halve = [&](uint32_t a, uint32_t b) {  // sorts order[a, b) along the widest centroid axis; returns the midpoint
        if (b - a > 1) {
            int axis = 0;
            float best = -1.0f;
            for (int d = 0; d < 3; ++d) {
                float l = kInf, h = -kInf;
                for (uint32_t i = a; i < b; ++i) { const float v = cen[(size_t)order[i] * 3 + d]; if (std::isfinite(v)) { l = std::min(l, v); h = std::max(h, v); } }
                if (std::isfinite(h - l) && h - l > best) { best = h - l; axis = d; }
            }
            std::stable_sort(order.begin() + a, order.begin() + b, [&](uint32_t p, uint32_t q) {
                return key_less(cen[(size_t)p * 3 + axis], cen[(size_t)q * 3 + axis]);
            });
        }
        return (a + b + 1) / 2;
    };
    struct Work { uint32_t a, b, node; };
    std::vector<Work> stack;
    uint32_t n_nodes = 0;
    auto new_node = [&]() {
        float *n = out_nodes + (size_t)n_nodes * 32;
        for (int w = 0; w < 24; ++w) n[w] = kInf;  // absent children: the all-+Inf box
        const uint32_t none = 0xFFFFFFFFu;
        for (int c = 0; c < 4; ++c) memcpy(n + 24 + c, &none, 4);
        for (int w = 28; w < 32; ++w) n[w] = 0.0f;
        return n_nodes++;
    };
    stack.push_back({0, n_tris, new_node()});
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        const uint32_t m = halve(w.a, w.b);
        const uint32_t cuts[5] = {w.a, halve(w.a, m), m, halve(m, w.b), w.b};
        for (int c = 0; c < 4; ++c) {
            const uint32_t s = cuts[c], e = cuts[c + 1];
            if (e <= s) continue;
            float *n = out_nodes + (size_t)w.node * 32;
            for (int a = 0; a < 3; ++a) {
                float l = kInf, h = -kInf;
                for (uint32_t i = s; i < e; ++i) { l = std::min(l, lo[(size_t)order[i] * 3 + a]); h = std::max(h, hi[(size_t)order[i] * 3 + a]); }
                n[6 * c + a] = l;
                n[6 * c + 3 + a] = h;
            }
            uint32_t ref;
            if (e - s <= leaf_size) {
                ref = 0x80000000u | (s << 3) | (e - s - 1);
            } else {
                ref = new_node();
                stack.push_back({s, e, ref});
            }
            memcpy(out_nodes + (size_t)w.node * 32 + 24 + c, &ref, 4);
        }
    }
    for (uint32_t i = 0; i < n_tris; ++i) {
        memcpy(out_tris + (size_t)i * 12, tris + (size_t)order[i] * 9, 9 * sizeof(float));
        out_tris[(size_t)i * 12 + 9] = out_tris[(size_t)i * 12 + 10] = out_tris[(size_t)i * 12 + 11] = 0.0f;
        out_order[i] = order[i];
    }
    *out_n_nodes = n_nodes;
    return RT_DP_OK;
}

}  // extern "C"
