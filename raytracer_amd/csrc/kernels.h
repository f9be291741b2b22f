// Device-side entry points of librt_dp: one launcher per datapath opcode.
// All pointers are device pointers of the current GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtdp {

constexpr int kBoxFields = 34;  // org xyz, dir xyz, inv xyz, extent, 4 x (x_min,y_min,z_min,x_max,y_max,z_max)
constexpr int kTriFields = 15;  // org xyz, shear xyz, A xyz, B xyz, C xyz

// Tiled layout of box / triangle batches: [tile][word][kTileOps], whole tiles (rt_dp.h: RT_DP_TILE_OPS and the
// RT_DP_*_TILE_*_WORDS counts).
constexpr uint32_t kTileOps = 8192;
constexpr int kBoxOutWords = 5;   // tmin_out[0..3], flags
constexpr int kTriInWords = 16;   // the 15 operands, then the packed axis selectors
constexpr int kTriOutWords = 3;   // t_num, t_denom, hit
constexpr int kTileMaxArrays = 40;
struct TileList { uint32_t *a[kTileMaxArrays]; };

struct BoxArgs {
    const float *in[kBoxFields];
    float *tmin[4];
    uint32_t *flags;
};

struct TriArgs {
    const float *in[kTriFields];
    const uint32_t *k;
    float *t_num;
    float *t_denom;
    uint32_t *hit;
};

struct VecArgs {
    const float *a;       // [n][16] (Euclidean) or [n][8] (angular query)
    const float *b;       // [n][16] (Euclidean) or [n][8] (angular candidate)
    const uint32_t *ctl;  // mask | reset << 16
    float *out0;          // euclidean_accumulator / angular_dot_product
    float *out1;          // unused / angular_norm
    const float *state_in;  // accumulator register(s) entering beat 0: [1] Euclidean, [2] angular
    float *state_out;       // accumulator register(s) after the last beat
    // hand-over of the running sums from tile to tile (the chained shape of the beat kernels): vec_chain_bytes(n) bytes,
    // zero when allocated and never cleared afterwards; a slot is valid for the launch whose `epoch` it carries.
    // NULL selects the shape that needs no workspace.
    unsigned long long *chain = nullptr;
    unsigned long long *chain_header = nullptr;  // word 0 counts hand-overs that were given up on (never, unless a workgroup is lost or starved)
    uint32_t epoch = 0;
    uint32_t diag = 0;  // diagnostic shapes of the chained kernel (A/B only, results meaningless): 1 no accumulation walk, 2 no LDS stage either
    uint32_t debug_skip = 0;  // test hook: workgroup debug_skip - 1 does not publish its running sums, and waits give up after 2^12 polls
};

// Hand-over words of the chained beat kernels for n beats (any PASSES): 16 bytes per 128-beat tile.
size_t vec_chain_bytes(uint64_t n);

struct RayPrepArgs {
    const float *dir[3];
    float *inv[3];     // all three NULL: not wanted
    uint32_t *k;       // NULL: not wanted
    float *shear[3];   // all three NULL: not wanted
    float *extent;     // NULL: not wanted
};

// Tunables, settable per launch (0 = library default).  Exposed so the bench
// harness can A/B kernel shapes in one process.
struct Tuning {
    int ops_per_lane;  // box / triangle: 1, 2 or 4 consecutive ops per lane
    int nontemporal;   // streaming loads/stores carry the nt cache hint: 1 on, 2 off, 0 default
    int vec_tile;      // Euclidean / angular: nominal beats per workgroup
    int variant;       // 0 normal; 10 diagnostic: the box kernel's memory pattern without its arithmetic
    int vec_passes;    // Euclidean / angular: 16-byte load pairs in flight per lane and sub-tile (2, 4 or 8)
    int wgs_per_cu;    // resident workgroups per CU of the op kernels: 0 per-kernel default, 1..16 that many, 99 whatever fits
};

hipError_t launch_box(const BoxArgs &args, uint64_t n, const Tuning &t, hipStream_t stream, bool derive = false);  // derive: in[6..8] unused
hipError_t launch_tri(const TriArgs &args, uint64_t n, const Tuning &t, hipStream_t stream, bool derive = false);  // derive: in[3..5] = dir, k unused
hipError_t launch_box_tiled(const float *in, uint32_t *out, uint64_t n, const Tuning &t, hipStream_t stream, bool derive);
hipError_t launch_tri_tiled(const float *in, uint32_t *out, uint64_t n, const Tuning &t, hipStream_t stream, bool derive);
hipError_t launch_tile_pack(const TileList &arrays, int count, uint64_t n, uint32_t *tiled, hipStream_t stream);    // arrays are only read
hipError_t launch_tile_unpack(const uint32_t *tiled, int count, uint64_t n, const TileList &arrays, hipStream_t stream);
hipError_t launch_euclid(const VecArgs &args, uint64_t n, const Tuning &t, hipStream_t stream);
int resolved_residency(const Tuning &t, bool box);  // workgroups per CU that large box / triangle batches run with (0: no cap), per the occupancy query
hipError_t launch_angular(const VecArgs &args, uint64_t n, const Tuning &t, hipStream_t stream);
// Names of the kernels a launch runs, as rocprofv3 prints them.  chain_ok: the launch is not being captured into a graph.
// ops_per_lane_possible: 1, 2 or 4 by the common alignment of the segment's arrays (4, 8, 16 bytes).
void kernel_name_vec(bool euclid, uint64_t n, const Tuning &t, bool chain_ok, char *buf, size_t len);
void kernel_name_ops(bool box, bool tiled, bool derive, int ops_per_lane_possible, const Tuning &t, char *buf, size_t len);

// Stream front-end (stream.hip): device-side stable partition of record streams by opcode.
hipError_t stream_count_scan(const uint32_t *in, uint64_t n, uint32_t *blockcnt, uint32_t *totals_and_flag, hipStream_t s);
hipError_t stream_scatter(const uint32_t *in, uint64_t n, const uint32_t *blockoff, uint32_t *slot, const BoxArgs &box, const TriArgs &tri,
                          const VecArgs &euc, const VecArgs &ang, hipStream_t s);
hipError_t stream_gather(const uint32_t *slot, uint64_t n, const BoxArgs &box, const TriArgs &tri, const VecArgs &euc, const VecArgs &ang,
                         int extended, uint32_t *out, hipStream_t s);

hipError_t launch_ray_prepare(const RayPrepArgs &args, uint64_t n, hipStream_t stream);

// Exact top-16 nearest points per query under the datapath's Euclidean beat (dim 16).
// out_d / out_i: [splits][n_qry][16], ascending by (distance, index); unused slots hold (+Inf, 0xFFFFFFFF).
hipError_t launch_knn_euclid16(const float *pts, uint32_t n_pts, const float *qry, uint32_t n_qry, uint32_t idx_base,
                               uint32_t splits, bool angular, float *out_d, uint32_t *out_i, hipStream_t stream);
// dynamic LDS (never touched) that keeps `wgs_per_cu` workgroups of `threads` threads of `kernel` resident per CU, verified with the occupancy query
unsigned residency_pad(const void *kernel, int threads, int wgs_per_cu, int *achieved = nullptr);  // *achieved: what the query reports for that size
// fused record-stream path (stream_fused.hip)
struct StreamSideView {
    const float *hs[3];
    const void *heads, *tails;              // 16 bytes per evaluation wave each
    const unsigned long long *occ[2], *rst[2];
    const uint32_t *flags;  // per launch: [1] opcode 2 present, [2] opcode 3 present, [4] / [5] 1 + the last wave holding one; [8] opcode > 3 seen (sticky, cleared by the caller)
};
size_t stream_fused_workspace(uint64_t capacity);
int stream_eval_waves_per_cu();  // resident waves per CU the evaluation kernel of the two-pass form is launched with (0 = no cap)
hipError_t stream_fused_eval(const uint32_t *in, uint64_t n, uint64_t capacity, int extended, uint32_t *out, char *workspace,
                             StreamSideView *view, hipStream_t s, bool first_piece = false);
hipError_t stream_fused_fold(const StreamSideView &v, uint64_t n, const float *state_in, float *state_out, uint32_t *out, hipStream_t s);
// the same in one pass (running sums handed from wave to wave); state_in / state_out: {euclid, dot, norm}
size_t stream_chain_workspace(uint64_t capacity);
hipError_t stream_chain_begin(char *workspace, hipStream_t s);
hipError_t stream_chain_eval(const uint32_t *in, uint64_t n, uint64_t capacity, int extended, uint32_t *out, char *workspace, const float *state_in,
                             float *state_out, const unsigned long long **sticky_out, hipStream_t s, int mute_wave = -1 /* test hook */);

// hit-mask compaction behind a Quadbox batch (compact.hip); blockcnt [ceil(n/256)] u32, offsets64 likewise u64, total one u64.
// tile_words 0: flags / tmin are plain arrays; kBoxOutWords: they point at their word's run in tile 0 of a tiled result buffer
hipError_t compact_hits(const uint32_t *flags, const float *const tmin[4], uint64_t n, const uint32_t *item, const uint32_t *children,
                        uint32_t *blockcnt, unsigned long long *offsets64, unsigned long long *total, uint64_t capacity, uint32_t *out_item,
                        uint32_t *out_child, float *out_tmin, hipStream_t s, uint32_t tile_words = 0);

// the same with the offsets in a file-scope device array (no allocation on the call path): compact_static_init once per
// device with that device current, then compact_hits_static; *d_count (device memory) receives the number of entries
uint64_t compact_static_max_ops();
hipError_t compact_static_init(int device);
hipError_t compact_hits_static(int device, const uint32_t *flags, const float *const tmin[4], uint64_t n, const uint32_t *item,
                               const uint32_t *children, uint64_t capacity, uint32_t *out_item, uint32_t *out_child, float *out_tmin,
                               unsigned long long *d_count, hipStream_t s);

// vector pairs <-> beats (pairs.hip)
hipError_t pairs_to_beats(const float *a, const float *b, const uint64_t *pair_offsets, const uint64_t *beat_offsets, uint64_t n_pairs,
                          uint64_t n_beats, uint32_t width, float *out_a, float *out_b, uint32_t *out_ctl, hipStream_t s);
hipError_t beats_to_pairs(const float *values, const uint64_t *beat_offsets, uint64_t n_pairs, float *out, hipStream_t s);

// hierarchy construction on the device (build_device.hip)
size_t bvh_device_workspace(uint32_t n_tris);
hipError_t bvh_build_device(const float *tris, uint32_t n, uint32_t leaf_size, char *workspace, size_t workspace_bytes, float *out_nodes,
                            uint32_t node_capacity, uint32_t *n_nodes, float *out_tris, uint32_t *out_order, hipStream_t s);

// k-d index construction on the device (build_kd.hip)
size_t kd_device_workspace(uint32_t total_slots, uint32_t n_leaves);
hipError_t kd_build_device(const float *pts, uint32_t n, uint32_t n_leaves, uint32_t leaf_size, uint32_t index_base, char *workspace,
                           size_t workspace_bytes, float *out_points, uint32_t *out_index, float *out_lo, float *out_hi, uint32_t *split_dim,
                           float *split_val, hipStream_t s);
hipError_t kd_home_leaf(const float *qry, uint32_t n_qry, uint32_t n_leaves, const uint32_t *split_dim, const float *split_val, uint32_t *home,
                        hipStream_t s);

// traversal driver (trace.hip); queues hold (ray, node or leaf reference) pairs of 8 bytes
hipError_t trace_validate(const float *nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t *bad, hipStream_t s);
hipError_t trace_init(const float *const origin[3], const float *const dir[3], uint32_t n, float extent, float *rays,
                      unsigned long long *best, void *queue, hipStream_t s);
hipError_t trace_box(const void *cur, uint32_t count, const float *rays, const float *nodes, const unsigned long long *best, void *next,
                     uint32_t next_cap, void *leaf, uint32_t leaf_cap, uint32_t *counters, hipStream_t s);
hipError_t trace_tri(const void *leaf, uint32_t count, const float *rays, const float *tris, unsigned long long *best, uint32_t *counters,
                     hipStream_t s);
hipError_t trace_dfs(const float *const origin[3], const float *const dir[3], uint32_t n, float extent, const float *nodes, const float *tris,
                     float *out_t, uint32_t *out_tri, unsigned long long *totals, uint32_t *overflow, unsigned long long *next /* zeroed */,
                     hipStream_t s);
hipError_t trace_finish(const unsigned long long *best, uint32_t n, float *out_t, uint32_t *out_tri, hipStream_t s);
hipError_t launch_knn_tree16(const float *pts, const uint32_t *index, const float *leaf_lo, const float *leaf_hi, uint32_t n_leaves,
                             uint32_t leaf_size, const float *qry, const uint32_t *home, uint32_t n_qry, uint32_t log2_splits, uint32_t walk,
                             uint32_t *bound, float *out_d, uint32_t *out_i, hipStream_t stream);
// the angular search through the same index; *scanned (or NULL) accumulates the leaves scanned, summed over waves of 64 queries
hipError_t launch_knn_tree_ang16(const float *pts, const uint32_t *index, const float *leaf_lo, const float *leaf_hi, uint32_t n_leaves,
                                 uint32_t leaf_size, const float *qry, const uint32_t *home, uint32_t n_qry, uint32_t log2_splits,
                                 uint32_t *bound, float *out_d, uint32_t *out_i, unsigned long long *scanned, hipStream_t stream);
struct KnnTreeView {
    const float *pts;
    const uint32_t *index;
    const float *leaf_lo, *leaf_hi;
    uint32_t n_leaves, leaf_size;
};
hipError_t launch_knn_groups(const uint32_t *group_first, const uint32_t *count, uint32_t n_leaves, uint32_t *group_leaf, hipStream_t stream);
hipError_t launch_knn_pairs(const KnnTreeView &t, const float *qry, const uint32_t *subset, uint32_t n_qry, const uint32_t *tau,
                            uint32_t *count, const uint32_t *offset, uint32_t *list, unsigned long long *masks, hipStream_t stream);
hipError_t launch_knn_leaf_scan(const KnnTreeView &t, const float *qry, const uint32_t *tau, uint32_t n_groups, const uint32_t *group_leaf,
                                const uint32_t *group_first, const uint32_t *count, const uint32_t *offset, const uint32_t *list,
                                uint32_t cap, uint32_t *n_cand, unsigned long long *cand, hipStream_t stream);
hipError_t launch_knn_select(const uint32_t *tau, const uint32_t *n_cand, const unsigned long long *cand, uint32_t cap, uint32_t n_qry,
                             float *out_d, uint32_t *out_i, uint32_t *overflow, hipStream_t stream);
hipError_t launch_knn_tau_merge(const float *lists, uint32_t splits, uint32_t n_qry, uint32_t *tau, hipStream_t stream);
hipError_t launch_knn_retau(uint32_t *tau, const uint32_t *overflow, const float *out_d, uint32_t n_qry, uint32_t *status, hipStream_t stream);

}  // namespace rtdp
