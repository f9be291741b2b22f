/*
 * rt_dp -- batch evaluator for the raytracer_datapath `UnifiedDatapath` module
 * on AMD Instinct MI355X (gfx950).  C ABI of librt_dp.so.
 *
 * The RTL consumes one EnhancedInputBundle per clock on a Decoupled port and
 * delivers one EnhancedOutputBundle per record, in order, 12 registers later
 * (reference: src/main/scala/raytracer_datapath/UnifiedDatapath.scala:50-59,
 * :1168-1284).  This library evaluates the same function over whole batches:
 * results are bit-identical to the RTL's HardFloat arithmetic (round to
 * nearest even, subnormals kept, canonical NaN 0x7FC00000, the RTL's
 * compare-select and sort-network ordering, no fused multiply-add).
 *
 * All citations are file:line under the reference's
 * src/main/scala/raytracer_datapath/ unless a longer path is given.
 *
 * Conventions
 *   - Plain pointers and sizes only.  Every array is owned by the caller and
 *     is neither retained nor freed by the library.
 *   - A batch is opcode-sorted: one segment per opcode, laid out as a
 *     structure of arrays.  Result i of a segment belongs to op i of that
 *     segment (the RTL's in-order pipe).  Within the Euclidean and within the
 *     angular segment, beats are in stream order: the accumulator registers
 *     run through them exactly as through the RTL's stage 10 / stage 9.
 *   - Functions return RT_DP_OK (0) or a negative RT_DP_E* code and never
 *     throw or abort; rt_dp_last_error() describes the last failure of the
 *     calling thread.
 */
#ifndef RT_DP_H
#define RT_DP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_DP_VERSION_MAJOR 0
#define RT_DP_VERSION_MINOR 1

/* UnifiedDatapathOpCode, IOSpec.scala:10-24 */
enum rt_dp_opcode {
    RT_DP_OP_TRIANGLE  = 0, /* ray-triangle test                                  */
    RT_DP_OP_QUADBOX   = 1, /* ray vs four AABBs, sorted                          */
    RT_DP_OP_EUCLIDEAN = 2, /* one 16-element beat of sum((a-b)^2), accumulated   */
    RT_DP_OP_ANGULAR   = 3  /* one 8-element beat of dot(q,c) and dot(c,c), acc.  */
};

enum rt_dp_status {
    RT_DP_OK       = 0,
    RT_DP_EINVAL   = -1, /* malformed argument (bad struct_size, unknown flag, count too large) */
    RT_DP_ENULL    = -2, /* a required pointer is NULL (e.g. an operand array of a non-empty segment) */
    RT_DP_EOPCODE  = -3, /* opcode > 3 in a bundle stream (cannot occur in the 2-bit RTL field) */
    RT_DP_EHIP     = -4, /* HIP runtime error; see rt_dp_last_error() */
    RT_DP_ENODEV   = -5, /* no usable gfx950 device */
    RT_DP_EALIGN   = -6, /* an array lacks the alignment its entry point asks for (4, 8 or 16 bytes) */
    RT_DP_ENOMEM   = -7  /* host or device allocation failed */
};

/* rt_dp_op_batch.flags */
#define RT_DP_F_DEVICE_PTRS 0x1u /* every array pointer in the batch and in the result is a device pointer
                                    of the context's GPU; without it they are host pointers and the library
                                    stages them through its own device buffers */
#define RT_DP_F_DERIVE_RAY  0x2u /* the ray's derived fields are computed inside the box and triangle kernels
                                    from dir, exactly as rt_dp_ray_prepare would write them (the reference
                                    derives them in its software model before a record is poked:
                                    SW_Data.scala:36-82): box.inv[] and tri.shear[], tri.k are not read and
                                    may be NULL, tri.dir[] is read instead.  12 B (box) and 4 B (triangle)
                                    less traffic per op; results equal those of the two-step path bit for
                                    bit */
#define RT_DP_F_BOX_TILED   0x4u /* the Quadbox segment is in tiled form (below) */
#define RT_DP_F_TRI_TILED   0x8u /* the triangle segment is in tiled form */
#define RT_DP_F_ALL_FLAGS   0xFu

/* ---- tiled segments ---------------------------------------------------------
 * The layout this library runs fastest on (MI355X, any batch size: Quadbox 0.84 of the HBM peak against 0.68 to
 * 0.82 for separate arrays, whose rate depends on where the caller's allocator put them; triangle 0.87 against
 * 0.75 to 0.82).  A tiled segment is ONE buffer of 32-bit words, [tile][word][RT_DP_TILE_OPS]: op i lives in tile
 * i / RT_DP_TILE_OPS at position i % RT_DP_TILE_OPS, and within a tile the values of one operand for all its ops
 * are contiguous.  Buffers span whole tiles (RT_DP_TILED_WORDS); what lies behind op n - 1 in the last tile is
 * neither read nor written.  The words of an op, in order:
 *   Quadbox operands  (34): rt_dp_box_in in declaration order: org xyz, dir xyz, inv xyz, extent, aabb[0][0..5] ..
 *                           aabb[3][0..5]; the buffer is passed in box.org[0], every other pointer is ignored
 *   Quadbox results    (5): tmin_out[0..3], flags; passed in box.tmin_out[0]
 *   triangle operands (16): org xyz, shear xyz, A xyz, B xyz, C xyz, k; passed in tri.org[0]
 *   triangle results   (3): t_num, t_denom, hit; passed in tri.t_num
 * Under RT_DP_F_DERIVE_RAY the inv words (box) and the k word (triangle) are not read, and the shear words hold
 * the ray direction.  Host buffers work as well (without RT_DP_F_DEVICE_PTRS): a tiled segment then crosses the
 * link in one copy each way instead of one per operand array.  Results are the same bits as those of the plain form.  No counterpart in the reference: its
 * testbench pokes one record per cycle (Datapath_test.scala:275-302).                                          */
#define RT_DP_TILE_OPS            8192u
#define RT_DP_BOX_TILE_IN_WORDS   34u
#define RT_DP_BOX_TILE_OUT_WORDS  5u
#define RT_DP_TRI_TILE_IN_WORDS   16u
#define RT_DP_TRI_TILE_OUT_WORDS  3u
#define RT_DP_TILED_WORDS(n, words) \
    ((((uint64_t)(n) + RT_DP_TILE_OPS - 1u) / RT_DP_TILE_OPS) * RT_DP_TILE_OPS * (uint64_t)(words))
#define RT_DP_TILED_INDEX(i, word, words) \
    (((uint64_t)(i) / RT_DP_TILE_OPS) * RT_DP_TILE_OPS * (uint64_t)(words) + (uint64_t)(word) * RT_DP_TILE_OPS + (uint64_t)(i) % RT_DP_TILE_OPS)

/* ---- opcode 1: Quadbox --------------------------------------------------
 * Live operands of EnhancedInputBundle for this opcode (IOSpec.scala:54-91,
 * :106-133; dataflow UnifiedDatapath.scala:167-220, :346-398, :534-631,
 * :1117-1132): 34 f32 per op.  aabb[b] lists x_min,y_min,z_min,x_max,y_max,z_max. */
typedef struct rt_dp_box_in {
    const float *org[3];     /* ray.origin.{x,y,z} */
    const float *dir[3];     /* ray.dir.{x,y,z}   (only its sign / zero test is used) */
    const float *inv[3];     /* ray.inv.{x,y,z} */
    const float *extent;     /* ray.extent */
    const float *aabb[4][6]; /* aabb(b).{x_min,y_min,z_min,x_max,y_max,z_max} */
} rt_dp_box_in;

/* tmin_out[k], k = 0..3: hits first, nearest first; misses carry +Inf
 * (IOSpec.scala:142-149).  flags packs the remaining output fields:
 *   bits 3:0   isIntersect[0..3]   (bit k <-> output slot k)
 *   bits 11:4  boxIndex[0..3], two bits each, slot k at bits 5+2k:4+2k        */
typedef struct rt_dp_box_out {
    float    *tmin_out[4];
    uint32_t *flags;
} rt_dp_box_out;

#define RT_DP_BOX_ISECT(flags, k) (((flags) >> (k)) & 1u)
#define RT_DP_BOX_INDEX(flags, k) (((flags) >> (4 + 2 * (k))) & 3u)

/* ---- opcode 0: Triangle -------------------------------------------------
 * Live operands (IOSpec.scala:54-71, :93-99; dataflow UnifiedDatapath.scala:
 * :118-166, :293-345, :486-533, :688-726, :753-781, :844-871, :898-913,
 * :990-1005, :1087-1116): 15 f32 + the three 2-bit axis selectors.           */
typedef struct rt_dp_tri_in {
    const float    *org[3];    /* ray.origin */
    const float    *shear[3];  /* ray.shear */
    const float    *vtx[3][3]; /* triangle.{A,B,C}.{x,y,z} */
    const uint32_t *k;         /* ray.kx | ray.ky << 2 | ray.kz << 4; a selector value of 3 reads z,
                                  like Float3.at (IOSpec.scala:41-51) */
    const float    *dir[3];    /* ray.dir: read only under RT_DP_F_DERIVE_RAY, in place of shear and k */
} rt_dp_tri_in;

typedef struct rt_dp_tri_out {
    float    *t_num;   /* always written, hit or not */
    float    *t_denom;
    uint32_t *hit;     /* triangle_hit, 0 or 1 */
} rt_dp_tri_out;

/* ---- opcodes 2 and 3: Euclidean / angular beats ---------------------------
 * a, b: the two vectors of the segment, beat after beat, row-major
 *   Euclidean: [n][16] = euclidean_a / euclidean_b            (IOSpec.scala:128-132)
 *   angular:   [n][8]  = query (euclidean_a[0..7]) / candidate (euclidean_b[0..7])
 *                                                              (IOSpec.scala:229-243)
 * ctl[i] = euclidean_mask | euclidean_reset_accum << 16 (mask bit j <-> element j;
 * the angular op looks at mask bits 7:0 only).  A set reset bit marks the LAST
 * beat of a vector pair: that beat's result still includes it, and the
 * accumulator restarts from +0 on the next beat (UnifiedDatapath.scala:1146-1152,
 * :1055-1062).                                                               */
typedef struct rt_dp_vec_in {
    const float    *a;
    const float    *b;
    const uint32_t *ctl;
} rt_dp_vec_in;

#define RT_DP_CTL(mask, reset) ((uint32_t)(mask) | ((uint32_t)((reset) != 0) << 16))

typedef struct rt_dp_euc_out {
    float *accumulator; /* euclidean_accumulator after each beat */
} rt_dp_euc_out;

typedef struct rt_dp_ang_out {
    float *dot_product; /* angular_dot_product after each beat */
    float *norm;        /* angular_norm after each beat */
} rt_dp_ang_out;

/* ---- the batch ------------------------------------------------------------ */
typedef struct rt_dp_op_batch {
    uint32_t     struct_size; /* sizeof(rt_dp_op_batch) */
    uint32_t     flags;       /* RT_DP_F_* */
    uint64_t     n[4];        /* ops (beats) per opcode segment, indexed by rt_dp_opcode */
    rt_dp_tri_in tri;
    rt_dp_box_in box;
    rt_dp_vec_in euc;
    rt_dp_vec_in ang;
} rt_dp_op_batch;

typedef struct rt_dp_result_batch {
    uint32_t      struct_size; /* sizeof(rt_dp_result_batch) */
    uint32_t      reserved;    /* 0 */
    rt_dp_tri_out tri;
    rt_dp_box_out box;
    rt_dp_euc_out euc;
    rt_dp_ang_out ang;
} rt_dp_result_batch;

/* ---- contexts ---------------------------------------------------------------
 * One context = one datapath instance: it owns the three accumulator registers
 * (UnifiedDatapath.scala:1031-1032, :1136), a HIP stream and grow-only staging
 * buffers on one GPU.  Distinct contexts may be used from distinct threads; a
 * single context is not re-entrant.                                            */
typedef struct rt_dp_ctx rt_dp_ctx;

/* RaytracerParams, UnifiedDatapath.scala:13-23.  Like the RTL's elaboration
 * asserts (:36-47) rt_dp_create rejects io_recorded_float != 0,
 * internal_recorded_float != 1 and support_euclidean not in {0, 16}.
 * support_euclidean == 0 is the baseline datapath: opcodes 2 and 3 flow through
 * untouched, so their result arrays are filled with zeros.                     */
typedef struct rt_dp_config {
    uint32_t struct_size;             /* sizeof(rt_dp_config) */
    int32_t  device;                  /* HIP device ordinal, -1 = current device */
    uint32_t io_recorded_float;       /* must be 0 */
    uint32_t internal_recorded_float; /* must be 1 */
    uint32_t support_euclidean;       /* 0 (None) or 16 (Some(16)) */
    uint32_t reserved;
} rt_dp_config;

int  rt_dp_create(const rt_dp_config *cfg /* NULL = current device, Some(16) */, rt_dp_ctx **out);
void rt_dp_destroy(rt_dp_ctx *ctx);

/* Replaces driving UnifiedDatapath.in / .out for one batch (the chiseltest
 * enqueueSeq / waitForValid+peek loop of src/test/scala/raytracer_datapath/
 * Datapath_test.scala:275-302, :341-375, :476-496).  Synchronous: with device
 * pointers the call first waits for all work already queued on the device (the
 * operands may still be in production on a caller stream), and results are
 * visible when it returns.  rt_dp_eval uses a per-thread default context on
 * the current device.                                                          */
int rt_dp_eval(const rt_dp_op_batch *in, rt_dp_result_batch *out);
int rt_dp_eval_ctx(rt_dp_ctx *ctx, const rt_dp_op_batch *in, rt_dp_result_batch *out);

/* Asynchronous form for callers that keep their streams in HBM: requires
 * RT_DP_F_DEVICE_PTRS, enqueues the kernels on `hip_stream` (a hipStream_t;
 * NULL = the context's own non-blocking stream) and returns without
 * synchronising.  Ordering against the producers of the operands is the
 * caller's: launch on the stream that produced them, or synchronise first.
 * Batches of one context execute in the order they were given (they share its
 * accumulator registers and its workspace): the library orders a batch with beats behind
 * the beat batches it put on other streams before (events both ways; not under graph
 * capture); box / triangle batches carry no state and are not ordered.
 * The same holds for every entry point below that is described as
 * asynchronous (rt_dp_ray_prepare, rt_dp_knn_euclid16, rt_dp_knn_angular16,
 * rt_dp_knn_tree_euclid16, rt_dp_knn_tree_angular16); the synchronous ones wait for the whole device
 * before they read their arguments.
 * `hip_stream` may be a stream under capture (hipStreamBeginCapture): the batch
 * then becomes kernel and 16-byte copy nodes of the caller's graph, nothing is
 * allocated, and every replay evaluates whatever the same arrays hold by then.
 * The accumulator registers carry from replay to replay and to and from batches
 * launched directly, in the order the work executes; run one such batch directly
 * before capturing, so that the one-time set-up of the kernels is not captured.
 *
 * Failure that only shows after the kernels ran: the default Euclidean kernel (and the angular one on short segments)
 * hands the running sum of a vector pair that spans two workgroups from one to the next; the receiving workgroup
 * waits a bounded time (about a second).  Should that wait ever run out (a lost or starved workgroup), the beat
 * results of that batch are not valid.  The synchronous entry points (rt_dp_eval, rt_dp_eval_ctx, rt_dp_eval_bundles,
 * rt_dp_eval_bundles_device) notice it with the results they wait for anyway and evaluate the beat segments again
 * with kernels that need no hand-over: they never return RT_DP_OK with such results.  Behind rt_dp_launch the
 * library holds no batch to repeat: rt_dp_synchronize and rt_dp_get_accum return RT_DP_EHIP instead, once, for the
 * batches launched since the last of those calls; a caller of rt_dp_launch must look at one of them before it
 * trusts beat results (or launch with option "variant" 23, whose kernels have no hand-over).
 *
 * rt_dp_synchronize waits for every batch the context was given, on its own stream and (through the event
 * the library records behind them) on caller streams; a batch captured into a graph is the caller's to wait for. */
int rt_dp_launch(rt_dp_ctx *ctx, const rt_dp_op_batch *in, rt_dp_result_batch *out, void *hip_stream);
int rt_dp_synchronize(rt_dp_ctx *ctx);

/* Plain arrays <-> tiled buffer, for producers and consumers that keep separate arrays: `arrays` is a HOST array
 * of n_arrays (1..40) device pointers to n 32-bit words each, `tiled` a device buffer of
 * RT_DP_TILED_WORDS(n, n_arrays) words; word k of op i is arrays[k][i].  Asynchronous on `hip_stream` (NULL = the
 * context's stream), one pass over the data.  No counterpart in the reference.                              */
int rt_dp_tile_pack(rt_dp_ctx *ctx, const void *const *arrays, uint32_t n_arrays, uint64_t n, void *tiled, void *hip_stream);
int rt_dp_tile_unpack(rt_dp_ctx *ctx, const void *tiled, void *const *arrays, uint32_t n_arrays, uint64_t n, void *hip_stream);

/* The accumulator registers: {euclidean, angular dot, angular norm}.
 * rt_dp_reset is the RTL's reset (all three to +0).  These calls are ordered behind every batch the
 * context was given before them, including batches that rt_dp_launch put on a caller's stream (the
 * library records an event behind such a batch and waits for it here); the exception is a stream under
 * graph capture, where the caller orders the graph's execution against these calls itself.              */
int rt_dp_reset(rt_dp_ctx *ctx);
int rt_dp_get_accum(rt_dp_ctx *ctx, float acc[3]);
int rt_dp_set_accum(rt_dp_ctx *ctx, const float acc[3]);

/* ---- record-at-a-time image of the RTL ports --------------------------------
 * EnhancedInputBundle / EnhancedOutputBundle (IOSpec.scala:106-173) with every
 * FP field as its IEEE-754 bit pattern.  rt_dp_eval_bundles takes a stream of
 * input records in arrival order (any opcode mix) from host memory, copies it
 * to the GPU, evaluates it there (as rt_dp_eval_bundles_device does) and writes
 * one output record per input record in the same order.  Output fields that do not belong to the record's opcode are 0;
 * the two *_reset_accum outputs echo the input bit for every opcode
 * (UnifiedDatapath.scala:1205, :1254, :1267-1268); with support_euclidean == 0
 * the euclidean/angular outputs do not exist and are written as 0.             */
typedef struct rt_dp_f3 { uint32_t x, y, z; } rt_dp_f3;
typedef struct rt_dp_ray {
    rt_dp_f3 origin, dir, inv;
    uint32_t extent;
    uint32_t kx, ky, kz;
    rt_dp_f3 shear;
} rt_dp_ray;
typedef struct rt_dp_aabb { uint32_t x_min, y_min, z_min, x_max, y_max, z_max; } rt_dp_aabb;
typedef struct rt_dp_triangle { rt_dp_f3 A, B, C; } rt_dp_triangle;

typedef struct rt_dp_input_bundle {
    uint32_t       opcode;
    rt_dp_ray      ray;
    rt_dp_aabb     aabb[4];
    rt_dp_triangle triangle;
    uint32_t       euclidean_a[16];
    uint32_t       euclidean_b[16];
    uint32_t       euclidean_mask;
    uint32_t       euclidean_reset_accum;
} rt_dp_input_bundle;

typedef struct rt_dp_output_bundle {
    uint32_t opcode;
    uint32_t tmin_out[4];
    uint32_t isIntersect[4];
    uint32_t boxIndex[4];
    uint32_t t_num, t_denom;
    uint32_t triangle_hit;
    uint32_t euclidean_accumulator;
    uint32_t euclidean_reset_accum;
    uint32_t angular_dot_product;
    uint32_t angular_norm;
    uint32_t angular_reset_accum;
} rt_dp_output_bundle;

int rt_dp_eval_bundles(rt_dp_ctx *ctx, const rt_dp_input_bundle *in, rt_dp_output_bundle *out, uint64_t n);

/* The same contract for a record stream that already lives in HBM (device pointers,
 * n < 2^29); everything runs on the GPU and the records are read once, in place.  With
 * a 16-byte aligned input array (option "stream_mode" 0 / 2; the output array may be a
 * slice that starts at any record of a larger one) one kernel evaluates every record
 * where it lies and a second, small one adds the distance beats that continue a vector
 * pair from an earlier part of the stream onto that pair's running sum, in stream order;
 * the context keeps a workspace of about 12.6 bytes per record of a piece (pieces of at
 * most 2^26 records).  Otherwise ("stream_mode" 1) the records are partitioned by opcode
 * first (stable, so every accumulator sees its beats in stream order), evaluated by the
 * batch kernels and re-assembled.  Same bits either way.  Synchronous.                   */
int rt_dp_eval_bundles_device(rt_dp_ctx *ctx, const rt_dp_input_bundle *d_in, rt_dp_output_bundle *d_out, uint64_t n);

/* ---- one stream over several contexts / GPUs -----------------------------------------------
 * The RTL is one datapath fed by one producer and drained by one consumer thread
 * (src/test/scala/raytracer_datapath/Datapath_test.scala:476-496).  Here a stream may be cut into
 * parts that are evaluated independently, one host thread + one context (+ its stream) per GPU, with no
 * communication between them: box and triangle ops carry no state, and a distance beat carries the
 * accumulator registers on only when its reset bit is clear (UnifiedDatapath.scala:1133-1154, :1028-1064),
 * so every cut is placed where the registers hold +0 -- which is what a fresh (or reset) context starts
 * from.  The parts' results, put together in order, are then bit for bit those of the whole stream on one
 * context that started from reset registers.  Host arrays, no GPU involved; the planners are pure functions.
 *
 * rt_dp_shard_plan: an opcode-sorted batch with n[opcode] units per segment; euc_ctl / ang_ctl are the
 *   segments' control words (host memory; may be NULL for an empty segment).  out[r] receives part r's range
 *   [start, stop) of every segment: contiguous near-equal ranges, the cuts of the beat segments moved forward
 *   to the next vector-pair boundary (a part can be empty when one pair spans several nominal ranges).
 *   flags: RT_DP_F_BOX_TILED / RT_DP_F_TRI_TILED place that segment's cuts on multiples of RT_DP_TILE_OPS, so
 *   that a part of a tiled segment is itself a tiled segment: the words from start * W on (W words per op).
 * rt_dp_shard_plan_records: a record stream in arrival order (any opcode mix); cuts[0..world] receives
 *   ascending record positions, cuts[0] = 0, cuts[world] = n; cut r is the first position at or after
 *   n * r / world in front of which neither distance opcode has a vector pair under way.                   */
typedef struct rt_dp_shard_range {
    uint64_t start[4]; /* indexed by rt_dp_opcode */
    uint64_t stop[4];
} rt_dp_shard_range;

int rt_dp_shard_plan(const uint64_t n[4], const uint32_t *euc_ctl, const uint32_t *ang_ctl, uint32_t world, uint32_t flags,
                     rt_dp_shard_range *out /* [world] */);
int rt_dp_shard_plan_records(const rt_dp_input_bundle *records, uint64_t n, uint32_t world, uint64_t *cuts /* [world + 1] */);

/* ---- vector pairs <-> beats (the format step on either side of the distance ops) ---------
 * The reference's job splitters (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:
 * 213-269 Euclidean, :271-308 angular): a pair of vectors of `len` elements becomes ceil(len / width)
 * beats (width 16 for the Euclidean op, 8 for the angular op), the last one zero-padded, mask = the
 * low min(width, remaining) bits, reset bit on the last beat; an empty pair yields no beat.
 *   pair_offsets[n_pairs + 1]  element offset of every pair in a / b, ascending, [0] = 0
 *   beat_offsets[n_pairs + 1]  beat offset of every pair; rt_dp_beat_offsets_host fills it from
 *                              pair_offsets (host arrays) and returns the number of beats
 * rt_dp_pairs_to_beats writes the segment a beat kernel reads: out_a / out_b [n_beats][width],
 * out_ctl[n_beats]; rt_dp_beats_to_pairs picks, per pair, the value of its last beat out of a
 * result array (what the reference's tests read: Datapath_test.scala:287-290, :353-356), +0 for an
 * empty pair.  Device pointers (offsets 8-byte, the rest 4-byte aligned); asynchronous on
 * `hip_stream` (NULL = context stream).                                                         */
uint64_t rt_dp_beat_offsets_host(const uint64_t *pair_offsets, uint64_t n_pairs, uint32_t width, uint64_t *beat_offsets);
int rt_dp_pairs_to_beats(rt_dp_ctx *ctx, const float *a, const float *b, const uint64_t *pair_offsets, const uint64_t *beat_offsets,
                         uint64_t n_pairs, uint64_t n_beats, uint32_t width, float *out_a, float *out_b, uint32_t *out_ctl, void *hip_stream);
int rt_dp_beats_to_pairs(rt_dp_ctx *ctx, const float *values, const uint64_t *beat_offsets, uint64_t n_pairs, float *out_pair_values,
                         void *hip_stream);

/* ---- ray pre-processing (the step in front of the datapath) ---------------------------
 * Derives, on the GPU, the ray fields that the reference's software model computes from
 * (origin, dir) before it pokes a record: inv, kx/ky/kz and shear of SW_Ray
 * (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:36-82) and the constant
 * extent 4000 (:55, :91-92).  Divisions are IEEE-754 correctly rounded like the JVM's;
 * NaN results are written as 0x7FC00000.  Device pointers only; any output group may be
 * NULL (inv: all three, shear: all three).  k is packed as in rt_dp_tri_in.
 * Asynchronous on `hip_stream` (NULL = context stream).                                  */
typedef struct rt_dp_ray_prep {
    const float *dir[3];
    float       *inv[3];
    uint32_t    *k;
    float       *shear[3];
    float       *extent;
} rt_dp_ray_prep;

int rt_dp_ray_prepare(rt_dp_ctx *ctx, const rt_dp_ray_prep *rays, uint64_t n, void *hip_stream);

/* ---- k-nearest neighbours on the Euclidean distance op ------------------------------
 * No counterpart in the reference's sources: the search layer that the distance
 * opcodes exist for (hierarchical search, README.md:39-45 of the reference) is not part
 * of its tree.  Exact top-16 per query over `n_points` 16-dimensional points, the
 * distance being precisely the single-beat Euclidean op (mask 0xFFFF, reset 1;
 * UnifiedDatapath.scala:221-245 ... :1133-1154).  All pointers are device pointers;
 * points [n_points][16], queries [n_queries][16].  `splits` >= 1 cuts the points into
 * that many ranges evaluated by separate workgroups; out_dist / out_idx are
 * [splits][n_queries][16], each list ascending by (distance, index), where index =
 * index_base + position of the point (so shards of a larger set report global indices).
 * Unused slots hold (+Inf, 0xFFFFFFFF); a NaN distance never enters a list.  Partial lists
 * (splits, GPUs) are combined by a (distance, index) merge, which makes the final result
 * independent of the partition.  Asynchronous on `hip_stream` (NULL = context stream).   */
int rt_dp_knn_euclid16(rt_dp_ctx *ctx, const float *points, uint32_t n_points, const float *queries, uint32_t n_queries,
                       uint32_t index_base, uint32_t splits, float *out_dist, uint32_t *out_idx, void *hip_stream);

/* The same exhaustive search ranked by the angular op (UnifiedDatapath.scala:246-268, :420-453,
 * :1028-1064): per pair, dot = sum q_i c_i and norm = sum c_i^2 exactly as two 8-element beats
 * deliver them (mask 0xFF, reset on the second), then the step the RTL leaves to its consumer
 * (IOSpec.scala:21-23): score = dot / sqrt(norm), both IEEE correctly rounded.  Largest score
 * first, ties by lower index; a pair whose score is NaN (zero or NaN norm) is never listed.
 * out_neg_score receives MINUS the score (-0 as +0), ascending, so that lists merge by the same
 * (value, index) order as distances; unused slots hold (+Inf, 0xFFFFFFFF).  Needs the extended
 * datapath.  Layouts, splits and stream as for rt_dp_knn_euclid16.                          */
int rt_dp_knn_angular16(rt_dp_ctx *ctx, const float *points, uint32_t n_points, const float *queries, uint32_t n_queries,
                        uint32_t index_base, uint32_t splits, float *out_neg_score, uint32_t *out_idx, void *hip_stream);

/* The same search through a tree over the points (no counterpart in the reference, as
 * above).  The caller supplies a balanced partition of the points into n_leaves (a power of
 * two) leaves of leaf_size slots (a multiple of 4), stored leaf by leaf:
 *   points  [n_leaves * leaf_size][16]   pad slots hold +Inf in every coordinate
 *   index   [n_leaves * leaf_size]       the index reported for the slot, 0xFFFFFFFF for a pad
 *   leaf_lo / leaf_hi [n_leaves][16]     bounding box of the leaf's points; a dimension in
 *                                        which a point of the leaf is NaN must be (-Inf, +Inf)
 * (raytracer_amd/knn_tree.py builds this from a point set with a median-split k-d tree).
 * queries [n_queries][16]; home_leaf[n_queries] names, per query, the leaf from which its walk
 * starts (any value is accepted and reduced modulo n_leaves; the leaf the query falls into prunes best, and queries should be
 * sorted by it, because 64 consecutive queries walk together).  `splits` (a power of two,
 * <= n_leaves) spreads the leaves of one query over that many workgroups; bound[n_queries]
 * is scratch the call initialises.  out_dist / out_idx are [splits][n_queries][16] as for
 * rt_dp_knn_euclid16, and merge the same way.  Leaves are skipped only when a lower bound on
 * the distance to their box proves that none of their points can enter the list, so the
 * merged result is bit-identical to rt_dp_knn_euclid16 on the same points.  Device pointers;
 * asynchronous on `hip_stream` (NULL = context stream).                                  */
typedef struct rt_dp_knn_tree {
    uint32_t        struct_size;
    uint32_t        n_leaves;
    uint32_t        leaf_size;
    uint32_t        reserved;
    const float    *points;
    const uint32_t *index;
    const float    *leaf_lo;
    const float    *leaf_hi;
} rt_dp_knn_tree;

int rt_dp_knn_tree_euclid16(rt_dp_ctx *ctx, const rt_dp_knn_tree *tree, const float *queries, const uint32_t *home_leaf,
                            uint32_t n_queries, uint32_t splits, uint32_t *bound, float *out_dist, uint32_t *out_idx, void *hip_stream);

/* The angular search (rt_dp_knn_angular16's lists, bit for bit) through the same tree.  A leaf is skipped only when an
 * upper bound of every score that the datapath can compute for a point inside the leaf's box -- from the box's largest
 * possible dot product with the query and its smallest (largest, for a negative dot) possible norm, widened for every
 * rounding on the way -- stays below the 16th best score a query holds; leaves with unbounded boxes (NaN or infinite
 * coordinates) are always scanned.  Arguments as for rt_dp_knn_tree_euclid16; out_neg_score / out_idx [splits][n_queries][16]
 * as for rt_dp_knn_angular16.  d_leaves_scanned (device memory, 8-byte aligned, may be NULL) is incremented by the number
 * of leaves scanned, summed over the waves of 64 consecutive queries: n_leaves x waves means nothing was pruned.  How
 * much is pruned depends on the data: directions that cluster prune well, uniform directions in 16 dimensions hardly.
 * Needs the extended datapath.  Asynchronous on `hip_stream` (NULL = context stream).  No counterpart in the reference.  */
int rt_dp_knn_tree_angular16(rt_dp_ctx *ctx, const rt_dp_knn_tree *tree, const float *queries, const uint32_t *home_leaf, uint32_t n_queries,
                             uint32_t splits, uint32_t *bound, float *out_neg_score, uint32_t *out_idx, uint64_t *d_leaves_scanned, void *hip_stream);

/* The leaf-major form of the same search, for data whose neighbouring queries do not share
 * their candidate leaves (e.g. uniform points in 16 dimensions, where the walk above prunes
 * nothing per wave).  Same tree, same answers, one merged list per query:
 *   1. the walk above, limited to the `walk_leaves` leaves nearest to each query, yields an upper
 *      bound of each query's 16th distance;
 *   2. per leaf, the queries whose box bound does not exceed their upper bound are listed;
 *   3. per leaf and 64 listed queries the leaf's points are scanned; every (distance, index)
 *      within the query's upper bound is kept as a candidate;
 *   4. per query the 16 smallest candidates are selected; a query that filled its candidate slots
 *      repeats 2-4 with the (tighter) 16th distance among the candidates it did collect.
 * queries / home_leaf as above (sorted by home leaf).  out_dist / out_idx are [n_queries][16].
 * Synchronous; scratch (lists and candidates: 8 KiB per query plus 4 bytes per listed pair) grows
 * inside the context.  stats (may be NULL): {listed pairs, 64-query groups, slots per query,
 * rounds}.  No counterpart in the reference.                                               */
int rt_dp_knn_tree_grouped_euclid16(rt_dp_ctx *ctx, const rt_dp_knn_tree *tree, const float *queries, const uint32_t *home_leaf,
                                    uint32_t n_queries, uint32_t walk_leaves, float *out_dist, uint32_t *out_idx, uint64_t stats[4]);

/* ---- hit-mask compaction behind a Quadbox batch ------------------------------------------
 * No counterpart in the reference's sources: the consumer step of a traversal loop.  The Quadbox
 * output lists the children sorted by entry distance with the hits first (UnifiedDatapath.scala:
 * 1117-1132); this turns the results of n ops (device arrays, as rt_dp_launch wrote them) into one
 * dense list with an entry per intersected child: ops in batch order, within an op closest first.
 *   out_item[k]  = item[i], or i when item is NULL            (e.g. the ray of op i)
 *   out_child[k] = children[4 i + boxIndex], or boxIndex when children is NULL
 *   out_tmin[k]  = the child's entry distance (skipped when out_tmin is NULL)
 * *count receives the number of entries the batch has; at most `capacity` are written.  The list
 * is a pure function of the inputs (offsets come from ballots and a scan, not from atomics).
 * Device pointers, 4-byte aligned; synchronous.                                              */
int rt_dp_box_compact(rt_dp_ctx *ctx, const rt_dp_box_out *results, uint64_t n, const uint32_t *item, const uint32_t *children,
                      uint32_t *out_item, uint32_t *out_child, float *out_tmin, uint64_t capacity, uint64_t *count);

/* The same over the tiled result buffer of a Quadbox segment (RT_DP_F_BOX_TILED: what rt_dp_launch wrote through
 * box.tmin_out[0]).                                                                              */
int rt_dp_box_compact_tiled(rt_dp_ctx *ctx, const void *tiled_results, uint64_t n, const uint32_t *item, const uint32_t *children,
                            uint32_t *out_item, uint32_t *out_child, float *out_tmin, uint64_t capacity, uint64_t *count);

/* The same compaction with no allocation anywhere on the call path.  It has no counterpart in the
 * reference; it exists for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  The block counts and offsets that
 * rt_dp_box_compact keeps in a buffer the context grows live here in a fixed array inside the library
 * (768 KiB per GPU, held until the process ends), so n is limited to RT_DP_BOX_COMPACT_STATIC_MAX_OPS
 * per call (RT_DP_EINVAL beyond).  Arguments as for rt_dp_box_compact, except that the number of
 * entries goes to *d_count in DEVICE memory (8-byte aligned) and nothing is copied to the host: the
 * call only enqueues kernels on `hip_stream` (NULL = context stream) and returns; ordering against the
 * producer of the results is the caller's.  All calls on one GPU share the array: calls that overlap
 * on one GPU (two streams, two contexts, two threads) race; calls in sequence on one stream are safe,
 * and nothing in the array has to be cleared between them.                                        */
#define RT_DP_BOX_COMPACT_STATIC_MAX_OPS (1u << 24)
int rt_dp_box_compact_static(rt_dp_ctx *ctx, const rt_dp_box_out *results, uint64_t n, const uint32_t *item, const uint32_t *children,
                             uint32_t *out_item, uint32_t *out_child, float *out_tmin, uint64_t capacity, uint64_t *d_count,
                             void *hip_stream);

/* ---- closest-hit traversal on the Quadbox and triangle ops ----------------------------
 * No counterpart in the reference's sources, which end at the datapath: this is the loop a
 * consumer runs around it.  The ops are those of UnifiedDatapath.scala (children sorted by
 * entry distance with hits first, :1117-1132; triangle hit / t_num / t_denom, :1087-1116);
 * the ray fields are derived as by rt_dp_ray_prepare.
 *
 * bvh: a 4-wide bounding volume hierarchy, root = node 0.
 *   nodes [n_nodes][32] words: 0..23 the four child boxes (child b: min xyz at 6b, max xyz at
 *     6b + 3; an absent child carries the all-+Inf box of SW_Data.scala:106-113), 24..27 the
 *     four child references (uint32 images), 28..31 unused.  A reference is 0xFFFFFFFF (absent),
 *     or a node index, or 0x80000000 | first_triangle << 3 | (count - 1) for a leaf of 1..8
 *     consecutive triangles.  An inner child must have a larger index than its parent; this and
 *     the range of every reference are checked on the device before the traversal starts
 *     (RT_DP_EINVAL).
 *   tris [n_tris][12] words: A xyz, B xyz, C xyz, 3 unused.
 * Rays: origin[3], dir[3] as separate arrays of n_rays floats; `extent` is the initial ray
 * extent (the reference's model uses 4000).  All of these are device pointers, 16-byte aligned.
 *
 * Two walks, chosen by `flags`; both report, per ray, the smallest (t, triangle index) among
 * the triangle hits they evaluate, t = t_num / t_denom (IEEE division, -0 read as +0, a NaN t
 * ignored), or (+Inf, 0xFFFFFFFF).  Both are deterministic (independent of scheduling).
 *   0 (depth first): one lane walks one ray with a stack.  Pop (reference, entry distance);
 *     limit = min(extent, best t so far); drop the entry unless entry distance < limit; a leaf
 *     runs one triangle op per triangle, in order; a node runs one Quadbox op with extent =
 *     limit and pushes its hit children farthest first, so the closest is popped next.  At most
 *     64 pending entries per ray (RT_DP_EINVAL beyond: roughly 21 levels of 4-wide nodes).
 *   RT_DP_TRACE_WAVEFRONT (breadth first): level by level, every (ray, node) pair runs one
 *     Quadbox op with extent = min(extent, the ray's best t after the previous level); hit inner
 *     children form the next level (appended by ballot compaction of the sorted hit masks), hit
 *     leaves run their triangle ops.  Visits every node the ray's extent reaches: the engine for
 *     "all candidates" queries; work queues grow inside the context as needed.
 * The two can differ only where a triangle's t and its box's entry distance round to
 * different sides of each other.  stats (may be NULL) receives {levels (wavefront) or 0, box
 * ops, triangle ops, largest queue (wavefront) or 0}.  Synchronous.                          */
#define RT_DP_TRACE_WAVEFRONT 1u

typedef struct rt_dp_bvh4 {
    uint32_t     struct_size;
    uint32_t     n_nodes;
    uint32_t     n_tris;
    uint32_t     reserved;
    const float *nodes;
    const float *tris;
} rt_dp_bvh4;

int rt_dp_trace_closest(rt_dp_ctx *ctx, const rt_dp_bvh4 *bvh, const float *const origin[3], const float *const dir[3],
                        uint32_t n_rays, float extent, uint32_t flags, float *out_t, uint32_t *out_tri, uint64_t stats[4]);

/* ---- host-side builders for the structures above ---------------------------------------
 * No counterpart in the reference.  Plain host arrays in and out, no GPU involved; the caller
 * uploads the results.  Simple median splits along the widest dimension: index construction is
 * outside the timed query paths.
 *
 * rt_dp_knn_tree_leaves: the n_leaves (a power of two) a point set needs at this leaf_size.
 * rt_dp_knn_tree_build_host: fills points[n_leaves * leaf_size][16] (pad slots +Inf),
 *   index[n_leaves * leaf_size] (index_base + position in `points`, 0xFFFFFFFF for a pad),
 *   lo / hi[n_leaves][16], and (optionally, both or neither) the split planes in heap order:
 *   split_dim / split_val[n_leaves - 1], node 2^l - 1 + s for segment s of level l; a query goes
 *   right when its coordinate >= split_val.
 * rt_dp_knn_home_leaf_host: the leaf each query reaches by descending those planes.
 * rt_dp_bvh4_max_nodes: rows to allocate for nodes[][32].
 * rt_dp_bvh4_build_host: tris[n_tris][9] (A xyz, B xyz, C xyz) -> nodes, *n_nodes, the triangles in
 *   leaf order as [n_tris][12] and order[i] = the caller's index of stored triangle i.           */
uint32_t rt_dp_knn_tree_leaves(uint32_t n_points, uint32_t leaf_size);
int rt_dp_knn_tree_build_host(const float *points, uint32_t n_points, uint32_t leaf_size, uint32_t index_base, float *out_points,
                              uint32_t *out_index, float *out_lo, float *out_hi, uint32_t *out_split_dim, float *out_split_val);
int rt_dp_knn_home_leaf_host(const float *queries, uint32_t n_queries, uint32_t n_leaves, const uint32_t *split_dim, const float *split_val,
                             uint32_t *out_home);
uint32_t rt_dp_bvh4_max_nodes(uint32_t n_tris);
int rt_dp_bvh4_build_host(const float *tris, uint32_t n_tris, uint32_t leaf_size, float *out_nodes, uint32_t *out_n_nodes, float *out_tris,
                          uint32_t *out_order);

/* ---- the k-d index built on the GPU -------------------------------------------------------
 * No counterpart in the reference.  rt_dp_knn_tree_build_device is rt_dp_knn_tree_build_host on device
 * arrays: same leaf count (rt_dp_knn_tree_leaves), same layout of out_points / out_index / out_lo /
 * out_hi / out_split_dim / out_split_val (heap order; both split arrays or neither), the same
 * partition rule (every segment sorted along its widest finite dimension; all segments of a level in
 * one radix sort); ties keep their current order instead of the host builder's order by index, so the
 * two may store equal-coordinate points in different slots.  rt_dp_knn_home_leaf is
 * rt_dp_knn_home_leaf_host on device arrays.  points [n_points][16], queries [n_queries][16]; every
 * pointer a device pointer, 16-byte aligned rows.  The build is synchronous (scratch grows inside
 * the context); the home-leaf kernel is asynchronous on `hip_stream` (NULL = context stream).      */
int rt_dp_knn_tree_build_device(rt_dp_ctx *ctx, const float *points, uint32_t n_points, uint32_t leaf_size, uint32_t index_base,
                                float *out_points, uint32_t *out_index, float *out_lo, float *out_hi, uint32_t *out_split_dim,
                                float *out_split_val);
int rt_dp_knn_home_leaf(rt_dp_ctx *ctx, const float *queries, uint32_t n_queries, uint32_t n_leaves, const uint32_t *split_dim,
                        const float *split_val, uint32_t *out_home, void *hip_stream);

/* ---- the same hierarchy built on the GPU -------------------------------------------------
 * No counterpart in the reference.  A linear (LBVH-style) construction: the triangles are ordered
 * along a Morton curve through their centroids (radix sort), a node's range of that order is halved
 * twice where the highest differing bit of the codes flips, parts of at most leaf_size (1..8)
 * triangles become leaves; boxes are exact.  out_nodes must hold rt_dp_bvh4_max_nodes(n_tris) rows of
 * 32 words; *out_n_nodes (host) receives the rows in use.  tris [n_tris][9], out_tris [n_tris][12]
 * (leaf order), out_order[i] = the caller's index of stored triangle i: device pointers, out_nodes /
 * out_tris 16-byte aligned.  Synchronous; scratch grows inside the context.                    */
int rt_dp_bvh4_build_device(rt_dp_ctx *ctx, const float *tris, uint32_t n_tris, uint32_t leaf_size, float *out_nodes, uint32_t *out_n_nodes,
                            float *out_tris, uint32_t *out_order);

/* ---- tuning and diagnostics ---------------------------------------------------
 * Kernel-shape knobs (0 = library default), for A/B measurements; results do not
 * depend on them.  ops_per_lane: consecutive box / triangle ops one lane owns
 * (1, 2 or 4; lowered automatically when an array is not 8- / 16-byte aligned).
 * nontemporal: streaming loads and stores carry the `nt` cache hint (1 on, 2 off).
 * vec_tile:
 * nominal beats per workgroup for opcodes 2/3.  The same knobs
 * are read once per context from RT_DP_OPS_PER_LANE, RT_DP_NONTEMPORAL and
 * RT_DP_VEC_TILE.  No counterpart in the reference.                              */
int rt_dp_set_tuning(rt_dp_ctx *ctx, int ops_per_lane, int nontemporal, int vec_tile);
/* Further integer knobs by name: "vec_passes" (opcodes 2/3: 16-byte load pairs in flight per
 * lane and sub-tile, 2 / 4 / 8), "variant" (0 normal; 10 diagnostic: the box kernel's memory
 * pattern without its arithmetic, outputs meaningless; the beat kernels have two shapes with identical results -- "owning": a workgroup owns whole vector
 * pairs and searches for their boundaries; "chained": fixed tiles that hand their running sums on from one to the
 * next through a workspace inside the context -- and each opcode defaults to the one that measured faster (Euclidean
 * chained; angular owning, chained for segments of up to 2^23 beats); 23: owning for both (21 / 22: with one or the other form of its serial accumulation),
 * 27: chained for both, 28: the chained shape as a persistent workgroup whose waves split the work -- three load and
 * sum the next tiles while the fourth accumulates the current one -- for both (measured slower on MI355X, 0.69 against
 * 0.75 - 0.78 of the HBM roof for angular beats; an option, not a default); 24 / 25 diagnostic: chained without the
 * accumulation / without the LDS stage either, outputs meaningless).  "vec_tile" applies to the owning shape only.  Environment: RT_DP_VEC_PASSES,
 * RT_DP_VARIANT.  "wgs_per_cu" (resident 256-thread workgroups per CU of the box / triangle / beat
 * kernels: 0 the measured per-kernel default, 3..16, 99 as many as fit; environment RT_DP_WGS_PER_CU).
 * "stream_mode" selects how rt_dp_eval_bundles_device works: 1 partitions the
 * records by opcode into structure-of-arrays segments first, 2 evaluates them in place and then adds up
 * the distance beats in a second kernel, 3 does both in one pass with the running sums handed from wave to
 * wave (2 and 3 need a 16-byte aligned input record array; same results, 2 is the fastest on MI355X),
 * 0 (default) picks 2 when the alignment allows. */
int rt_dp_set_option(rt_dp_ctx *ctx, const char *key, int value);
/* Reads a knob back as the library resolved it: the keys above give the value in effect ("ops_per_lane",
 * "nontemporal" 0/1, "vec_tile" as requested or 512: it applies to the owning shape of the beat kernels only, whose angular default is 504), and "box_wgs_per_cu" / "tri_wgs_per_cu" give the
 * residency that box / triangle batches large enough for the cap run with, as the runtime's occupancy query
 * reports it for the launch configuration in use (0 = uncapped).  The cap is set through the amount of
 * dynamic LDS a launch asks for; the amount is verified against that query once per process, so a runtime
 * that hands out LDS differently shows up here instead of silently changing the residency.
 * "stream_waves_per_cu": the same for the evaluation kernel of rt_dp_eval_bundles_device ("stream_mode" 0 / 2), in waves
 * (its workgroups are single waves): 8 on MI355X; environment RT_DP_STREAM_WAVES asks for another number, 0 for no cap. */
int rt_dp_get_option(rt_dp_ctx *ctx, const char *key, int *value);
/* "vec_chain_giveups" (read only): hand-overs between workgroups of the beat kernels that were given up on after a
 * bounded wait, since the context was created.  "stream_giveups" (read only): the same count for the hand-overs
 * from wave to wave of stream_mode 3.  "chain_fallbacks" (read only): batches / record streams that a synchronous
 * entry point evaluated a second time, without hand-overs, because of such a give-up.  All three stay 0 on a
 * healthy device.  "debug_chain_skip" (write only, for tests): workgroup k of the chained beat kernels does not
 * publish its running sums and waits are cut to 2^12 polls, so the give-up path can be exercised; -1 turns it off. */

/* The kernel that rt_dp_launch / rt_dp_eval_ctx runs for the segment of `opcode` in this batch under the context's
 * present tuning (outside graph capture), named as a profiler prints it, e.g. "rtdp::k_box_tiled<true, false>"; an
 * empty string for an empty segment.  For harnesses that keep counter measurements per kernel and want to know
 * whether they still describe what the library picks.  No counterpart in the reference.                        */
int rt_dp_kernel_name(rt_dp_ctx *ctx, const rt_dp_op_batch *in, const rt_dp_result_batch *out, int opcode, char *buf, uint64_t len);

const char *rt_dp_last_error(void);
const char *rt_dp_version(void);
int         rt_dp_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_DP_H */
