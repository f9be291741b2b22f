// C ABI of librt_dp (see include/rt_dp.h): argument checking, contexts,
// accumulator registers, host-pointer staging and the record-at-a-time path.
#include "../../include/rt_dp.h"
#include "kernels.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

using namespace rtdp;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RT_DP_EHIP, "%s: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

constexpr uint64_t kMaxOps = 1ull << 40;

}  // namespace

namespace rtdp {
// for the other host-side sources of the library (shard.cpp): same per-thread message buffer as rt_dp_last_error()
int api_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace rtdp

struct rt_dp_ctx {
    int device = 0;
    bool extended = true;  // support_euclidean == Some(16)
    hipStream_t stream = nullptr;
    unsigned long long *h_words = nullptr;  // 16 pinned host words: small read-backs (flags, counts) land here without a staging copy; [15]: the give-up count
    // accumulator registers live in HBM; two copies so that the workgroup that
    // reads the incoming value never races the one that writes the outgoing one
    // The accumulator registers, [3][4] floats: {euclid, dot, norm, pad}; copy 2 is a call's backup.  A beat kernel reads its
    // opcode's register(s) from one of the copies 0 / 1 and leaves the new value in the other, so the reader of a launch never
    // races its writer; which copy is current is kept per opcode (a batch without beats of an opcode leaves that opcode's
    // register where it is, and nothing has to be copied to carry it along).
    float *d_state = nullptr;
    int cur_e = 0, cur_a = 0;                  // current copy of the Euclidean register / of the two angular registers
    bool pinned_e = false, pinned_a = false;   // a captured graph holds the address of that copy: the register stays there from then on
    // grow-only staging for host-pointer batches
    char *d_stage = nullptr;
    size_t stage_bytes = 0;
    char *h_stage = nullptr;  // pinned mirror of the front of d_stage, for small host-pointer batches
    // grow-only workspace of the device-side record stream path (slots, block offsets)
    char *d_stream = nullptr;
    size_t stream_bytes = 0;
    // staging of host-resident record streams (input records, then output records)
    char *d_records = nullptr;
    size_t records_bytes = 0;
    // traversal driver: ray records + best[] + counters in d_trace, three work queues
    char *d_trace = nullptr;
    size_t trace_bytes = 0;
    char *d_queue[3] = {nullptr, nullptr, nullptr};
    size_t queue_items[3] = {0, 0, 0};
    // leaf-major k-NN: per-query / per-leaf tables, the (leaf, query) lists, the candidate slots
    char *d_knn[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t knn_bytes[4] = {0, 0, 0, 0};
    // rt_dp_launch on a caller stream: this event is recorded behind the batch, and everything that touches the
    // accumulator registers on the context's own stream (or from the host) waits for it first
    hipEvent_t launched = nullptr;
    bool launched_pending = false;
    // hand-over words of the chained beat kernels (kernels.hip); grows, is zero when allocated and never cleared:
    // a word is valid for the launch whose epoch it carries.  A buffer that was outgrown may still be in use by a
    // launch on a caller's stream, so it is kept until the context goes.
    char *d_chain = nullptr;
    size_t chain_bytes = 0;
    uint32_t chain_epoch = 0;
    std::vector<char *> chain_retired;
    // Word 0 counts hand-overs that a chained beat kernel gave up on (a bounded wait ran out).  It lives apart from the
    // hand-over words, so the count survives their re-allocation.  Every entry point that waits for its results reads it
    // with them and, when it moved, evaluates the beat segments again in the shape that needs no hand-over.
    unsigned long long *d_chain_hdr = nullptr;
    unsigned long long chain_seen = 0;       // the count when it was last looked at
    unsigned long long chain_fallbacks = 0;  // batches whose beat segments were evaluated again after a give-up
    bool chain_dirty = false;                // a chained kernel was launched since the count was last looked at
    int debug_chain_skip = -1;               // test hook: this workgroup of the chained kernels does not publish (-1: off)
    // beat batches that rt_dp_launch put on the context's own stream and nobody has waited for yet: a batch on a caller
    // stream is ordered behind them through this event
    hipEvent_t own_done = nullptr;
    bool own_pending = false;
    Tuning tuning{0, 0, 0, 0, 0, 0};
    int stream_mode = 0;  // record streams in HBM: 0 automatic (2 when the arrays allow it), 1 partition by opcode first, 2 evaluate then fold, 3 single pass
    unsigned long long stream_giveups = 0;  // hand-overs of the single-pass record-stream kernel that were given up on (never, unless a wave is lost)
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess)) ok = true;
    }
    ~DeviceGuard() {
        int now = -1;
        if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    }
};

int check_batch(const rt_dp_op_batch *in, const rt_dp_result_batch *out) {
    if (!in || !out) return fail(RT_DP_ENULL, "rt_dp: batch pointer is NULL");
    if (in->struct_size != sizeof(rt_dp_op_batch)) return fail(RT_DP_EINVAL, "rt_dp_op_batch.struct_size is %u, expected %zu", in->struct_size, sizeof(rt_dp_op_batch));
    if (out->struct_size != sizeof(rt_dp_result_batch)) return fail(RT_DP_EINVAL, "rt_dp_result_batch.struct_size is %u, expected %zu", out->struct_size, sizeof(rt_dp_result_batch));
    if (in->flags & ~RT_DP_F_ALL_FLAGS) return fail(RT_DP_EINVAL, "unknown flag bits 0x%x", in->flags & ~RT_DP_F_ALL_FLAGS);
    for (int op = 0; op < 4; ++op)
        if (in->n[op] > kMaxOps) return fail(RT_DP_EINVAL, "n[%d] = %llu is not representable (limit 2^40)", op, (unsigned long long)in->n[op]);

#define NEED(ptr, what)                                                                            \
    do {                                                                                           \
        if (!(ptr)) return fail(RT_DP_ENULL, "%s is NULL for a non-empty segment", what);          \
        if (reinterpret_cast<uintptr_t>(ptr) & 3u) return fail(RT_DP_EALIGN, "%s is not 4-byte aligned", what); \
    } while (0)

    const bool derive = (in->flags & RT_DP_F_DERIVE_RAY) != 0;
    if (in->n[RT_DP_OP_TRIANGLE] && (in->flags & RT_DP_F_TRI_TILED)) {
        NEED(in->tri.org[0], "tri.org[0] (the tiled operand buffer)");
        NEED(out->tri.t_num, "result tri.t_num (the tiled result buffer)");
    } else if (in->n[RT_DP_OP_TRIANGLE]) {
        for (int a = 0; a < 3; ++a) {
            NEED(in->tri.org[a], "tri.org[]");
            if (derive) NEED(in->tri.dir[a], "tri.dir[]");
            else NEED(in->tri.shear[a], "tri.shear[]");
        }
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) NEED(in->tri.vtx[v][a], "tri.vtx[][]");
        if (!derive) NEED(in->tri.k, "tri.k");
        NEED(out->tri.t_num, "result tri.t_num");
        NEED(out->tri.t_denom, "result tri.t_denom");
        NEED(out->tri.hit, "result tri.hit");
    }
    if (in->n[RT_DP_OP_QUADBOX] && (in->flags & RT_DP_F_BOX_TILED)) {
        NEED(in->box.org[0], "box.org[0] (the tiled operand buffer)");
        NEED(out->box.tmin_out[0], "result box.tmin_out[0] (the tiled result buffer)");
    } else if (in->n[RT_DP_OP_QUADBOX]) {
        for (int a = 0; a < 3; ++a) {
            NEED(in->box.org[a], "box.org[]");
            NEED(in->box.dir[a], "box.dir[]");
            if (!derive) NEED(in->box.inv[a], "box.inv[]");
        }
        NEED(in->box.extent, "box.extent");
        for (int b = 0; b < 4; ++b)
            for (int f = 0; f < 6; ++f) NEED(in->box.aabb[b][f], "box.aabb[][]");
        for (int k = 0; k < 4; ++k) NEED(out->box.tmin_out[k], "result box.tmin_out[]");
        NEED(out->box.flags, "result box.flags");
    }
    if (in->n[RT_DP_OP_EUCLIDEAN]) {
        NEED(in->euc.a, "euc.a"); NEED(in->euc.b, "euc.b"); NEED(in->euc.ctl, "euc.ctl");
        NEED(out->euc.accumulator, "result euc.accumulator");
    }
    if (in->n[RT_DP_OP_ANGULAR]) {
        NEED(in->ang.a, "ang.a"); NEED(in->ang.b, "ang.b"); NEED(in->ang.ctl, "ang.ctl");
        NEED(out->ang.dot_product, "result ang.dot_product");
        NEED(out->ang.norm, "result ang.norm");
    }
#undef NEED
    if (in->flags & RT_DP_F_DEVICE_PTRS) {  // host arrays are staged into aligned device buffers
        if (in->n[RT_DP_OP_EUCLIDEAN] && ((reinterpret_cast<uintptr_t>(in->euc.a) | reinterpret_cast<uintptr_t>(in->euc.b)) & 15u))
            return fail(RT_DP_EALIGN, "euc.a / euc.b must be 16-byte aligned device pointers");
        if (in->n[RT_DP_OP_ANGULAR] && ((reinterpret_cast<uintptr_t>(in->ang.a) | reinterpret_cast<uintptr_t>(in->ang.b)) & 15u))
            return fail(RT_DP_EALIGN, "ang.a / ang.b must be 16-byte aligned device pointers");
    }
    return RT_DP_OK;
}

void fill_box(BoxArgs &a, const rt_dp_box_in &in, const rt_dp_box_out &out, bool derive = false) {
    int f = 0;
    for (int i = 0; i < 3; ++i) a.in[f++] = in.org[i];
    for (int i = 0; i < 3; ++i) a.in[f++] = in.dir[i];
    for (int i = 0; i < 3; ++i) a.in[f++] = derive ? nullptr : in.inv[i];
    a.in[f++] = in.extent;
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 6; ++k) a.in[f++] = in.aabb[b][k];
    for (int k = 0; k < 4; ++k) a.tmin[k] = out.tmin_out[k];
    a.flags = out.flags;
}

void fill_tri(TriArgs &a, const rt_dp_tri_in &in, const rt_dp_tri_out &out, bool derive = false) {
    int f = 0;
    for (int i = 0; i < 3; ++i) a.in[f++] = in.org[i];
    for (int i = 0; i < 3; ++i) a.in[f++] = derive ? in.dir[i] : in.shear[i];
    for (int v = 0; v < 3; ++v)
        for (int i = 0; i < 3; ++i) a.in[f++] = in.vtx[v][i];
    a.k = derive ? nullptr : in.k;
    a.t_num = out.t_num;
    a.t_denom = out.t_denom;
    a.hit = out.hit;
}

// Orders the context's own stream (and, with host_too, the calling thread) behind the last batch that was
// launched on a caller stream.
int join_caller_stream(rt_dp_ctx *ctx, bool host_too) {
    if (!ctx->launched_pending) return RT_DP_OK;
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->launched, 0));
    if (host_too) {
        HIP_TRY(hipEventSynchronize(ctx->launched));
        ctx->launched_pending = false;
    }
    return RT_DP_OK;
}

// Pinned landing place of small device-to-host read-backs (a copy into pageable memory goes through a staging buffer of the
// runtime and costs several microseconds more).
int pinned_words(rt_dp_ctx *ctx) {
    if (ctx->h_words) return RT_DP_OK;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_words), 16 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipHostMalloc(128) failed: %s", hipGetErrorString(e));
    return RT_DP_OK;
}

// Up to 64 bytes from device memory into *dst through the pinned words; synchronises s.
int read_back(rt_dp_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t s) {
    int rc = pinned_words(ctx);
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_words, src, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memcpy(dst, ctx->h_words, bytes);
    return RT_DP_OK;
}

// The chained beat kernels' workspace for a segment of n beats and a fresh epoch, on stream s.
int chain_for(rt_dp_ctx *ctx, uint64_t n, hipStream_t s, VecArgs &a) {
    const size_t need = vec_chain_bytes(n);
    if (need > ctx->chain_bytes) {
        const size_t want = need + need / 4;
        char *fresh = nullptr;
        hipError_t e = hipMalloc(&fresh, want);
        if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipMalloc(%zu) for the beat kernels' hand-over words failed: %s", want, hipGetErrorString(e));
        if (ctx->d_chain) ctx->chain_retired.push_back(ctx->d_chain);
        ctx->d_chain = fresh;
        ctx->chain_bytes = want;
        HIP_TRY(hipMemsetAsync(fresh, 0, want, s));
    }
    if (++ctx->chain_epoch == 0) {  // 2^32 launches later: start over with clean words
        HIP_TRY(hipMemsetAsync(ctx->d_chain, 0, ctx->chain_bytes, s));
        ctx->chain_epoch = 1;
    }
    a.chain = reinterpret_cast<unsigned long long *>(ctx->d_chain);
    a.chain_header = ctx->d_chain_hdr;
    a.epoch = ctx->chain_epoch;
    a.debug_skip = ctx->debug_chain_skip >= 0 ? (uint32_t)ctx->debug_chain_skip + 1u : 0u;
    ctx->chain_dirty = true;
    return RT_DP_OK;
}

// Enqueues, behind what was launched on s, the read-back of the give-up count into the pinned words.
int chain_probe(rt_dp_ctx *ctx, hipStream_t s) {
    if (!ctx->chain_dirty) return RT_DP_OK;
    int rc = pinned_words(ctx);
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_words + 15, ctx->d_chain_hdr, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    return RT_DP_OK;
}

// After s has been waited for: did a hand-over give up since the count was last looked at?
bool chain_moved(rt_dp_ctx *ctx) {
    if (!ctx->chain_dirty) return false;
    ctx->chain_dirty = false;
    const unsigned long long now = ctx->h_words[15];
    if (now == ctx->chain_seen) return false;
    ctx->chain_seen = now;
    return true;
}

// For callers that hold no batch to evaluate again (rt_dp_synchronize, rt_dp_get_accum): s is idle.
int chain_verdict(rt_dp_ctx *ctx, hipStream_t s) {
    if (!ctx->chain_dirty) return RT_DP_OK;
    int rc = chain_probe(ctx, s);
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    if (chain_moved(ctx))
        return fail(RT_DP_EHIP, "a beat kernel gave up waiting for the running sums of its predecessor (a workgroup was lost or starved): the Euclidean / angular "
                                "results and the accumulator registers of the batches launched since the last rt_dp_synchronize are not valid; "
                                "rt_dp_eval_ctx evaluates such a batch again by itself");
    return RT_DP_OK;
}

bool is_capturing(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cap != hipStreamCaptureStatusNone;
}

// A beat kernel read its opcode's register(s) (`words` floats at `slot`) from copy `cur` and left the new ones in the other
// copy: make that the current one.  Normally the two copies just swap roles; once a graph has captured their addresses the
// new values are copied back instead, so that replays and direct launches keep finding the register in one place.
int commit_state(rt_dp_ctx *ctx, int &cur, bool pinned, int slot, int words, hipStream_t s) {
    if (!pinned) { cur ^= 1; return RT_DP_OK; }
    HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * cur + slot, ctx->d_state + 4 * (cur ^ 1) + slot, words * sizeof(float), hipMemcpyDeviceToDevice, s));
    return RT_DP_OK;
}

// The beat segments of a batch (extended datapath).  owning_only: the kernel shape that needs no hand-over between
// workgroups, whatever the defaults say.
int launch_beats(rt_dp_ctx *ctx, const rt_dp_op_batch *in, const rt_dp_result_batch *out, hipStream_t s, bool owning_only) {
    const uint64_t n_euc = in->n[RT_DP_OP_EUCLIDEAN], n_ang = in->n[RT_DP_OP_ANGULAR];
    // (16-byte alignment of a / b: check_batch, before anything was enqueued)
    // A batch that is being captured into a graph runs again later with the arguments of today: the hand-over
    // words of the chained kernels are told apart by a per-launch number, which a replay would repeat, so the
    // captured kernels are the ones that own whole pairs and need no workspace; and the registers stay at the
    // address the graph holds.
    const bool captured = is_capturing(s);
    const bool no_chain = captured || owning_only;
    if (n_euc) {
        if (captured) ctx->pinned_e = true;
        VecArgs a{in->euc.a, in->euc.b, in->euc.ctl, out->euc.accumulator, nullptr, ctx->d_state + 4 * ctx->cur_e, ctx->d_state + 4 * (ctx->cur_e ^ 1)};
        int rc = no_chain ? RT_DP_OK : chain_for(ctx, n_euc, s, a);
        if (rc != RT_DP_OK) return rc;
        HIP_TRY(launch_euclid(a, n_euc, ctx->tuning, s));
        rc = commit_state(ctx, ctx->cur_e, ctx->pinned_e, 0, 1, s);
        if (rc != RT_DP_OK) return rc;
    }
    if (n_ang) {
        if (captured) ctx->pinned_a = true;
        VecArgs a{in->ang.a, in->ang.b, in->ang.ctl, out->ang.dot_product, out->ang.norm, ctx->d_state + 4 * ctx->cur_a + 1,
                  ctx->d_state + 4 * (ctx->cur_a ^ 1) + 1};
        int rc = no_chain ? RT_DP_OK : chain_for(ctx, n_ang, s, a);
        if (rc != RT_DP_OK) return rc;
        HIP_TRY(launch_angular(a, n_ang, ctx->tuning, s));
        rc = commit_state(ctx, ctx->cur_a, ctx->pinned_a, 1, 2, s);
        if (rc != RT_DP_OK) return rc;
    }
    return RT_DP_OK;
}

// The registers as they are before a batch, for the entry points that may have to evaluate its beat segments twice.  A
// beat kernel leaves the copy it read intact, so going back normally means pointing at that copy again; only a register
// whose place a captured graph has fixed is overwritten by the batch and needs the backup copy (d_state + 8).
struct RegSnapshot { int cur_e, cur_a; bool backup; };

int regs_snapshot(rt_dp_ctx *ctx, hipStream_t s, RegSnapshot &snap) {
    snap = {ctx->cur_e, ctx->cur_a, ctx->pinned_e || ctx->pinned_a};
    if (snap.backup) {
        HIP_TRY(hipMemcpyAsync(ctx->d_state + 8, ctx->d_state + 4 * ctx->cur_e, sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(ctx->d_state + 9, ctx->d_state + 4 * ctx->cur_a + 1, 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return RT_DP_OK;
}

int regs_restore(rt_dp_ctx *ctx, hipStream_t s, const RegSnapshot &snap) {
    ctx->cur_e = snap.cur_e;
    ctx->cur_a = snap.cur_a;
    if (snap.backup && ctx->pinned_e) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * ctx->cur_e, ctx->d_state + 8, sizeof(float), hipMemcpyDeviceToDevice, s));
    if (snap.backup && ctx->pinned_a) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * ctx->cur_a + 1, ctx->d_state + 9, 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return RT_DP_OK;
}

// A hand-over was given up on while the batch ran (s is idle): registers back to where they were, beat segments again
// in the owning shape.  The caller waits for s afterwards.
int beats_again(rt_dp_ctx *ctx, const rt_dp_op_batch *in, const rt_dp_result_batch *out, hipStream_t s, const RegSnapshot &snap) {
    int rc = regs_restore(ctx, s, snap);
    if (rc != RT_DP_OK) return rc;
    ++ctx->chain_fallbacks;
    return launch_beats(ctx, in, out, s, true);
}

// Enqueue the kernels of one batch whose arrays are all device pointers.
int launch_device(rt_dp_ctx *ctx, const rt_dp_op_batch *in, const rt_dp_result_batch *out, hipStream_t s) {
    const uint64_t n_tri = in->n[RT_DP_OP_TRIANGLE], n_box = in->n[RT_DP_OP_QUADBOX];
    const uint64_t n_euc = in->n[RT_DP_OP_EUCLIDEAN], n_ang = in->n[RT_DP_OP_ANGULAR];
    const bool derive = (in->flags & RT_DP_F_DERIVE_RAY) != 0;
    if (n_box && (in->flags & RT_DP_F_BOX_TILED)) {
        HIP_TRY(launch_box_tiled(in->box.org[0], reinterpret_cast<uint32_t *>(out->box.tmin_out[0]), n_box, ctx->tuning, s, derive));
    } else if (n_box) {
        BoxArgs a;
        fill_box(a, in->box, out->box, derive);
        HIP_TRY(launch_box(a, n_box, ctx->tuning, s, derive));
    }
    if (n_tri && (in->flags & RT_DP_F_TRI_TILED)) {
        HIP_TRY(launch_tri_tiled(in->tri.org[0], reinterpret_cast<uint32_t *>(out->tri.t_num), n_tri, ctx->tuning, s, derive));
    } else if (n_tri) {
        TriArgs a;
        fill_tri(a, in->tri, out->tri, derive);
        HIP_TRY(launch_tri(a, n_tri, ctx->tuning, s, derive));
    }
    if (!ctx->extended) {
        // baseline datapath: opcodes 2/3 flow through untouched, outputs stay 0 and the
        // accumulators are not part of the design (UnifiedDatapath.scala:224, :247)
        if (n_euc) HIP_TRY(hipMemsetAsync(out->euc.accumulator, 0, n_euc * sizeof(float), s));
        if (n_ang) {
            HIP_TRY(hipMemsetAsync(out->ang.dot_product, 0, n_ang * sizeof(float), s));
            HIP_TRY(hipMemsetAsync(out->ang.norm, 0, n_ang * sizeof(float), s));
        }
        return RT_DP_OK;
    }
    if (n_euc || n_ang) return launch_beats(ctx, in, out, s, false);
    return RT_DP_OK;
}

struct Span { const void *src; void *dst_host; size_t bytes; size_t off; };

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

constexpr size_t kPinnedStage = size_t(8) << 20;  // host-pointer batches up to this size travel through pinned memory

// Host-pointer batch: stage through the context's device buffer.
int eval_host(rt_dp_ctx *ctx, const rt_dp_op_batch *in, rt_dp_result_batch *out) {
    std::vector<Span> ins, outs;
    size_t total = 0;
    rt_dp_op_batch din = *in;
    rt_dp_result_batch dout = *out;
    din.flags |= RT_DP_F_DEVICE_PTRS;

    // inputs first, outputs behind them: each side is one contiguous range of the staging buffer
    size_t in_total = 0, out_total = 0;
    auto add_in = [&](const void *host, size_t bytes) -> size_t { ins.push_back({host, nullptr, bytes, in_total}); size_t o = in_total; in_total += align_up(bytes); return o; };
    auto add_out = [&](void *host, size_t bytes) -> size_t { outs.push_back({nullptr, host, bytes, out_total}); size_t o = out_total; out_total += align_up(bytes); return o; };

    std::vector<std::pair<const void **, size_t>> fix_in;  // (field address, offset)
    std::vector<std::pair<void **, size_t>> fix_out;
#define IN_ARR(field, count, elt) fix_in.push_back({reinterpret_cast<const void **>(&(field)), add_in((field), (size_t)(count) * (elt))})
#define OUT_ARR(field, count, elt) fix_out.push_back({reinterpret_cast<void **>(&(field)), add_out((field), (size_t)(count) * (elt))})

    const uint64_t n_tri = in->n[RT_DP_OP_TRIANGLE], n_box = in->n[RT_DP_OP_QUADBOX];
    const uint64_t n_euc = in->n[RT_DP_OP_EUCLIDEAN], n_ang = in->n[RT_DP_OP_ANGULAR];
    const bool derive = (in->flags & RT_DP_F_DERIVE_RAY) != 0;
    // A tiled segment is one buffer each way.  The operands are copied whole (the buffer spans whole tiles); of the results
    // everything up to the last tile is copied back in one piece and the last tile word by word, so that what lies behind
    // op n - 1 in the caller's buffer stays as it was.
    auto tiled_out = [&](void **field, void *host, uint64_t n, uint32_t words) {
        const uint64_t full = (n - 1) / kTileOps, rem = n - full * kTileOps;  // tiles in front of the last one, ops in the last one
        char *h = static_cast<char *>(host);
        const size_t base = out_total;
        fix_out.push_back({field, base});
        if (full) outs.push_back({nullptr, h, (size_t)(full * words * kTileOps * 4), base});
        for (uint32_t w = 0; w < words; ++w) {
            const size_t o = (size_t)((full * words + w) * kTileOps * 4);
            outs.push_back({nullptr, h + o, (size_t)(rem * 4), base + o});
        }
        out_total += align_up((size_t)((full + 1) * words * kTileOps * 4));
    };
    if (n_tri && (in->flags & RT_DP_F_TRI_TILED)) {
        IN_ARR(din.tri.org[0], RT_DP_TILED_WORDS(n_tri, RT_DP_TRI_TILE_IN_WORDS), 4);
        tiled_out(reinterpret_cast<void **>(&dout.tri.t_num), out->tri.t_num, n_tri, RT_DP_TRI_TILE_OUT_WORDS);
    } else if (n_tri) {
        for (int a = 0; a < 3; ++a) {
            IN_ARR(din.tri.org[a], n_tri, 4);
            if (derive) IN_ARR(din.tri.dir[a], n_tri, 4);
            else IN_ARR(din.tri.shear[a], n_tri, 4);
        }
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) IN_ARR(din.tri.vtx[v][a], n_tri, 4);
        if (!derive) IN_ARR(din.tri.k, n_tri, 4);
        OUT_ARR(dout.tri.t_num, n_tri, 4); OUT_ARR(dout.tri.t_denom, n_tri, 4); OUT_ARR(dout.tri.hit, n_tri, 4);
    }
    if (n_box && (in->flags & RT_DP_F_BOX_TILED)) {
        IN_ARR(din.box.org[0], RT_DP_TILED_WORDS(n_box, RT_DP_BOX_TILE_IN_WORDS), 4);
        tiled_out(reinterpret_cast<void **>(&dout.box.tmin_out[0]), out->box.tmin_out[0], n_box, RT_DP_BOX_TILE_OUT_WORDS);
    } else if (n_box) {
        for (int a = 0; a < 3; ++a) {
            IN_ARR(din.box.org[a], n_box, 4);
            IN_ARR(din.box.dir[a], n_box, 4);
            if (!derive) IN_ARR(din.box.inv[a], n_box, 4);
        }
        IN_ARR(din.box.extent, n_box, 4);
        for (int b = 0; b < 4; ++b)
            for (int f = 0; f < 6; ++f) IN_ARR(din.box.aabb[b][f], n_box, 4);
        for (int k = 0; k < 4; ++k) OUT_ARR(dout.box.tmin_out[k], n_box, 4);
        OUT_ARR(dout.box.flags, n_box, 4);
    }
    if (n_euc) {
        IN_ARR(din.euc.a, n_euc * 16, 4); IN_ARR(din.euc.b, n_euc * 16, 4); IN_ARR(din.euc.ctl, n_euc, 4);
        OUT_ARR(dout.euc.accumulator, n_euc, 4);
    }
    if (n_ang) {
        IN_ARR(din.ang.a, n_ang * 8, 4); IN_ARR(din.ang.b, n_ang * 8, 4); IN_ARR(din.ang.ctl, n_ang, 4);
        OUT_ARR(dout.ang.dot_product, n_ang, 4); OUT_ARR(dout.ang.norm, n_ang, 4);
    }
#undef IN_ARR
#undef OUT_ARR
    total = in_total + out_total;
    if (total == 0) return RT_DP_OK;
    for (auto &s : outs) s.off += in_total;
    for (auto &f : fix_out) f.second += in_total;

    if (total > ctx->stage_bytes) {
        if (ctx->d_stage) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->d_stage)); ctx->d_stage = nullptr; ctx->stage_bytes = 0; }
        size_t want = total + total / 4;
        if (want < kPinnedStage) want = kPinnedStage;
        hipError_t e = hipMalloc(&ctx->d_stage, want);
        if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipMalloc(%zu) for staging failed: %s", want, hipGetErrorString(e));
        ctx->stage_bytes = want;
    }
    for (auto &f : fix_in) *f.first = ctx->d_stage + f.second;
    for (auto &f : fix_out) *f.first = ctx->d_stage + f.second;

    // Small batches (the testbench's 100-op suites): dozens of tiny pageable copies cost more than the work.
    // Gather the operands in a pinned mirror, move each side with one copy.
    const bool pinned = total <= kPinnedStage;
    if (pinned && !ctx->h_stage) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage), kPinnedStage, hipHostMallocDefault);
        if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipHostMalloc(%zu) for staging failed: %s", kPinnedStage, hipGetErrorString(e));
    }
    if (pinned) {
        for (auto &s : ins) memcpy(ctx->h_stage + s.off, s.src, s.bytes);
        if (in_total) HIP_TRY(hipMemcpyAsync(ctx->d_stage, ctx->h_stage, in_total, hipMemcpyHostToDevice, ctx->stream));
    } else {
        for (auto &s : ins) {
            hipError_t e = hipMemcpyAsync(ctx->d_stage + s.off, s.src, s.bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(ctx->stream);
                return fail(RT_DP_EHIP, "staging copy failed: %s", hipGetErrorString(e));
            }
        }
    }
    RegSnapshot snap;
    int rc = regs_snapshot(ctx, ctx->stream, snap);
    if (rc == RT_DP_OK) rc = launch_device(ctx, &din, &dout, ctx->stream);
    if (rc == RT_DP_OK) rc = chain_probe(ctx, ctx->stream);
    if (rc != RT_DP_OK) {
        (void)hipStreamSynchronize(ctx->stream);  // the staged copies must not outlive the call: the next one reuses the buffers
        return rc;
    }
    for (int round = 0; round < 2; ++round) {  // a second round only behind a give-up of the chained beat kernels
        if (pinned) {
            if (out_total) HIP_TRY(hipMemcpyAsync(ctx->h_stage + in_total, ctx->d_stage + in_total, out_total, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            for (auto &s : outs) HIP_TRY(hipMemcpyAsync(s.dst_host, ctx->d_stage + s.off, s.bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->own_pending = false;
        if (round == 1 || !chain_moved(ctx)) break;
        rc = beats_again(ctx, &din, &dout, ctx->stream, snap);
        if (rc != RT_DP_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    }
    if (pinned)
        for (auto &s : outs) memcpy(s.dst_host, ctx->h_stage + s.off, s.bytes);
    return RT_DP_OK;
}

// rt_dp_eval's per-thread default context; destroyed when its thread ends
struct DefaultCtx {
    rt_dp_ctx *p = nullptr;
    ~DefaultCtx() {
        if (p) rt_dp_destroy(p);
    }
};
thread_local DefaultCtx g_default_ctx;

}  // namespace

extern "C" {

const char *rt_dp_last_error(void) { return g_err; }

const char *rt_dp_version(void) { return "rt_dp 0.1 (gfx950)"; }

int rt_dp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_dp_create(const rt_dp_config *cfg, rt_dp_ctx **out) {
    if (!out) return fail(RT_DP_ENULL, "rt_dp_create: out is NULL");
    *out = nullptr;
    rt_dp_config c{sizeof(rt_dp_config), -1, 0, 1, 16, 0};
    if (cfg) {
        if (cfg->struct_size != sizeof(rt_dp_config)) return fail(RT_DP_EINVAL, "rt_dp_config.struct_size is %u, expected %zu", cfg->struct_size, sizeof(rt_dp_config));
        c = *cfg;
    }
    // elaboration-time asserts of the RTL (UnifiedDatapath.scala:36-47)
    if (c.io_recorded_float != 0) return fail(RT_DP_EINVAL, "we don't support recorded-format Input/Output");
    if (c.internal_recorded_float != 1) return fail(RT_DP_EINVAL, "we don't support standard-format intermediates");
    if (c.support_euclidean != 0 && c.support_euclidean != 16)
        return fail(RT_DP_EINVAL, "raytracer currently only support processing 16-element vector pairs for euclidean distance calculation");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_DP_ENODEV, "no HIP device available");
    int dev = c.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(RT_DP_ENODEV, "hipGetDevice failed");
    if (dev >= ndev) return fail(RT_DP_ENODEV, "device %d out of range (%d devices)", dev, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_DP_ENODEV, "device %d is %s; librt_dp carries gfx950 code only", dev, prop.gcnArchName);

    DeviceGuard guard(dev);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", dev);
    rt_dp_ctx *ctx = new (std::nothrow) rt_dp_ctx;
    if (!ctx) return fail(RT_DP_ENOMEM, "out of host memory");
    ctx->device = dev;
    ctx->extended = (c.support_euclidean == 16);
    ctx->tuning.ops_per_lane = env_int("RT_DP_OPS_PER_LANE", 0);
    ctx->tuning.nontemporal = env_int("RT_DP_NONTEMPORAL", 0);
    ctx->tuning.vec_tile = env_int("RT_DP_VEC_TILE", 0);
    ctx->tuning.variant = env_int("RT_DP_VARIANT", 0);
    ctx->tuning.vec_passes = env_int("RT_DP_VEC_PASSES", 0);
    ctx->tuning.wgs_per_cu = env_int("RT_DP_WGS_PER_CU", 0);
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->launched, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->own_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_chain_hdr, 256);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_chain_hdr, 0, 256, ctx->stream);
    if (e == hipSuccess) e = compact_static_init(dev);  // address of this GPU's copy of the allocation-free entry point's scratch
    if (e == hipSuccess) e = hipMalloc(&ctx->d_state, 12 * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_state, 0, 12 * sizeof(float), ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        int rc = fail(RT_DP_EHIP, "context set-up failed: %s", hipGetErrorString(e));
        rt_dp_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return RT_DP_OK;
}

void rt_dp_destroy(rt_dp_ctx *ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    if (ctx->launched_pending) (void)hipEventSynchronize(ctx->launched);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->launched) (void)hipEventDestroy(ctx->launched);
    if (ctx->own_done) (void)hipEventDestroy(ctx->own_done);
    if (ctx->d_chain_hdr) (void)hipFree(ctx->d_chain_hdr);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_stream) (void)hipFree(ctx->d_stream);
    if (ctx->d_records) (void)hipFree(ctx->d_records);
    if (ctx->d_trace) (void)hipFree(ctx->d_trace);
    for (char *q : ctx->d_queue)
        if (q) (void)hipFree(q);
    for (char *q : ctx->d_knn)
        if (q) (void)hipFree(q);
    if (ctx->d_state) (void)hipFree(ctx->d_state);
    if (ctx->d_chain) (void)hipFree(ctx->d_chain);
    for (char *old_chain : ctx->chain_retired) (void)hipFree(old_chain);
    if (ctx->h_words) (void)hipHostFree(ctx->h_words);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (g_default_ctx.p == ctx) g_default_ctx.p = nullptr;
    delete ctx;
}

int rt_dp_launch(rt_dp_ctx *ctx, const rt_dp_op_batch *in, rt_dp_result_batch *out, void *hip_stream) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_launch: ctx is NULL");
    int rc = check_batch(in, out);
    if (rc != RT_DP_OK) return rc;
    if (!(in->flags & RT_DP_F_DEVICE_PTRS)) return fail(RT_DP_EINVAL, "rt_dp_launch needs RT_DP_F_DEVICE_PTRS; use rt_dp_eval_ctx for host arrays");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    const bool beats = ctx->extended && (in->n[RT_DP_OP_EUCLIDEAN] || in->n[RT_DP_OP_ANGULAR]);
    if (s == ctx->stream) {
        rc = join_caller_stream(ctx, false);
        if (rc != RT_DP_OK) return rc;
    } else if (beats && ctx->own_pending && !is_capturing(s)) {
        // beat batches are still queued on the context's own stream: this one shares their registers and hand-over
        // words, so it goes behind them
        HIP_TRY(hipEventRecord(ctx->own_done, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(s, ctx->own_done, 0));
        ctx->own_pending = false;
    }
    rc = launch_device(ctx, in, out, s);
    if (rc != RT_DP_OK) return rc;
    if (s == ctx->stream && beats) ctx->own_pending = true;
    if (s != ctx->stream && beats) {
        // the registers were read and written on the caller's stream: remember where, unless the stream is being
        // captured into a graph (an event recorded there cannot be waited for from outside the graph)
        if (!is_capturing(s)) {
            HIP_TRY(hipEventRecord(ctx->launched, s));
            ctx->launched_pending = true;
        }
    }
    return RT_DP_OK;
}

static int tile_args(const char *who, rt_dp_ctx *ctx, const void *const *arrays, uint32_t n_arrays, uint64_t n, const void *tiled, TileList &l) {
    if (!ctx) return fail(RT_DP_ENULL, "%s: ctx is NULL", who);
    if (n_arrays < 1 || n_arrays > (uint32_t)kTileMaxArrays) return fail(RT_DP_EINVAL, "%s: n_arrays must be 1..%d", who, kTileMaxArrays);
    if (n > kMaxOps) return fail(RT_DP_EINVAL, "%s: n = %llu is not representable (limit 2^40)", who, (unsigned long long)n);
    if (n == 0) return RT_DP_OK;
    if (!arrays || !tiled) return fail(RT_DP_ENULL, "%s: NULL argument", who);
    if (reinterpret_cast<uintptr_t>(tiled) & 3u) return fail(RT_DP_EALIGN, "%s: the tiled buffer is not 4-byte aligned", who);
    for (uint32_t k = 0; k < n_arrays; ++k) {
        if (!arrays[k]) return fail(RT_DP_ENULL, "%s: arrays[%u] is NULL", who, k);
        if (reinterpret_cast<uintptr_t>(arrays[k]) & 3u) return fail(RT_DP_EALIGN, "%s: arrays[%u] is not 4-byte aligned", who, k);
        l.a[k] = static_cast<uint32_t *>(const_cast<void *>(arrays[k]));
    }
    for (uint32_t k = n_arrays; k < (uint32_t)kTileMaxArrays; ++k) l.a[k] = nullptr;
    return RT_DP_OK;
}

int rt_dp_tile_pack(rt_dp_ctx *ctx, const void *const *arrays, uint32_t n_arrays, uint64_t n, void *tiled, void *hip_stream) {
    TileList l;
    int rc = tile_args("rt_dp_tile_pack", ctx, arrays, n_arrays, n, tiled, l);
    if (rc != RT_DP_OK || n == 0) return rc;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(launch_tile_pack(l, (int)n_arrays, n, static_cast<uint32_t *>(tiled), hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

int rt_dp_tile_unpack(rt_dp_ctx *ctx, const void *tiled, void *const *arrays, uint32_t n_arrays, uint64_t n, void *hip_stream) {
    TileList l;
    int rc = tile_args("rt_dp_tile_unpack", ctx, arrays, n_arrays, n, tiled, l);
    if (rc != RT_DP_OK || n == 0) return rc;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(launch_tile_unpack(static_cast<const uint32_t *>(tiled), (int)n_arrays, n, l, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

int rt_dp_synchronize(rt_dp_ctx *ctx) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_synchronize: ctx is NULL");
    DeviceGuard guard(ctx->device);
    int rc = join_caller_stream(ctx, true);  // batches that rt_dp_launch put on a caller stream count as well
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->own_pending = false;
    return chain_verdict(ctx, ctx->stream);
}

int rt_dp_eval_ctx(rt_dp_ctx *ctx, const rt_dp_op_batch *in, rt_dp_result_batch *out) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_eval_ctx: ctx is NULL");
    int rc = check_batch(in, out);
    if (rc != RT_DP_OK) return rc;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    if (in->flags & RT_DP_F_DEVICE_PTRS) {
        // The operands may still be in production on some other stream of the caller; the
        // synchronous entry point waits for the whole device before it reads them.
        HIP_TRY(hipDeviceSynchronize());
        ctx->launched_pending = false;  // whatever ran on a caller stream has finished
        ctx->own_pending = false;
        rc = chain_verdict(ctx, ctx->stream);  // batches launched asynchronously before this one: their failure is theirs
        if (rc != RT_DP_OK) return rc;
        RegSnapshot snap;
        rc = regs_snapshot(ctx, ctx->stream, snap);
        if (rc == RT_DP_OK) rc = launch_device(ctx, in, out, ctx->stream);
        if (rc == RT_DP_OK) rc = chain_probe(ctx, ctx->stream);
        if (rc != RT_DP_OK) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (chain_moved(ctx)) {  // a hand-over between workgroups was given up on: the same beats again, without hand-overs
            rc = beats_again(ctx, in, out, ctx->stream, snap);
            if (rc != RT_DP_OK) return rc;
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        return RT_DP_OK;
    }
    rc = join_caller_stream(ctx, false);
    if (rc != RT_DP_OK) return rc;
    return eval_host(ctx, in, out);
}

int rt_dp_eval(const rt_dp_op_batch *in, rt_dp_result_batch *out) {
    if (!g_default_ctx.p) {
        int rc = rt_dp_create(nullptr, &g_default_ctx.p);
        if (rc != RT_DP_OK) return rc;
    }
    return rt_dp_eval_ctx(g_default_ctx.p, in, out);
}

int rt_dp_reset(rt_dp_ctx *ctx) {
    const float z[3] = {0.0f, 0.0f, 0.0f};
    return rt_dp_set_accum(ctx, z);
}

int rt_dp_get_accum(rt_dp_ctx *ctx, float acc[3]) {
    if (!ctx || !acc) return fail(RT_DP_ENULL, "rt_dp_get_accum: NULL argument");
    DeviceGuard guard(ctx->device);
    int rc = join_caller_stream(ctx, true);
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->own_pending = false;
    rc = chain_verdict(ctx, ctx->stream);
    if (rc != RT_DP_OK) return rc;
    float h[8];
    HIP_TRY(hipMemcpy(h, ctx->d_state, sizeof h, hipMemcpyDeviceToHost));
    acc[0] = h[4 * ctx->cur_e];
    acc[1] = h[4 * ctx->cur_a + 1];
    acc[2] = h[4 * ctx->cur_a + 2];
    return RT_DP_OK;
}

int rt_dp_set_accum(rt_dp_ctx *ctx, const float acc[3]) {
    if (!ctx || !acc) return fail(RT_DP_ENULL, "rt_dp_set_accum: NULL argument");
    DeviceGuard guard(ctx->device);
    int rc = join_caller_stream(ctx, true);
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const float v[8] = {acc[0], acc[1], acc[2], 0.0f, acc[0], acc[1], acc[2], 0.0f};  // both copies: whichever is current per opcode
    HIP_TRY(hipMemcpy(ctx->d_state, v, sizeof v, hipMemcpyHostToDevice));
    return RT_DP_OK;
}

int rt_dp_set_tuning(rt_dp_ctx *ctx, int ops_per_lane, int nontemporal, int vec_tile) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_set_tuning: ctx is NULL");
    if (ops_per_lane != 0 && ops_per_lane != 1 && ops_per_lane != 2 && ops_per_lane != 4) return fail(RT_DP_EINVAL, "ops_per_lane must be 0, 1, 2 or 4");
    if (vec_tile < 0) return fail(RT_DP_EINVAL, "vec_tile must not be negative");
    ctx->tuning.ops_per_lane = ops_per_lane;
    if (nontemporal < 0 || nontemporal > 2) return fail(RT_DP_EINVAL, "nontemporal must be 0 (default), 1 (on) or 2 (off)");
    ctx->tuning.nontemporal = nontemporal;
    ctx->tuning.vec_tile = vec_tile;
    return RT_DP_OK;
}

// Record stream resident in HBM: partition, evaluate and re-assemble on the device.
// ordered: the records were produced on the context's own stream (no wait for the rest of the device needed).
// pinned_out: when the fused path runs, the output records are also copied to this pinned host buffer in front of the
// call's one synchronisation, and *copied says so.
static int bundles_device(rt_dp_ctx *ctx, const rt_dp_input_bundle *d_in, rt_dp_output_bundle *d_out, uint64_t n, bool ordered, void *pinned_out,
                          bool *copied);

int rt_dp_eval_bundles_device(rt_dp_ctx *ctx, const rt_dp_input_bundle *d_in, rt_dp_output_bundle *d_out, uint64_t n) {
    return bundles_device(ctx, d_in, d_out, n, false, nullptr, nullptr);
}

static int bundles_device(rt_dp_ctx *ctx, const rt_dp_input_bundle *d_in, rt_dp_output_bundle *d_out, uint64_t n, bool ordered, void *pinned_out,
                          bool *copied) {
    if (copied) *copied = false;
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_eval_bundles_device: ctx is NULL");
    if (n == 0) return RT_DP_OK;
    if (!d_in || !d_out) return fail(RT_DP_ENULL, "rt_dp_eval_bundles_device: in/out is NULL");
    if (n >= (1ull << 29)) return fail(RT_DP_EINVAL, "rt_dp_eval_bundles_device: at most 2^29 - 1 records per call");
    if ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 3u) return fail(RT_DP_EALIGN, "record arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    hipStream_t s = ctx->stream;
    if (ordered) {  // the records are on the context's stream; the registers may have been left by a batch on a caller stream
        int rcj = join_caller_stream(ctx, false);
        if (rcj != RT_DP_OK) return rcj;
    } else {
        HIP_TRY(hipDeviceSynchronize());  // the records may still be in production on a caller stream
        ctx->launched_pending = false;
        ctx->own_pending = false;
    }

    auto grow = [&](char *&buf, size_t &have, size_t want) -> int {
        if (want <= have) return RT_DP_OK;
        if (buf) { (void)hipFree(buf); buf = nullptr; have = 0; }
        want += want / 4;
        hipError_t e = hipMalloc(&buf, want);
        if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        have = want;
        return RT_DP_OK;
    };
    // the in-place paths read the records in 16-byte pieces; the output records may start anywhere (a slice of a larger array
    // starts at a multiple of 84 bytes)
    const bool aligned16 = (reinterpret_cast<uintptr_t>(d_in) & 15u) == 0;
    if ((ctx->stream_mode == 2 || ctx->stream_mode == 3) && !aligned16)
        return fail(RT_DP_EALIGN, "the in-place record-stream paths need a 16-byte aligned input record array");
    if (ctx->stream_mode == 3) {
        // ONE pass over the records: every wave evaluates its 64 records and adds up its distance beats in stream order,
        // the running sums travel from wave to wave (stream_fused.hip: k_stream_chain).  Pieces of 2^26 records bound
        // the workspace (24 bytes per 64 records of a piece).  Same bits as the two-pass form below, but slower on
        // MI355X (7.6 against 13.4 G records/s at 2^24 records: the look-back is one more memory round trip in the life
        // of a wave, and only 7 waves fit a CU next to their 21 KiB of staged records), so it is never chosen by itself.
        const int lg = env_int("RT_DP_STREAM_CHUNK_LOG2", 26);
        const uint64_t chunk = 1ull << (lg < 11 ? 11 : (lg > 30 ? 30 : lg));
        const uint64_t first = n < chunk ? n : chunk;
        int rc = grow(ctx->d_stream, ctx->stream_bytes, stream_chain_workspace(first));
        if (rc != RT_DP_OK) return rc;
        const uint32_t *in_words = reinterpret_cast<const uint32_t *>(d_in);
        uint32_t *out_words = reinterpret_cast<uint32_t *>(d_out);
        // all three registers travel in one copy here: bring the Euclidean one over to the angular ones' copy first
        const int home_e = ctx->cur_e, home_a = ctx->cur_a;
        if (ctx->extended && home_e != home_a)
            HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * home_a, ctx->d_state + 4 * home_e, sizeof(float), hipMemcpyDeviceToDevice, s));
        int c = home_a;
        HIP_TRY(hipMemcpyAsync(ctx->d_state + 8, ctx->d_state + 4 * c, 4 * sizeof(float), hipMemcpyDeviceToDevice, s));  // backup
        HIP_TRY(stream_chain_begin(ctx->d_stream, s));
        const unsigned long long *sticky = nullptr;
        for (uint64_t off = 0; off < n; off += chunk) {
            const uint64_t m = (n - off < chunk) ? (n - off) : chunk;
            HIP_TRY(stream_chain_eval(in_words + off * 84, m, first, ctx->extended ? 1 : 0, out_words + off * 21, ctx->d_stream, ctx->d_state + 4 * c,
                                      ctx->d_state + 4 * (c ^ 1), &sticky, s, ctx->debug_chain_skip));
            if (ctx->extended) c ^= 1;
        }
        rc = pinned_words(ctx);
        if (rc != RT_DP_OK) return rc;
        unsigned long long *seen = ctx->h_words;  // {give-ups, opcode > 3}
        if (pinned_out) {
            HIP_TRY(hipMemcpyAsync(pinned_out, d_out, n * sizeof(rt_dp_output_bundle), hipMemcpyDeviceToHost, s));
            *copied = true;
        }
        HIP_TRY(hipMemcpyAsync(seen, sticky, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const bool bad = seen[1] != 0;
        const bool gave_up = seen[0] != 0;
        ctx->stream_giveups += seen[0];
        if (bad || gave_up) {  // the registers get their values from before the call back
            c = home_a;
            HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * c, ctx->d_state + 8, 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        if (ctx->extended) {
            // all three registers are in copy c now; a register whose address a graph holds goes back to that address
            // (copies on s, which is idle: whatever the context launches next is ordered behind them)
            if (ctx->pinned_e && c != home_e) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * home_e, ctx->d_state + 4 * c, sizeof(float), hipMemcpyDeviceToDevice, s));
            if (!ctx->pinned_e) ctx->cur_e = c;
            if (ctx->pinned_a && c != home_a) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * home_a + 1, ctx->d_state + 4 * c + 1, 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (!ctx->pinned_a) ctx->cur_a = c;
        }
        if (bad) return fail(RT_DP_EOPCODE, "the record stream holds an opcode > 3; the RTL's opcode field is 2 bits wide");
        if (gave_up) {
            // a wave waited in vain for the running sums of the waves before it: the two-pass form has no such wait and
            // delivers the same records
            ++ctx->chain_fallbacks;
            ctx->stream_mode = 2;
            const int rc2 = bundles_device(ctx, d_in, d_out, n, true, pinned_out, copied);
            ctx->stream_mode = 3;
            return rc2;
        }
        return RT_DP_OK;
    }
    if (ctx->stream_mode == 2 || (ctx->stream_mode == 0 && aligned16)) {
        // one pass over the records: evaluate in place, then add up the distance beats in stream order
        // in pieces of 2^26 records, which bounds the workspace (12.6 bytes per record of a piece); smaller pieces
        // measured slower (2^20: -17 %), the per-piece launches cost more than cache residency returns
        const int lg = env_int("RT_DP_STREAM_CHUNK_LOG2", 26);
        const uint64_t chunk = 1ull << (lg < 11 ? 11 : (lg > 30 ? 30 : lg));
        const uint64_t first = n < chunk ? n : chunk;
        int rc = grow(ctx->d_stream, ctx->stream_bytes, stream_fused_workspace(first));
        if (rc != RT_DP_OK) return rc;
        StreamSideView view;
        const uint32_t *in_words = reinterpret_cast<const uint32_t *>(d_in);
        uint32_t *out_words = reinterpret_cast<uint32_t *>(d_out);
        // the fold kernel takes all three registers from one copy: bring the Euclidean one over to the angular ones' copy first
        const int home_e = ctx->cur_e, home_a = ctx->cur_a;
        if (ctx->extended && home_e != home_a)
            HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * home_a, ctx->d_state + 4 * home_e, sizeof(float), hipMemcpyDeviceToDevice, s));
        int c = home_a;
        // A fold reads the registers from one copy and writes all three into the other, so a call of one piece leaves the copy
        // it started from intact; only a call of several pieces needs the backup for the case that it is refused.
        const bool one_piece = n <= chunk;
        if (ctx->extended && !one_piece)
            HIP_TRY(hipMemcpyAsync(ctx->d_state + 8, ctx->d_state + 4 * c, 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
        for (uint64_t off = 0; off < n; off += chunk) {
            const uint64_t m = (n - off < chunk) ? (n - off) : chunk;
            HIP_TRY(stream_fused_eval(in_words + off * 84, m, first, ctx->extended ? 1 : 0, out_words + off * 21, ctx->d_stream, &view, s, off == 0));
            if (ctx->extended) {
                HIP_TRY(stream_fused_fold(view, m, ctx->d_state + 4 * c, ctx->d_state + 4 * (c ^ 1), out_words + off * 21, s));
                c ^= 1;
            }
        }
        rc = pinned_words(ctx);
        if (rc != RT_DP_OK) return rc;
        if (pinned_out) {
            HIP_TRY(hipMemcpyAsync(pinned_out, d_out, n * sizeof(rt_dp_output_bundle), hipMemcpyDeviceToHost, s));
            *copied = true;
        }
        HIP_TRY(hipMemcpyAsync(ctx->h_words, view.flags + 8, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const uint32_t bad = *reinterpret_cast<const uint32_t *>(ctx->h_words);
        if (bad) {  // the registers keep their values from before the call
            c = home_a;
            if (ctx->extended && !one_piece) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * c, ctx->d_state + 8, 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        if (ctx->extended) {
            // all three registers are in copy c now; a register whose address a graph holds goes back to that address
            // (copies on s, which is idle: whatever the context launches next is ordered behind them)
            if (ctx->pinned_e && c != home_e) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * home_e, ctx->d_state + 4 * c, sizeof(float), hipMemcpyDeviceToDevice, s));
            if (!ctx->pinned_e) ctx->cur_e = c;
            if (ctx->pinned_a && c != home_a) HIP_TRY(hipMemcpyAsync(ctx->d_state + 4 * home_a + 1, ctx->d_state + 4 * c + 1, 2 * sizeof(float), hipMemcpyDeviceToDevice, s));
            if (!ctx->pinned_a) ctx->cur_a = c;
        }
        if (bad) return fail(RT_DP_EOPCODE, "the record stream holds an opcode > 3; the RTL's opcode field is 2 bits wide");
        return RT_DP_OK;
    }
    const uint64_t nblk = (n + 255) / 256;
    const size_t off_slot = 0, off_cnt = align_up(n * 4), off_tot = off_cnt + align_up(nblk * 16);
    int rc = grow(ctx->d_stream, ctx->stream_bytes, off_tot + 256);
    if (rc != RT_DP_OK) return rc;
    uint32_t *slot = reinterpret_cast<uint32_t *>(ctx->d_stream + off_slot);
    uint32_t *blockcnt = reinterpret_cast<uint32_t *>(ctx->d_stream + off_cnt);
    uint32_t *totals = reinterpret_cast<uint32_t *>(ctx->d_stream + off_tot);
    const uint32_t *in_words = reinterpret_cast<const uint32_t *>(d_in);

    HIP_TRY(stream_count_scan(in_words, n, blockcnt, totals, s));
    uint32_t h[8];
    rc = read_back(ctx, h, totals, sizeof h, s);
    if (rc != RT_DP_OK) return rc;
    if (h[4]) return fail(RT_DP_EOPCODE, "the record stream holds an opcode > 3; the RTL's opcode field is 2 bits wide");
    const uint64_t nt = h[0], nb = h[1], ne = h[2], na = h[3];

    // carve the structure-of-arrays segments (operands and results) out of the staging buffer
    size_t total = 0;
    auto take = [&](size_t bytes) { size_t o = total; total += align_up(bytes ? bytes : 4); return o; };
    // box and triangle segments in tiled form: one buffer each way, which is what their kernels run fastest on
    const size_t o_boxin = take((size_t)RT_DP_TILED_WORDS(nb, RT_DP_BOX_TILE_IN_WORDS) * 4), o_boxout = take((size_t)RT_DP_TILED_WORDS(nb, RT_DP_BOX_TILE_OUT_WORDS) * 4);
    const size_t o_triin = take((size_t)RT_DP_TILED_WORDS(nt, RT_DP_TRI_TILE_IN_WORDS) * 4), o_triout = take((size_t)RT_DP_TILED_WORDS(nt, RT_DP_TRI_TILE_OUT_WORDS) * 4);
    const size_t o_ea = take(ne * 64), o_eb = take(ne * 64), o_ec = take(ne * 4), o_eo = take(ne * 4);
    const size_t o_aa = take(na * 32), o_ab = take(na * 32), o_ac = take(na * 4), o_ad = take(na * 4), o_an = take(na * 4);
    rc = grow(ctx->d_stage, ctx->stage_bytes, total);
    if (rc != RT_DP_OK) return rc;
    char *base = ctx->d_stage;
    auto F = [&](size_t o) { return reinterpret_cast<float *>(base + o); };
    auto U = [&](size_t o) { return reinterpret_cast<uint32_t *>(base + o); };

    rt_dp_op_batch ob;
    rt_dp_result_batch rb;
    memset(&ob, 0, sizeof ob);
    memset(&rb, 0, sizeof rb);
    ob.struct_size = sizeof ob; rb.struct_size = sizeof rb;
    ob.flags = RT_DP_F_DEVICE_PTRS | RT_DP_F_BOX_TILED | RT_DP_F_TRI_TILED;
    ob.n[0] = nt; ob.n[1] = nb; ob.n[2] = ne; ob.n[3] = na;
    ob.box.org[0] = F(o_boxin); rb.box.tmin_out[0] = F(o_boxout);
    ob.tri.org[0] = F(o_triin); rb.tri.t_num = F(o_triout);
    ob.euc.a = F(o_ea); ob.euc.b = F(o_eb); ob.euc.ctl = U(o_ec); rb.euc.accumulator = F(o_eo);
    ob.ang.a = F(o_aa); ob.ang.b = F(o_ab); ob.ang.ctl = U(o_ac); rb.ang.dot_product = F(o_ad); rb.ang.norm = F(o_an);

    // what the partition and re-assembly kernels address: the run of every word in tile 0 (they add the tile's offset)
    BoxArgs ba;
    TriArgs ta;
    for (int f = 0; f < kBoxFields; ++f) ba.in[f] = F(o_boxin) + (size_t)f * kTileOps;
    for (int k = 0; k < 4; ++k) ba.tmin[k] = F(o_boxout) + (size_t)k * kTileOps;
    ba.flags = U(o_boxout) + (size_t)4 * kTileOps;
    for (int f = 0; f < kTriFields; ++f) ta.in[f] = F(o_triin) + (size_t)f * kTileOps;
    ta.k = U(o_triin) + (size_t)kTriFields * kTileOps;
    ta.t_num = F(o_triout); ta.t_denom = F(o_triout) + (size_t)kTileOps; ta.hit = U(o_triout) + (size_t)2 * kTileOps;
    VecArgs ea{ob.euc.a, ob.euc.b, ob.euc.ctl, rb.euc.accumulator, nullptr, nullptr, nullptr};
    VecArgs aa{ob.ang.a, ob.ang.b, ob.ang.ctl, rb.ang.dot_product, rb.ang.norm, nullptr, nullptr};
    HIP_TRY(stream_scatter(in_words, n, blockcnt, slot, ba, ta, ea, aa, s));
    RegSnapshot snap;
    rc = regs_snapshot(ctx, s, snap);
    if (rc == RT_DP_OK) rc = launch_device(ctx, &ob, &rb, s);
    if (rc == RT_DP_OK) rc = chain_probe(ctx, s);
    if (rc != RT_DP_OK) return rc;
    for (int round = 0; round < 2; ++round) {  // a second round only behind a give-up of the chained beat kernels
        HIP_TRY(stream_gather(slot, n, ba, ta, ea, aa, ctx->extended ? 1 : 0, reinterpret_cast<uint32_t *>(d_out), s));
        HIP_TRY(hipStreamSynchronize(s));
        if (round == 1 || !chain_moved(ctx)) break;
        rc = beats_again(ctx, &ob, &rb, s, snap);
        if (rc != RT_DP_OK) return rc;
    }
    return RT_DP_OK;
}

int rt_dp_ray_prepare(rt_dp_ctx *ctx, const rt_dp_ray_prep *rays, uint64_t n, void *hip_stream) {
    if (!ctx || !rays) return fail(RT_DP_ENULL, "rt_dp_ray_prepare: NULL argument");
    if (n == 0) return RT_DP_OK;
    if (n > kMaxOps) return fail(RT_DP_EINVAL, "n = %llu is not representable (limit 2^40)", (unsigned long long)n);
    RayPrepArgs a;
    for (int c = 0; c < 3; ++c) {
        if (!rays->dir[c]) return fail(RT_DP_ENULL, "rt_dp_ray_prepare: dir[] is NULL");
        a.dir[c] = rays->dir[c];
        a.inv[c] = rays->inv[c];
        a.shear[c] = rays->shear[c];
    }
    if ((!a.inv[0]) != (!a.inv[1]) || (!a.inv[0]) != (!a.inv[2])) return fail(RT_DP_ENULL, "rt_dp_ray_prepare: give all three inv arrays or none");
    if ((!a.shear[0]) != (!a.shear[1]) || (!a.shear[0]) != (!a.shear[2])) return fail(RT_DP_ENULL, "rt_dp_ray_prepare: give all three shear arrays or none");
    a.k = rays->k;
    a.extent = rays->extent;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(launch_ray_prepare(a, n, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

namespace {

int knn_exhaustive(rt_dp_ctx *ctx, const char *who, bool angular, const float *points, uint32_t n_points, const float *queries,
                   uint32_t n_queries, uint32_t index_base, uint32_t splits, float *out_val, uint32_t *out_idx, void *hip_stream) {
    if (!ctx) return fail(RT_DP_ENULL, "%s: ctx is NULL", who);
    if (n_queries == 0) return RT_DP_OK;
    if (!queries || !out_val || !out_idx || (n_points && !points)) return fail(RT_DP_ENULL, "%s: NULL array", who);
    if (splits == 0 || splits > 65535u) return fail(RT_DP_EINVAL, "splits must be 1..65535");
    if ((uint64_t)index_base + n_points > 0xFFFFFFFFull) return fail(RT_DP_EINVAL, "index_base + n_points exceeds the 32-bit index range");
    if ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(queries)) & 7u)
        return fail(RT_DP_EALIGN, "%s: points and queries must be 8-byte aligned", who);
    if ((reinterpret_cast<uintptr_t>(out_val) | reinterpret_cast<uintptr_t>(out_idx)) & 3u)
        return fail(RT_DP_EALIGN, "%s: output arrays must be 4-byte aligned", who);
    if (angular && !ctx->extended) return fail(RT_DP_EINVAL, "%s: the baseline datapath has no angular op", who);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(launch_knn_euclid16(points, n_points, queries, n_queries, index_base, splits, angular, out_val, out_idx,
                                hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

}  // namespace

int rt_dp_knn_euclid16(rt_dp_ctx *ctx, const float *points, uint32_t n_points, const float *queries, uint32_t n_queries,
                       uint32_t index_base, uint32_t splits, float *out_dist, uint32_t *out_idx, void *hip_stream) {
    return knn_exhaustive(ctx, "rt_dp_knn_euclid16", false, points, n_points, queries, n_queries, index_base, splits, out_dist, out_idx, hip_stream);
}

int rt_dp_knn_angular16(rt_dp_ctx *ctx, const float *points, uint32_t n_points, const float *queries, uint32_t n_queries,
                        uint32_t index_base, uint32_t splits, float *out_neg_score, uint32_t *out_idx, void *hip_stream) {
    return knn_exhaustive(ctx, "rt_dp_knn_angular16", true, points, n_points, queries, n_queries, index_base, splits, out_neg_score, out_idx,
                          hip_stream);
}

int rt_dp_knn_tree_euclid16(rt_dp_ctx *ctx, const rt_dp_knn_tree *tree, const float *queries, const uint32_t *home_leaf,
                            uint32_t n_queries, uint32_t splits, uint32_t *bound, float *out_dist, uint32_t *out_idx, void *hip_stream) {
    if (!ctx || !tree) return fail(RT_DP_ENULL, "rt_dp_knn_tree_euclid16: NULL argument");
    if (tree->struct_size != sizeof(rt_dp_knn_tree)) return fail(RT_DP_EINVAL, "rt_dp_knn_tree.struct_size does not match this library");
    if (n_queries == 0) return RT_DP_OK;
    if (!queries || !home_leaf || !bound || !out_dist || !out_idx || !tree->points || !tree->index || !tree->leaf_lo || !tree->leaf_hi)
        return fail(RT_DP_ENULL, "rt_dp_knn_tree_euclid16: NULL array");
    const uint32_t nl = tree->n_leaves, ls = tree->leaf_size;
    if (nl == 0 || (nl & (nl - 1))) return fail(RT_DP_EINVAL, "n_leaves = %u is not a power of two", nl);
    if (ls == 0 || (ls & 3u)) return fail(RT_DP_EINVAL, "leaf_size = %u is not a positive multiple of 4", ls);
    if ((uint64_t)nl * ls > 0xFFFFFFFFull) return fail(RT_DP_EINVAL, "n_leaves * leaf_size exceeds the 32-bit slot range");
    if (splits == 0 || (splits & (splits - 1)) || splits > nl || splits > 65535u)
        return fail(RT_DP_EINVAL, "splits = %u must be a power of two not above n_leaves", splits);
    if ((reinterpret_cast<uintptr_t>(tree->points) | reinterpret_cast<uintptr_t>(tree->leaf_lo) | reinterpret_cast<uintptr_t>(tree->leaf_hi) |
         reinterpret_cast<uintptr_t>(queries)) & 7u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_euclid16: coordinate arrays must be 8-byte aligned");
    if ((reinterpret_cast<uintptr_t>(tree->index) | reinterpret_cast<uintptr_t>(home_leaf) | reinterpret_cast<uintptr_t>(bound) |
         reinterpret_cast<uintptr_t>(out_dist) | reinterpret_cast<uintptr_t>(out_idx)) & 3u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_euclid16: arrays must be 4-byte aligned");
    uint32_t log2_splits = 0;
    while ((1u << log2_splits) < splits) ++log2_splits;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(launch_knn_tree16(tree->points, tree->index, tree->leaf_lo, tree->leaf_hi, nl, ls, queries, home_leaf, n_queries, log2_splits, 0, bound,
                              out_dist, out_idx, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

int rt_dp_knn_tree_angular16(rt_dp_ctx *ctx, const rt_dp_knn_tree *tree, const float *queries, const uint32_t *home_leaf, uint32_t n_queries,
                             uint32_t splits, uint32_t *bound, float *out_neg_score, uint32_t *out_idx, uint64_t *d_leaves_scanned, void *hip_stream) {
    if (!ctx || !tree) return fail(RT_DP_ENULL, "rt_dp_knn_tree_angular16: NULL argument");
    if (tree->struct_size != sizeof(rt_dp_knn_tree)) return fail(RT_DP_EINVAL, "rt_dp_knn_tree.struct_size does not match this library");
    if (!ctx->extended) return fail(RT_DP_EINVAL, "rt_dp_knn_tree_angular16: the baseline datapath has no angular op");
    if (n_queries == 0) return RT_DP_OK;
    if (!queries || !home_leaf || !bound || !out_neg_score || !out_idx || !tree->points || !tree->index || !tree->leaf_lo || !tree->leaf_hi)
        return fail(RT_DP_ENULL, "rt_dp_knn_tree_angular16: NULL array");
    const uint32_t nl = tree->n_leaves, ls = tree->leaf_size;
    if (nl == 0 || (nl & (nl - 1))) return fail(RT_DP_EINVAL, "n_leaves = %u is not a power of two", nl);
    if (ls == 0 || (ls & 3u)) return fail(RT_DP_EINVAL, "leaf_size = %u is not a positive multiple of 4", ls);
    if ((uint64_t)nl * ls > 0xFFFFFFFFull) return fail(RT_DP_EINVAL, "n_leaves * leaf_size exceeds the 32-bit slot range");
    if (splits == 0 || (splits & (splits - 1)) || splits > nl || splits > 65535u)
        return fail(RT_DP_EINVAL, "splits = %u must be a power of two not above n_leaves", splits);
    if ((reinterpret_cast<uintptr_t>(tree->points) | reinterpret_cast<uintptr_t>(queries)) & 7u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_angular16: coordinate arrays must be 8-byte aligned");
    if ((reinterpret_cast<uintptr_t>(tree->index) | reinterpret_cast<uintptr_t>(tree->leaf_lo) | reinterpret_cast<uintptr_t>(tree->leaf_hi) |
         reinterpret_cast<uintptr_t>(home_leaf) | reinterpret_cast<uintptr_t>(bound) | reinterpret_cast<uintptr_t>(out_neg_score) |
         reinterpret_cast<uintptr_t>(out_idx)) & 3u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_angular16: arrays must be 4-byte aligned");
    if (reinterpret_cast<uintptr_t>(d_leaves_scanned) & 7u) return fail(RT_DP_EALIGN, "rt_dp_knn_tree_angular16: d_leaves_scanned must be 8-byte aligned");
    uint32_t log2_splits = 0;
    while ((1u << log2_splits) < splits) ++log2_splits;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(launch_knn_tree_ang16(tree->points, tree->index, tree->leaf_lo, tree->leaf_hi, nl, ls, queries, home_leaf, n_queries, log2_splits, bound,
                                  out_neg_score, out_idx, reinterpret_cast<unsigned long long *>(d_leaves_scanned),
                                  hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

namespace {

// Replaces a device buffer whose contents are dead by one of at least `want` bytes.
int regrow(char *&buf, size_t &have, size_t want, const char *what) {
    if (want <= have) return RT_DP_OK;
    if (buf) { hipError_t e = hipFree(buf); buf = nullptr; have = 0; if (e != hipSuccess) return fail(RT_DP_EHIP, "hipFree: %s", hipGetErrorString(e)); }
    hipError_t e = hipMalloc(&buf, want);
    if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipMalloc(%zu) for %s failed: %s", want, what, hipGetErrorString(e));
    have = want;
    return RT_DP_OK;
}

}  // namespace

int rt_dp_knn_tree_grouped_euclid16(rt_dp_ctx *ctx, const rt_dp_knn_tree *tree, const float *queries, const uint32_t *home_leaf,
                                    uint32_t n_queries, uint32_t walk_leaves, float *out_dist, uint32_t *out_idx, uint64_t stats[4]) {
    if (!ctx || !tree) return fail(RT_DP_ENULL, "rt_dp_knn_tree_grouped_euclid16: NULL argument");
    if (tree->struct_size != sizeof(rt_dp_knn_tree)) return fail(RT_DP_EINVAL, "rt_dp_knn_tree.struct_size does not match this library");
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (n_queries == 0) return RT_DP_OK;
    if (!queries || !home_leaf || !out_dist || !out_idx || !tree->points || !tree->index || !tree->leaf_lo || !tree->leaf_hi)
        return fail(RT_DP_ENULL, "rt_dp_knn_tree_grouped_euclid16: NULL array");
    const uint32_t nl = tree->n_leaves, ls = tree->leaf_size;
    if (nl == 0 || (nl & (nl - 1))) return fail(RT_DP_EINVAL, "n_leaves = %u is not a power of two", nl);
    if (ls == 0 || (ls & 3u)) return fail(RT_DP_EINVAL, "leaf_size = %u is not a positive multiple of 4", ls);
    if ((uint64_t)nl * ls > 0xFFFFFFFFull) return fail(RT_DP_EINVAL, "n_leaves * leaf_size exceeds the 32-bit slot range");
    if ((reinterpret_cast<uintptr_t>(tree->points) | reinterpret_cast<uintptr_t>(tree->leaf_lo) | reinterpret_cast<uintptr_t>(tree->leaf_hi) |
         reinterpret_cast<uintptr_t>(queries)) & 7u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_grouped_euclid16: coordinate arrays must be 8-byte aligned");
    if ((reinterpret_cast<uintptr_t>(tree->index) | reinterpret_cast<uintptr_t>(home_leaf) | reinterpret_cast<uintptr_t>(out_dist) |
         reinterpret_cast<uintptr_t>(out_idx)) & 3u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_grouped_euclid16: arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    hipStream_t s = ctx->stream;
    HIP_TRY(hipDeviceSynchronize());  // the caller's arrays may still be in flight on other streams
    const KnnTreeView view{tree->points, tree->index, tree->leaf_lo, tree->leaf_hi, nl, ls};
    const uint64_t nq = n_queries;

    // tables: tau[nq] | n_cand[nq] | overflow[nq] | count[nl] | offset[nl] | group_first[nl]
    const size_t o_tau = 0, o_nc = align_up(nq * 4), o_ov = o_nc + align_up(nq * 4), o_cnt = o_ov + align_up(nq * 4);
    const size_t o_off = o_cnt + align_up((size_t)nl * 4), o_gf = o_off + align_up((size_t)nl * 4), tables = o_gf + align_up((size_t)nl * 4);
    int rc = regrow(ctx->d_knn[0], ctx->knn_bytes[0], tables, "k-NN tables");
    if (rc != RT_DP_OK) return rc;
    char *tb = ctx->d_knn[0];
    uint32_t *tau = reinterpret_cast<uint32_t *>(tb + o_tau), *n_cand = reinterpret_cast<uint32_t *>(tb + o_nc);
    uint32_t *overflow = reinterpret_cast<uint32_t *>(tb + o_ov), *count = reinterpret_cast<uint32_t *>(tb + o_cnt);
    uint32_t *offset = reinterpret_cast<uint32_t *>(tb + o_off), *group_first = reinterpret_cast<uint32_t *>(tb + o_gf);

    // 1. upper bounds: the wave walk over the nearest leaves only, in up to 8 splits so that a small query set still
    //    fills the chip; the splits' lists (parked in the candidate buffer) are merged into one bound per query
    {
        uint32_t walk = walk_leaves ? walk_leaves : 16u;
        if (walk > nl) walk = nl;
        uint32_t lg = 0;
        while (lg < 3 && (2u << lg) <= nl && (walk >> (lg + 1)) >= 4u && ((nq + 255) / 256 << lg) < 4096) ++lg;
        const uint32_t sp = 1u << lg;
        rc = regrow(ctx->d_knn[2], ctx->knn_bytes[2], (size_t)sp * nq * 16 * 8, "k-NN bound lists");
        if (rc != RT_DP_OK) return rc;
        float *bd = reinterpret_cast<float *>(ctx->d_knn[2]);
        uint32_t *bi = reinterpret_cast<uint32_t *>(ctx->d_knn[2] + (size_t)sp * nq * 16 * 4);
        HIP_TRY(launch_knn_tree16(view.pts, view.index, view.leaf_lo, view.leaf_hi, nl, ls, queries, home_leaf, n_queries, lg,
                                  (walk + sp - 1) / sp, tau, bd, bi, s));
        HIP_TRY(launch_knn_tau_merge(bd, sp, n_queries, tau, s));
        HIP_TRY(hipStreamSynchronize(s));  // the candidate buffer is re-used (and possibly re-allocated) below
    }
    // Rounds of: per-leaf query lists (count, prefix sums on the host, fill), scan, select.  Queries that fill
    // their candidate slots go into another round with a tighter bound; everything else is settled.
    uint64_t cap = 1024, pairs_total = 0, groups_total = 0, rounds = 0;
    const uint64_t slots = (uint64_t)nl * ls;
    std::vector<uint32_t> h_count(nl), h_off(nl), h_gf(nl), continuing;
    for (;;) {
        ++rounds;
        if (cap > slots) cap = slots;
        if (nq * cap * 8 > (64ull << 30)) return fail(RT_DP_ENOMEM, "candidate slots would need %llu GiB: use rt_dp_knn_tree_euclid16 for this data",
                                                      (unsigned long long)((nq * cap * 8) >> 30));
        // the first pass over the lists covers every query and remembers its box-test ballots (8 bytes per wave and leaf) for the fill pass;
        // later rounds cover the few queries that continue
        const uint32_t *subset = nullptr;
        uint32_t n_sub = n_queries;
        unsigned long long *masks = nullptr;
        if (rounds == 1) {
            const size_t mask_bytes = (size_t)((nq + 63) / 64) * nl * 8;
            if (mask_bytes <= (8ull << 30)) {
                rc = regrow(ctx->d_knn[3], ctx->knn_bytes[3], mask_bytes, "k-NN ballots");
                if (rc != RT_DP_OK) return rc;
                masks = reinterpret_cast<unsigned long long *>(ctx->d_knn[3]);
            }
        } else {
            rc = regrow(ctx->d_knn[3], ctx->knn_bytes[3], continuing.size() * 4, "k-NN subset");
            if (rc != RT_DP_OK) return rc;
            HIP_TRY(hipMemcpyAsync(ctx->d_knn[3], continuing.data(), continuing.size() * 4, hipMemcpyHostToDevice, s));
            subset = reinterpret_cast<const uint32_t *>(ctx->d_knn[3]);
            n_sub = (uint32_t)continuing.size();
        }
        HIP_TRY(hipMemsetAsync(count, 0, (size_t)nl * 4, s));
        HIP_TRY(launch_knn_pairs(view, queries, subset, n_sub, tau, count, nullptr, nullptr, masks, s));
        HIP_TRY(hipMemcpyAsync(h_count.data(), count, (size_t)nl * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        uint64_t pairs = 0, groups = 0;
        for (uint32_t l = 0; l < nl; ++l) {
            h_off[l] = (uint32_t)pairs;
            h_gf[l] = (uint32_t)groups;
            pairs += h_count[l];
            groups += (h_count[l] + 63u) / 64u;
            if (pairs > 0xFFFFFFFFull) return fail(RT_DP_ENOMEM, "more than 2^32 (leaf, query) pairs: use rt_dp_knn_tree_euclid16 for this data");
        }
        pairs_total += pairs;
        groups_total += groups;
        rc = regrow(ctx->d_knn[1], ctx->knn_bytes[1], align_up(pairs * 4) + align_up(groups * 4) + 256, "k-NN lists");
        if (rc != RT_DP_OK) return rc;
        rc = regrow(ctx->d_knn[2], ctx->knn_bytes[2], nq * cap * 8, "k-NN candidates");
        if (rc != RT_DP_OK) return rc;
        uint32_t *list = reinterpret_cast<uint32_t *>(ctx->d_knn[1]);
        uint32_t *group_leaf = reinterpret_cast<uint32_t *>(ctx->d_knn[1] + align_up(pairs * 4));
        unsigned long long *cand = reinterpret_cast<unsigned long long *>(ctx->d_knn[2]);
        HIP_TRY(hipMemcpyAsync(offset, h_off.data(), (size_t)nl * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(group_first, h_gf.data(), (size_t)nl * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemsetAsync(count, 0, (size_t)nl * 4, s));
        HIP_TRY(launch_knn_pairs(view, queries, subset, n_sub, tau, count, offset, list, masks, s));
        HIP_TRY(launch_knn_groups(group_first, count, nl, group_leaf, s));
        HIP_TRY(hipMemsetAsync(n_cand, 0, 2 * align_up(nq * 4), s));  // n_cand and overflow
        HIP_TRY(launch_knn_leaf_scan(view, queries, tau, (uint32_t)groups, group_leaf, group_first, count, offset, list, (uint32_t)cap, n_cand, cand, s));
        HIP_TRY(launch_knn_select(tau, n_cand, cand, (uint32_t)cap, n_queries, out_dist, out_idx, overflow, s));
        HIP_TRY(hipMemsetAsync(group_first, 0, 8, s));  // re-used as the round's status words
        HIP_TRY(launch_knn_retau(tau, overflow, out_dist, n_queries, group_first, s));
        uint32_t status[2];
        HIP_TRY(hipMemcpyAsync(status, group_first, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (status[0] == 0) break;
        {  // the queries that continue: those whose bound is still a distance (not -Inf)
            std::vector<uint32_t> h_tau(nq);
            HIP_TRY(hipMemcpy(h_tau.data(), tau, nq * 4, hipMemcpyDeviceToHost));
            continuing.clear();
            for (uint32_t i = 0; i < n_queries; ++i)
                if (h_tau[i] != 0xFF800000u) continuing.push_back(i);
        }
        if (status[1]) {  // more equal-distance candidates than slots: only more slots help
            if (cap == slots) return fail(RT_DP_EHIP, "internal error: candidate slots cover every point and still overflow");
            cap *= 4;
        }
        if (rounds > 64) return fail(RT_DP_EHIP, "internal error: the leaf-major search does not converge");
    }
    const uint64_t pairs = pairs_total, groups = groups_total;
    if (stats) { stats[0] = pairs; stats[1] = groups; stats[2] = cap; stats[3] = rounds; }
    return RT_DP_OK;
}

uint64_t rt_dp_beat_offsets_host(const uint64_t *pair_offsets, uint64_t n_pairs, uint32_t width, uint64_t *beat_offsets) {
    if (!pair_offsets || (width != 16 && width != 8)) return 0;
    uint64_t beats = 0;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        if (beat_offsets) beat_offsets[p] = beats;
        const uint64_t len = pair_offsets[p + 1] >= pair_offsets[p] ? pair_offsets[p + 1] - pair_offsets[p] : 0;
        beats += (len + width - 1) / width;
    }
    if (beat_offsets) beat_offsets[n_pairs] = beats;
    return beats;
}

int rt_dp_pairs_to_beats(rt_dp_ctx *ctx, const float *a, const float *b, const uint64_t *pair_offsets, const uint64_t *beat_offsets,
                         uint64_t n_pairs, uint64_t n_beats, uint32_t width, float *out_a, float *out_b, uint32_t *out_ctl, void *hip_stream) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_pairs_to_beats: ctx is NULL");
    if (width != 16 && width != 8) return fail(RT_DP_EINVAL, "width must be 16 (Euclidean) or 8 (angular)");
    if (n_beats == 0) return RT_DP_OK;
    if (n_beats > kMaxOps || n_pairs > kMaxOps) return fail(RT_DP_EINVAL, "counts are not representable (limit 2^40)");
    if (!a || !b || !pair_offsets || !beat_offsets || !out_a || !out_b || !out_ctl || n_pairs == 0) return fail(RT_DP_ENULL, "rt_dp_pairs_to_beats: NULL array");
    if ((reinterpret_cast<uintptr_t>(pair_offsets) | reinterpret_cast<uintptr_t>(beat_offsets)) & 7u) return fail(RT_DP_EALIGN, "offset arrays must be 8-byte aligned");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out_a) | reinterpret_cast<uintptr_t>(out_b) |
         reinterpret_cast<uintptr_t>(out_ctl)) & 3u)
        return fail(RT_DP_EALIGN, "arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(pairs_to_beats(a, b, pair_offsets, beat_offsets, n_pairs, n_beats, width, out_a, out_b, out_ctl,
                           hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

int rt_dp_beats_to_pairs(rt_dp_ctx *ctx, const float *values, const uint64_t *beat_offsets, uint64_t n_pairs, float *out_pair_values,
                         void *hip_stream) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_beats_to_pairs: ctx is NULL");
    if (n_pairs == 0) return RT_DP_OK;
    if (n_pairs > kMaxOps) return fail(RT_DP_EINVAL, "n_pairs is not representable (limit 2^40)");
    if (!values || !beat_offsets || !out_pair_values) return fail(RT_DP_ENULL, "rt_dp_beats_to_pairs: NULL array");
    if (reinterpret_cast<uintptr_t>(beat_offsets) & 7u) return fail(RT_DP_EALIGN, "beat_offsets must be 8-byte aligned");
    if ((reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(out_pair_values)) & 3u) return fail(RT_DP_EALIGN, "arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(beats_to_pairs(values, beat_offsets, n_pairs, out_pair_values, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

static int box_compact(const char *who, rt_dp_ctx *ctx, const uint32_t *flags, const float *const tm[4], uint32_t tile_words, uint64_t n,
                       const uint32_t *item, const uint32_t *children, uint32_t *out_item, uint32_t *out_child, float *out_tmin,
                       uint64_t capacity, uint64_t *count) {
    if (n > kMaxOps) return fail(RT_DP_EINVAL, "n = %llu is not representable (limit 2^40)", (unsigned long long)n);
    if (!flags || (capacity && (!out_item || !out_child))) return fail(RT_DP_ENULL, "%s: NULL array", who);
    uintptr_t acc = reinterpret_cast<uintptr_t>(flags) | reinterpret_cast<uintptr_t>(item) | reinterpret_cast<uintptr_t>(children) |
                    reinterpret_cast<uintptr_t>(out_item) | reinterpret_cast<uintptr_t>(out_child) | reinterpret_cast<uintptr_t>(out_tmin);
    if (out_tmin)
        for (int k = 0; k < 4; ++k) {
            if (!tm[k]) return fail(RT_DP_ENULL, "%s: out_tmin needs all four tmin_out arrays", who);
            acc |= reinterpret_cast<uintptr_t>(tm[k]);
        }
    if (acc & 3u) return fail(RT_DP_EALIGN, "%s: arrays must be 4-byte aligned", who);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    hipStream_t s = ctx->stream;
    HIP_TRY(hipDeviceSynchronize());  // the results may still be in production on a caller stream
    const uint64_t nblk = (n + 255) / 256;
    const size_t o_cnt = 0, o_off = align_up(nblk * 4), o_tot = o_off + align_up(nblk * 8);
    int rc = regrow(ctx->d_trace, ctx->trace_bytes, o_tot + 256, "compaction offsets");
    if (rc != RT_DP_OK) return rc;
    uint32_t *blockcnt = reinterpret_cast<uint32_t *>(ctx->d_trace + o_cnt);
    unsigned long long *offsets64 = reinterpret_cast<unsigned long long *>(ctx->d_trace + o_off);
    unsigned long long *total = reinterpret_cast<unsigned long long *>(ctx->d_trace + o_tot);
    HIP_TRY(compact_hits(flags, tm, n, item, children, blockcnt, offsets64, total, capacity, out_item, out_child, out_tmin, s, tile_words));
    rc = pinned_words(ctx);
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_words, total, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *count = ctx->h_words[0];
    return RT_DP_OK;
}

int rt_dp_box_compact(rt_dp_ctx *ctx, const rt_dp_box_out *results, uint64_t n, const uint32_t *item, const uint32_t *children,
                      uint32_t *out_item, uint32_t *out_child, float *out_tmin, uint64_t capacity, uint64_t *count) {
    if (!ctx || !results || !count) return fail(RT_DP_ENULL, "rt_dp_box_compact: NULL argument");
    *count = 0;
    if (n == 0) return RT_DP_OK;
    const float *tm[4] = {results->tmin_out[0], results->tmin_out[1], results->tmin_out[2], results->tmin_out[3]};
    return box_compact("rt_dp_box_compact", ctx, results->flags, tm, 0u, n, item, children, out_item, out_child, out_tmin, capacity, count);
}

int rt_dp_box_compact_tiled(rt_dp_ctx *ctx, const void *tiled_results, uint64_t n, const uint32_t *item, const uint32_t *children,
                            uint32_t *out_item, uint32_t *out_child, float *out_tmin, uint64_t capacity, uint64_t *count) {
    if (!ctx || !count) return fail(RT_DP_ENULL, "rt_dp_box_compact_tiled: NULL argument");
    *count = 0;
    if (n == 0) return RT_DP_OK;
    if (!tiled_results) return fail(RT_DP_ENULL, "rt_dp_box_compact_tiled: the result buffer is NULL");
    const uint32_t *words = static_cast<const uint32_t *>(tiled_results);   // word k of tile 0 starts at k * RT_DP_TILE_OPS
    const float *tm[4];
    for (int k = 0; k < 4; ++k) tm[k] = reinterpret_cast<const float *>(words + (size_t)k * kTileOps);
    return box_compact("rt_dp_box_compact_tiled", ctx, words + (size_t)4 * kTileOps, tm, (uint32_t)kBoxOutWords, n, item, children, out_item, out_child,
                       out_tmin, capacity, count);
}

// rt_dp_box_compact for callers that cannot allocate: offsets in a file-scope device array (compact.hip), the
// count delivered to device memory, nothing but kernel launches (and one memset for an empty batch) on the path.
int rt_dp_box_compact_static(rt_dp_ctx *ctx, const rt_dp_box_out *results, uint64_t n, const uint32_t *item, const uint32_t *children,
                             uint32_t *out_item, uint32_t *out_child, float *out_tmin, uint64_t capacity, uint64_t *d_count, void *hip_stream) {
    if (!ctx || !results || !d_count) return fail(RT_DP_ENULL, "rt_dp_box_compact_static: NULL argument");
    if (n > compact_static_max_ops())
        return fail(RT_DP_EINVAL, "rt_dp_box_compact_static: n = %llu exceeds the static scratch (at most %llu ops per call); use rt_dp_box_compact",
                    (unsigned long long)n, (unsigned long long)compact_static_max_ops());
    if (reinterpret_cast<uintptr_t>(d_count) & 7u) return fail(RT_DP_EALIGN, "rt_dp_box_compact_static: d_count must be 8-byte aligned");
    if (n && (!results->flags || (capacity && (!out_item || !out_child)))) return fail(RT_DP_ENULL, "rt_dp_box_compact_static: NULL array");
    uintptr_t acc = reinterpret_cast<uintptr_t>(results->flags) | reinterpret_cast<uintptr_t>(item) | reinterpret_cast<uintptr_t>(children) |
                    reinterpret_cast<uintptr_t>(out_item) | reinterpret_cast<uintptr_t>(out_child) | reinterpret_cast<uintptr_t>(out_tmin);
    if (n && out_tmin)
        for (int k = 0; k < 4; ++k) {
            if (!results->tmin_out[k]) return fail(RT_DP_ENULL, "rt_dp_box_compact_static: out_tmin needs all four tmin_out arrays");
            acc |= reinterpret_cast<uintptr_t>(results->tmin_out[k]);
        }
    if (n && (acc & 3u)) return fail(RT_DP_EALIGN, "rt_dp_box_compact_static: arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint64_t), s));
        return RT_DP_OK;
    }
    const float *tm[4] = {results->tmin_out[0], results->tmin_out[1], results->tmin_out[2], results->tmin_out[3]};
    HIP_TRY(compact_hits_static(ctx->device, results->flags, tm, n, item, children, capacity, out_item, out_child, out_tmin,
                                reinterpret_cast<unsigned long long *>(d_count), s));
    return RT_DP_OK;
}

int rt_dp_trace_closest(rt_dp_ctx *ctx, const rt_dp_bvh4 *bvh, const float *const origin[3], const float *const dir[3],
                        uint32_t n_rays, float extent, uint32_t flags, float *out_t, uint32_t *out_tri, uint64_t stats[4]) {
    if (!ctx || !bvh || !origin || !dir) return fail(RT_DP_ENULL, "rt_dp_trace_closest: NULL argument");
    if (bvh->struct_size != sizeof(rt_dp_bvh4)) return fail(RT_DP_EINVAL, "rt_dp_bvh4.struct_size does not match this library");
    if (flags & ~RT_DP_TRACE_WAVEFRONT) return fail(RT_DP_EINVAL, "rt_dp_trace_closest: unknown flag bits 0x%x", flags & ~RT_DP_TRACE_WAVEFRONT);
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (n_rays == 0) return RT_DP_OK;
    if (!out_t || !out_tri || !bvh->nodes || (bvh->n_tris && !bvh->tris)) return fail(RT_DP_ENULL, "rt_dp_trace_closest: NULL array");
    if (bvh->n_nodes == 0) return fail(RT_DP_EINVAL, "the hierarchy has no root node");
    if (bvh->n_nodes >= 0x80000000u || bvh->n_tris > (1u << 28)) return fail(RT_DP_EINVAL, "hierarchy too large for 32-bit references");
    uintptr_t acc = reinterpret_cast<uintptr_t>(bvh->nodes) | reinterpret_cast<uintptr_t>(bvh->tris);
    for (int c = 0; c < 3; ++c) {
        if (!origin[c] || !dir[c]) return fail(RT_DP_ENULL, "rt_dp_trace_closest: origin[] / dir[] is NULL");
        acc |= reinterpret_cast<uintptr_t>(origin[c]) | reinterpret_cast<uintptr_t>(dir[c]);
    }
    if (acc & 15u) return fail(RT_DP_EALIGN, "rt_dp_trace_closest: nodes, tris and ray arrays must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out_t) | reinterpret_cast<uintptr_t>(out_tri)) & 3u)
        return fail(RT_DP_EALIGN, "rt_dp_trace_closest: output arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    hipStream_t s = ctx->stream;
    HIP_TRY(hipDeviceSynchronize());  // the caller's arrays may still be in flight on other streams

    const bool wavefront = (flags & RT_DP_TRACE_WAVEFRONT) != 0;
    const size_t ray_bytes = wavefront ? (size_t)n_rays * 64 : 0, best_bytes = wavefront ? (size_t)n_rays * 8 : 0;
    HIP_TRY(hipStreamSynchronize(s));
    int rc = regrow(ctx->d_trace, ctx->trace_bytes, ray_bytes + best_bytes + 256, "ray records");
    if (rc != RT_DP_OK) return rc;
    float *rays = reinterpret_cast<float *>(ctx->d_trace);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(ctx->d_trace + ray_bytes);
    uint32_t *counters = reinterpret_cast<uint32_t *>(ctx->d_trace + ray_bytes + best_bytes);
    {
        uint32_t bad = 0;
        HIP_TRY(hipMemsetAsync(counters, 0, 16, s));
        HIP_TRY(trace_validate(bvh->nodes, bvh->n_nodes, bvh->n_tris, counters, s));
        rc = read_back(ctx, &bad, counters, 4, s);
        if (rc != RT_DP_OK) return rc;
        if (bad)
            return fail(RT_DP_EINVAL, "node %u holds a child reference outside the hierarchy (inner children must have a larger index than their parent)", bad - 1);
    }
    if (!wavefront) {
        unsigned long long *totals = reinterpret_cast<unsigned long long *>(counters);  // [0] box ops, [1] triangle ops, [2] the overflow flag, [3] the ray counter
        uint32_t *overflow = counters + 4;
        HIP_TRY(hipMemsetAsync(counters, 0, 32, s));
        HIP_TRY(trace_dfs(origin, dir, n_rays, extent, bvh->nodes, bvh->tris, out_t, out_tri, totals, overflow, totals + 3, s));
        unsigned long long host[3];
        rc = read_back(ctx, host, counters, 24, s);
        if (rc != RT_DP_OK) return rc;
        if (host[2] & 0xFFFFFFFFull) return fail(RT_DP_EINVAL, "a ray had more than 64 pending entries: the hierarchy is too deep for the depth-first walk");
        if (stats) { stats[1] = host[0]; stats[2] = host[1]; }
        return RT_DP_OK;
    }
    const size_t first_items = (size_t)n_rays + (size_t)n_rays / 2 + 1024;
    for (int q = 0; q < 3; ++q) {
        size_t bytes = ctx->queue_items[q] * 8;
        rc = regrow(ctx->d_queue[q], bytes, first_items * 8, "work queue");
        ctx->queue_items[q] = bytes / 8;
        if (rc != RT_DP_OK) return rc;
    }
    int cur = 0, nxt = 1;
    const int lf = 2;
    HIP_TRY(trace_init(origin, dir, n_rays, extent, rays, best, ctx->d_queue[cur], s));
    uint64_t count = n_rays, levels = 0, box_ops = 0, tri_ops = 0, widest = n_rays;
    while (count) {
        if (++levels > 4096) return fail(RT_DP_EINVAL, "the hierarchy is deeper than 4096 levels (a cycle in the child references?)");
        uint32_t host[4];
        for (;;) {  // repeated only when a queue turned out too small for this level
            HIP_TRY(hipMemsetAsync(counters, 0, 16, s));
            HIP_TRY(trace_box(ctx->d_queue[cur], (uint32_t)count, rays, bvh->nodes, best, ctx->d_queue[nxt], (uint32_t)ctx->queue_items[nxt],
                              ctx->d_queue[lf], (uint32_t)ctx->queue_items[lf], counters, s));
            rc = read_back(ctx, host, counters, 16, s);
            if (rc != RT_DP_OK) return rc;
            if (host[0] <= ctx->queue_items[nxt] && host[1] <= ctx->queue_items[lf]) break;
            if (host[0] > ctx->queue_items[nxt]) {
                size_t bytes = ctx->queue_items[nxt] * 8;
                rc = regrow(ctx->d_queue[nxt], bytes, ((size_t)host[0] + host[0] / 4) * 8, "work queue");
                ctx->queue_items[nxt] = bytes / 8;
                if (rc != RT_DP_OK) return rc;
            }
            if (host[1] > ctx->queue_items[lf]) {
                size_t bytes = ctx->queue_items[lf] * 8;
                rc = regrow(ctx->d_queue[lf], bytes, ((size_t)host[1] + host[1] / 4) * 8, "leaf queue");
                ctx->queue_items[lf] = bytes / 8;
                if (rc != RT_DP_OK) return rc;
            }
        }
        box_ops += count;
        if (host[1]) {
            HIP_TRY(trace_tri(ctx->d_queue[lf], host[1], rays, bvh->tris, best, counters, s));
            rc = read_back(ctx, host, counters, 16, s);
            if (rc != RT_DP_OK) return rc;
            tri_ops += host[2];
        }
        count = host[0];
        if (count > widest) widest = count;
        if (host[1] > widest) widest = host[1];
        const int t = cur; cur = nxt; nxt = t;
    }
    HIP_TRY(trace_finish(best, n_rays, out_t, out_tri, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (stats) { stats[0] = levels; stats[1] = box_ops; stats[2] = tri_ops; stats[3] = widest; }
    return RT_DP_OK;
}

int rt_dp_knn_tree_build_device(rt_dp_ctx *ctx, const float *points, uint32_t n_points, uint32_t leaf_size, uint32_t index_base,
                                float *out_points, uint32_t *out_index, float *out_lo, float *out_hi, uint32_t *out_split_dim,
                                float *out_split_val) {
    if (!ctx || !out_points || !out_index || !out_lo || !out_hi || (!points && n_points)) return fail(RT_DP_ENULL, "rt_dp_knn_tree_build_device: NULL argument");
    if ((out_split_dim == nullptr) != (out_split_val == nullptr)) return fail(RT_DP_ENULL, "out_split_dim and out_split_val go together");
    if (leaf_size == 0 || (leaf_size & 3u)) return fail(RT_DP_EINVAL, "leaf_size = %u is not a positive multiple of 4", leaf_size);
    if ((uint64_t)index_base + n_points > 0xFFFFFFFFull) return fail(RT_DP_EINVAL, "index_base + n_points exceeds the 32-bit index range");
    const uint32_t leaves = rt_dp_knn_tree_leaves(n_points, leaf_size);
    if ((uint64_t)leaves * leaf_size > 0x7FFFFFFFull) return fail(RT_DP_EINVAL, "n_leaves * leaf_size exceeds 2^31 - 1 slots");
    if ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out_points) | reinterpret_cast<uintptr_t>(out_lo) |
         reinterpret_cast<uintptr_t>(out_hi)) & 15u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_build_device: coordinate arrays must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out_index) | reinterpret_cast<uintptr_t>(out_split_dim) | reinterpret_cast<uintptr_t>(out_split_val)) & 3u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_tree_build_device: arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(hipDeviceSynchronize());  // the points may still be in production on a caller stream
    const size_t need = kd_device_workspace(leaves * leaf_size, leaves);
    int rc = regrow(ctx->d_trace, ctx->trace_bytes, need, "index construction");
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(kd_build_device(points, n_points, leaves, leaf_size, index_base, ctx->d_trace, ctx->trace_bytes, out_points, out_index, out_lo, out_hi,
                            out_split_dim, out_split_val, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_DP_OK;
}

int rt_dp_knn_home_leaf(rt_dp_ctx *ctx, const float *queries, uint32_t n_queries, uint32_t n_leaves, const uint32_t *split_dim,
                        const float *split_val, uint32_t *out_home, void *hip_stream) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_knn_home_leaf: ctx is NULL");
    if (n_queries == 0) return RT_DP_OK;
    if (!queries || !out_home || (n_leaves > 1 && (!split_dim || !split_val))) return fail(RT_DP_ENULL, "rt_dp_knn_home_leaf: NULL array");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return fail(RT_DP_EINVAL, "n_leaves = %u is not a power of two", n_leaves);
    if ((reinterpret_cast<uintptr_t>(queries) | reinterpret_cast<uintptr_t>(split_dim) | reinterpret_cast<uintptr_t>(split_val) |
         reinterpret_cast<uintptr_t>(out_home)) & 3u)
        return fail(RT_DP_EALIGN, "rt_dp_knn_home_leaf: arrays must be 4-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(kd_home_leaf(queries, n_queries, n_leaves, split_dim, split_val, out_home, hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream));
    return RT_DP_OK;
}

int rt_dp_bvh4_build_device(rt_dp_ctx *ctx, const float *tris, uint32_t n_tris, uint32_t leaf_size, float *out_nodes, uint32_t *out_n_nodes,
                            float *out_tris, uint32_t *out_order) {
    if (!ctx || !tris || !out_nodes || !out_n_nodes || !out_tris || !out_order) return fail(RT_DP_ENULL, "rt_dp_bvh4_build_device: NULL argument");
    *out_n_nodes = 0;
    if (n_tris == 0 || n_tris > (1u << 28)) return fail(RT_DP_EINVAL, "n_tris must be 1 .. 2^28");
    if (leaf_size == 0 || leaf_size > 8) return fail(RT_DP_EINVAL, "leaf_size must be 1 .. 8");
    if ((reinterpret_cast<uintptr_t>(tris) | reinterpret_cast<uintptr_t>(out_order)) & 3u) return fail(RT_DP_EALIGN, "tris / out_order must be 4-byte aligned");
    if ((reinterpret_cast<uintptr_t>(out_nodes) | reinterpret_cast<uintptr_t>(out_tris)) & 15u)
        return fail(RT_DP_EALIGN, "out_nodes / out_tris must be 16-byte aligned");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    HIP_TRY(hipDeviceSynchronize());  // the triangles may still be in production on a caller stream
    const size_t need = bvh_device_workspace(n_tris);
    int rc = regrow(ctx->d_trace, ctx->trace_bytes, need, "hierarchy construction");
    if (rc != RT_DP_OK) return rc;
    HIP_TRY(bvh_build_device(tris, n_tris, leaf_size, ctx->d_trace, ctx->trace_bytes, out_nodes, rt_dp_bvh4_max_nodes(n_tris), out_n_nodes, out_tris,
                             out_order, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_DP_OK;
}

int rt_dp_kernel_name(rt_dp_ctx *ctx, const rt_dp_op_batch *in, const rt_dp_result_batch *out, int opcode, char *buf, uint64_t len) {
    if (!ctx || !buf || len == 0) return fail(RT_DP_ENULL, "rt_dp_kernel_name: NULL argument");
    if (opcode < 0 || opcode > 3) return fail(RT_DP_EOPCODE, "rt_dp_kernel_name: opcode %d", opcode);
    int rc = check_batch(in, out);
    if (rc != RT_DP_OK) return rc;
    buf[0] = 0;
    const uint64_t n = in->n[opcode];
    if (n == 0) return RT_DP_OK;
    const bool derive = (in->flags & RT_DP_F_DERIVE_RAY) != 0;
    auto lanes = [](uintptr_t acc) { return (acc & 15u) == 0 ? 4 : ((acc & 7u) == 0 ? 2 : 1); };
    if (opcode == RT_DP_OP_QUADBOX) {
        const bool tiled = (in->flags & RT_DP_F_BOX_TILED) != 0;
        uintptr_t acc = 0;
        if (!tiled) {
            BoxArgs a;
            fill_box(a, in->box, out->box, derive);
            for (int f = 0; f < kBoxFields; ++f) acc |= reinterpret_cast<uintptr_t>(a.in[f]);
            for (int k = 0; k < 4; ++k) acc |= reinterpret_cast<uintptr_t>(a.tmin[k]);
            acc |= reinterpret_cast<uintptr_t>(a.flags);
        }
        kernel_name_ops(true, tiled, derive, lanes(acc), ctx->tuning, buf, (size_t)len);
    } else if (opcode == RT_DP_OP_TRIANGLE) {
        const bool tiled = (in->flags & RT_DP_F_TRI_TILED) != 0;
        uintptr_t acc = 0;
        if (!tiled) {
            TriArgs a;
            fill_tri(a, in->tri, out->tri, derive);
            for (int f = 0; f < kTriFields; ++f) acc |= reinterpret_cast<uintptr_t>(a.in[f]);
            acc |= reinterpret_cast<uintptr_t>(a.k) | reinterpret_cast<uintptr_t>(a.t_num) | reinterpret_cast<uintptr_t>(a.t_denom) | reinterpret_cast<uintptr_t>(a.hit);
        }
        kernel_name_ops(false, tiled, derive, lanes(acc), ctx->tuning, buf, (size_t)len);
    } else if (!ctx->extended) {
        snprintf(buf, (size_t)len, "(memset: the baseline datapath has no distance ops)");
    } else {
        kernel_name_vec(opcode == RT_DP_OP_EUCLIDEAN, n, ctx->tuning, true, buf, (size_t)len);
    }
    return RT_DP_OK;
}

int rt_dp_set_option(rt_dp_ctx *ctx, const char *key, int value) {
    if (!ctx || !key) return fail(RT_DP_ENULL, "rt_dp_set_option: NULL argument");
    if (!strcmp(key, "variant")) {
        if (!(value == 0 || value == 10 || (value >= 21 && value <= 28 && value != 26))) return fail(RT_DP_EINVAL, "variant must be 0, 10, 21..25, 27 or 28 (kernel shapes)");
        ctx->tuning.variant = value;
    } else if (!strcmp(key, "wgs_per_cu")) {
        if (!(value == 0 || value == 99 || (value >= 3 && value <= 16))) return fail(RT_DP_EINVAL, "wgs_per_cu must be 0 (default), 3..16, or 99 (no cap)");
        ctx->tuning.wgs_per_cu = value;
    } else if (!strcmp(key, "stream_mode")) {
        if (value < 0 || value > 3) return fail(RT_DP_EINVAL, "stream_mode must be 0 (automatic), 1 (partition), 2 (evaluate, then fold) or 3 (single pass)");
        ctx->stream_mode = value;
    } else if (!strcmp(key, "vec_passes")) {
        if (value != 0 && value != 2 && value != 4 && value != 8) return fail(RT_DP_EINVAL, "vec_passes must be 0, 2, 4 or 8");
        ctx->tuning.vec_passes = value;
    } else if (!strcmp(key, "debug_chain_skip")) {
        if (value < -1) return fail(RT_DP_EINVAL, "debug_chain_skip must be -1 (off) or a workgroup index");
        ctx->debug_chain_skip = value;
    } else {
        return fail(RT_DP_EINVAL, "unknown option '%s'", key);
    }
    return RT_DP_OK;
}

int rt_dp_get_option(rt_dp_ctx *ctx, const char *key, int *value) {
    if (!ctx || !key || !value) return fail(RT_DP_ENULL, "rt_dp_get_option: NULL argument");
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    if (!strcmp(key, "variant")) *value = ctx->tuning.variant;
    else if (!strcmp(key, "wgs_per_cu")) *value = ctx->tuning.wgs_per_cu;
    else if (!strcmp(key, "stream_mode")) *value = ctx->stream_mode;
    else if (!strcmp(key, "stream_giveups")) *value = (int)(ctx->stream_giveups > 0x7FFFFFFFull ? 0x7FFFFFFFull : ctx->stream_giveups);
    else if (!strcmp(key, "vec_passes")) *value = ctx->tuning.vec_passes ? ctx->tuning.vec_passes : 0;
    else if (!strcmp(key, "ops_per_lane")) *value = ctx->tuning.ops_per_lane ? ctx->tuning.ops_per_lane : 1;
    else if (!strcmp(key, "nontemporal")) *value = ctx->tuning.nontemporal == 2 ? 0 : 1;
    else if (!strcmp(key, "vec_tile")) *value = ctx->tuning.vec_tile > 0 ? (ctx->tuning.vec_tile < 64 ? 64 : ctx->tuning.vec_tile) : 512;
    else if (!strcmp(key, "vec_chain_giveups")) {
        unsigned long long h = 0;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(&h, ctx->d_chain_hdr, sizeof h, hipMemcpyDeviceToHost));
        *value = h > 0x7FFFFFFFull ? 0x7FFFFFFF : (int)h;
    } else if (!strcmp(key, "chain_fallbacks")) {
        *value = ctx->chain_fallbacks > 0x7FFFFFFFull ? 0x7FFFFFFF : (int)ctx->chain_fallbacks;
    } else if (!strcmp(key, "box_wgs_per_cu")) *value = resolved_residency(ctx->tuning, true);
    else if (!strcmp(key, "tri_wgs_per_cu")) *value = resolved_residency(ctx->tuning, false);
    else if (!strcmp(key, "stream_waves_per_cu")) *value = stream_eval_waves_per_cu();
    else return fail(RT_DP_EINVAL, "unknown option '%s'", key);
    return RT_DP_OK;
}

// ---------------------------------------------------------------------------
// Record-at-a-time stream: partition by opcode (stable), evaluate, scatter.
int rt_dp_eval_bundles(rt_dp_ctx *ctx, const rt_dp_input_bundle *in, rt_dp_output_bundle *out, uint64_t n) {
    if (!ctx) return fail(RT_DP_ENULL, "rt_dp_eval_bundles: ctx is NULL");
    if (n == 0) return RT_DP_OK;
    if (!in || !out) return fail(RT_DP_ENULL, "rt_dp_eval_bundles: in/out is NULL");
    if (n > kMaxOps) return fail(RT_DP_EINVAL, "n = %llu is not representable (limit 2^40)", (unsigned long long)n);
    for (uint64_t i = 0; i < n; ++i)
        if (in[i].opcode > 3u)
            return fail(RT_DP_EOPCODE, "record %llu has opcode %u; the RTL's opcode field is 2 bits wide", (unsigned long long)i, in[i].opcode);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(RT_DP_EHIP, "cannot select device %d", ctx->device);
    // host records: staged through device memory in pieces and evaluated by the device-resident path
    // (stable per opcode, so the accumulators run on from piece to piece like from call to call)
    const uint64_t piece = 1ull << 22;
    const uint64_t cap = n < piece ? n : piece;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int rc = regrow(ctx->d_records, ctx->records_bytes, cap * (sizeof(rt_dp_input_bundle) + sizeof(rt_dp_output_bundle)), "record staging");
    if (rc != RT_DP_OK) return rc;
    rt_dp_input_bundle *d_in = reinterpret_cast<rt_dp_input_bundle *>(ctx->d_records);
    rt_dp_output_bundle *d_out = reinterpret_cast<rt_dp_output_bundle *>(ctx->d_records + cap * sizeof(rt_dp_input_bundle));
    if (n * (sizeof(rt_dp_input_bundle) + sizeof(rt_dp_output_bundle)) <= kPinnedStage) {
        // Small streams (the testbench's suites are 100 records) are bound by latency: the records cross the link through a
        // pinned mirror, in and out on the context's stream, and the call synchronises once.
        if (!ctx->h_stage) {
            hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage), kPinnedStage, hipHostMallocDefault);
            if (e != hipSuccess) return fail(RT_DP_ENOMEM, "hipHostMalloc(%zu) for staging failed: %s", kPinnedStage, hipGetErrorString(e));
        }
        const size_t in_bytes = n * sizeof(rt_dp_input_bundle), out_bytes = n * sizeof(rt_dp_output_bundle);
        memcpy(ctx->h_stage, in, in_bytes);
        HIP_TRY(hipMemcpyAsync(d_in, ctx->h_stage, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        bool copied = false;
        rc = bundles_device(ctx, d_in, d_out, n, true, ctx->h_stage + align_up(in_bytes), &copied);
        if (rc != RT_DP_OK) return rc;
        if (copied) memcpy(out, ctx->h_stage + align_up(in_bytes), out_bytes);
        else HIP_TRY(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
        return RT_DP_OK;
    }
    for (uint64_t off = 0; off < n; off += piece) {
        const uint64_t m = (n - off < piece) ? (n - off) : piece;
        HIP_TRY(hipMemcpy(d_in, in + off, m * sizeof(rt_dp_input_bundle), hipMemcpyHostToDevice));
        rc = rt_dp_eval_bundles_device(ctx, d_in, d_out, m);
        if (rc != RT_DP_OK) return rc;
        HIP_TRY(hipMemcpy(out + off, d_out, m * sizeof(rt_dp_output_bundle), hipMemcpyDeviceToHost));
    }
    return RT_DP_OK;
}

}  // extern "C"
