/* One opcode-sorted stream over several datapath contexts, from plain C: one host thread + one context (and its
 * HIP stream) per GPU, no communication between them.
 *
 *     multi_gpu [--contexts N] [--log2-units L] [--tiled]
 *
 * The reference drives ONE datapath with a producer and a consumer thread
 * (src/test/scala/raytracer_datapath/Datapath_test.scala:476-496).  Here rt_dp_shard_plan cuts the stream into N
 * parts (the beat segments on vector-pair boundaries, where the accumulator registers hold +0), thread r creates a
 * context on GPU r % rt_dp_device_count() and evaluates part r straight from / into the whole stream's host
 * arrays (pointer + offset, so the parts' results are stitched in order by construction), and the outcome is
 * compared bit for bit with the whole stream evaluated by a single context.  Exit status 0 = identical.
 * On a one-GPU machine all N contexts share the GPU; on a node with 8 GPUs `--contexts 8` uses them all.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rt_dp.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {  /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 0x2545F4914F6CDD1Dull) >> 32);
}
static float uni(float lo, float hi) { return lo + (hi - lo) * (float)(rnd() >> 8) * (1.0f / 16777216.0f); }

typedef struct stream {
    uint64_t n[4];
    int tiled;
    float *box_in[34], *box_tiled_in;   /* plain: 34 arrays; tiled: one buffer */
    float *tri_in[15], *tri_tiled_in;
    uint32_t *tri_k;
    float *ea, *eb, *aa, *ab;
    uint32_t *ectl, *actl;
} stream;

typedef struct results {
    float *tmin[4]; uint32_t *flags; uint32_t *box_tiled;
    float *t_num, *t_denom; uint32_t *hit; uint32_t *tri_tiled;
    float *acc, *dot, *norm;
} results;

static void *xmalloc(size_t bytes) {
    void *p = calloc(bytes ? bytes : 4, 1);
    if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
    return p;
}

/* beats of random vector pairs of 1..128 elements, split like the reference's job splitters
 * (src/test/scala/raytracer_datapath/raytracer_gold/SW_Data.scala:213-308) */
static uint64_t make_beats(uint64_t want, int width, float **a, float **b, uint32_t **ctl) {
    *a = xmalloc(want * width * 4); *b = xmalloc(want * width * 4); *ctl = xmalloc(want * 4);
    uint64_t beat = 0;
    while (beat < want) {
        int len = 1 + (int)(rnd() % 128u);
        while (len > 0 && beat < want) {
            const int here = len < width ? len : width;
            for (int j = 0; j < here; ++j) { (*a)[beat * width + j] = uni(-1e4f, 1e4f); (*b)[beat * width + j] = uni(-1e4f, 1e4f); }
            len -= here;
            (*ctl)[beat] = RT_DP_CTL((1u << here) - 1u, len == 0 || beat + 1 == want);
            ++beat;
        }
    }
    return beat;
}

static void make_stream(stream *s, uint64_t units, int tiled) {
    const uint64_t q = units / 4;
    memset(s, 0, sizeof *s);
    s->tiled = tiled;
    s->n[RT_DP_OP_QUADBOX] = q; s->n[RT_DP_OP_TRIANGLE] = q;
    float *bx[34], *tr[15];
    for (int f = 0; f < 34; ++f) bx[f] = xmalloc(q * 4);
    for (int f = 0; f < 15; ++f) tr[f] = xmalloc(q * 4);
    s->tri_k = xmalloc(q * 4);
    for (uint64_t i = 0; i < q; ++i) {
        for (int c = 0; c < 3; ++c) {
            bx[c][i] = uni(-100, 100);
            const float d = (rnd() % 97u == 0) ? 0.0f : uni(-100, 100);   /* some axis-parallel rays: inv = Inf */
            bx[3 + c][i] = d; bx[6 + c][i] = 1.0f / d;
        }
        bx[9][i] = 4000.0f;
        for (int b = 0; b < 4; ++b)
            for (int c = 0; c < 3; ++c) {
                float lo = uni(-1e4f, 1e4f), hi = uni(-1e4f, 1e4f);
                if (lo > hi) { float t = lo; lo = hi; hi = t; }
                bx[10 + 6 * b + c][i] = lo; bx[13 + 6 * b + c][i] = hi;
            }
        for (int f = 0; f < 15; ++f) tr[f][i] = uni(-1000, 1000);
        const uint32_t kz = rnd() % 3u, kx = (kz + 1) % 3u, ky = (kx + 1) % 3u;
        s->tri_k[i] = kx | ky << 2 | kz << 4;
    }
    if (tiled) {
        s->box_tiled_in = xmalloc(RT_DP_TILED_WORDS(q, RT_DP_BOX_TILE_IN_WORDS) * 4);
        s->tri_tiled_in = xmalloc(RT_DP_TILED_WORDS(q, RT_DP_TRI_TILE_IN_WORDS) * 4);
        for (uint64_t i = 0; i < q; ++i) {
            for (int f = 0; f < 34; ++f) s->box_tiled_in[RT_DP_TILED_INDEX(i, f, RT_DP_BOX_TILE_IN_WORDS)] = bx[f][i];
            for (int f = 0; f < 15; ++f) s->tri_tiled_in[RT_DP_TILED_INDEX(i, f, RT_DP_TRI_TILE_IN_WORDS)] = tr[f][i];
            memcpy(&s->tri_tiled_in[RT_DP_TILED_INDEX(i, 15, RT_DP_TRI_TILE_IN_WORDS)], &s->tri_k[i], 4);
        }
        for (int f = 0; f < 34; ++f) free(bx[f]);
        for (int f = 0; f < 15; ++f) free(tr[f]);
    } else {
        memcpy(s->box_in, bx, sizeof bx);
        memcpy(s->tri_in, tr, sizeof tr);
    }
    s->n[RT_DP_OP_EUCLIDEAN] = make_beats(q, 16, &s->ea, &s->eb, &s->ectl);
    s->n[RT_DP_OP_ANGULAR] = make_beats(q, 8, &s->aa, &s->ab, &s->actl);
}

static void make_results(results *r, const stream *s) {
    memset(r, 0, sizeof *r);
    const uint64_t nb = s->n[RT_DP_OP_QUADBOX], nt = s->n[RT_DP_OP_TRIANGLE];
    if (s->tiled) {
        r->box_tiled = xmalloc(RT_DP_TILED_WORDS(nb, RT_DP_BOX_TILE_OUT_WORDS) * 4);
        r->tri_tiled = xmalloc(RT_DP_TILED_WORDS(nt, RT_DP_TRI_TILE_OUT_WORDS) * 4);
    } else {
        for (int k = 0; k < 4; ++k) r->tmin[k] = xmalloc(nb * 4);
        r->flags = xmalloc(nb * 4);
        r->t_num = xmalloc(nt * 4); r->t_denom = xmalloc(nt * 4); r->hit = xmalloc(nt * 4);
    }
    r->acc = xmalloc(s->n[RT_DP_OP_EUCLIDEAN] * 4);
    r->dot = xmalloc(s->n[RT_DP_OP_ANGULAR] * 4); r->norm = xmalloc(s->n[RT_DP_OP_ANGULAR] * 4);
}

/* part [rg->start, rg->stop) of every segment, as a batch over the whole stream's arrays */
static void part_batch(const stream *s, results *r, const rt_dp_shard_range *rg, rt_dp_op_batch *in, rt_dp_result_batch *out) {
    memset(in, 0, sizeof *in); memset(out, 0, sizeof *out);
    in->struct_size = sizeof *in; out->struct_size = sizeof *out;
    for (int op = 0; op < 4; ++op) in->n[op] = rg->stop[op] - rg->start[op];
    const uint64_t b0 = rg->start[RT_DP_OP_QUADBOX], t0 = rg->start[RT_DP_OP_TRIANGLE];
    const uint64_t e0 = rg->start[RT_DP_OP_EUCLIDEAN], a0 = rg->start[RT_DP_OP_ANGULAR];
    if (s->tiled) {   /* a part that starts on a tile boundary is itself a tiled segment */
        in->flags |= RT_DP_F_BOX_TILED | RT_DP_F_TRI_TILED;
        in->box.org[0] = s->box_tiled_in + b0 * RT_DP_BOX_TILE_IN_WORDS;
        out->box.tmin_out[0] = (float *)(r->box_tiled + b0 * RT_DP_BOX_TILE_OUT_WORDS);
        in->tri.org[0] = s->tri_tiled_in + t0 * RT_DP_TRI_TILE_IN_WORDS;
        out->tri.t_num = (float *)(r->tri_tiled + t0 * RT_DP_TRI_TILE_OUT_WORDS);
    } else {
        for (int c = 0; c < 3; ++c) { in->box.org[c] = s->box_in[c] + b0; in->box.dir[c] = s->box_in[3 + c] + b0; in->box.inv[c] = s->box_in[6 + c] + b0; }
        in->box.extent = s->box_in[9] + b0;
        for (int b = 0; b < 4; ++b)
            for (int f = 0; f < 6; ++f) in->box.aabb[b][f] = s->box_in[10 + 6 * b + f] + b0;
        for (int k = 0; k < 4; ++k) out->box.tmin_out[k] = r->tmin[k] + b0;
        out->box.flags = r->flags + b0;
        for (int c = 0; c < 3; ++c) { in->tri.org[c] = s->tri_in[c] + t0; in->tri.shear[c] = s->tri_in[3 + c] + t0; }
        for (int v = 0; v < 3; ++v)
            for (int c = 0; c < 3; ++c) in->tri.vtx[v][c] = s->tri_in[6 + 3 * v + c] + t0;
        in->tri.k = s->tri_k + t0;
        out->tri.t_num = r->t_num + t0; out->tri.t_denom = r->t_denom + t0; out->tri.hit = r->hit + t0;
    }
    in->euc.a = s->ea + e0 * 16; in->euc.b = s->eb + e0 * 16; in->euc.ctl = s->ectl + e0; out->euc.accumulator = r->acc + e0;
    in->ang.a = s->aa + a0 * 8; in->ang.b = s->ab + a0 * 8; in->ang.ctl = s->actl + a0;
    out->ang.dot_product = r->dot + a0; out->ang.norm = r->norm + a0;
}

typedef struct worker {
    pthread_t thread;
    int rank, device, rc;
    const stream *s;
    results *r;
    rt_dp_shard_range range;
    char err[256];
} worker;

static void *work(void *arg) {
    worker *w = arg;
    rt_dp_config cfg = {sizeof(rt_dp_config), w->device, 0, 1, 16, 0};
    rt_dp_ctx *ctx = NULL;
    w->rc = rt_dp_create(&cfg, &ctx);            /* a fresh context: accumulators at +0, like every part expects */
    if (w->rc == RT_DP_OK) {
        rt_dp_op_batch in; rt_dp_result_batch out;
        part_batch(w->s, w->r, &w->range, &in, &out);
        w->rc = rt_dp_eval_ctx(ctx, &in, &out);
    }
    if (w->rc != RT_DP_OK) snprintf(w->err, sizeof w->err, "%s", rt_dp_last_error());  /* the message is per thread */
    rt_dp_destroy(ctx);
    return NULL;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int same(const void *a, const void *b, size_t bytes, const char *what) {
    if (bytes == 0 || memcmp(a, b, bytes) == 0) return 1;
    printf("MISMATCH in %s\n", what);
    return 0;
}

int main(int argc, char **argv) {
    int contexts = 2, log2_units = 18, tiled = 0;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--contexts") && i + 1 < argc) contexts = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--log2-units") && i + 1 < argc) log2_units = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--tiled")) tiled = 1;
        else { fprintf(stderr, "usage: %s [--contexts N] [--log2-units L] [--tiled]\n", argv[0]); return 2; }
    }
    if (contexts < 1 || contexts > 64 || log2_units < 4 || log2_units > 26) { fprintf(stderr, "contexts 1..64, log2-units 4..26\n"); return 2; }
    const int devices = rt_dp_device_count();
    if (devices <= 0) { fprintf(stderr, "no GPU\n"); return 2; }

    stream s;
    make_stream(&s, 1ull << log2_units, tiled);
    results whole, parts;
    make_results(&whole, &s);
    make_results(&parts, &s);

    /* the whole stream on one context */
    worker one = {.rank = 0, .device = 0, .s = &s, .r = &whole};
    for (int op = 0; op < 4; ++op) { one.range.start[op] = 0; one.range.stop[op] = s.n[op]; }
    work(&one);
    if (one.rc != RT_DP_OK) { fprintf(stderr, "whole stream: %s\n", one.err); return 3; }

    /* the same stream in `contexts` parts, one thread + one context each */
    rt_dp_shard_range *plan = xmalloc(sizeof(rt_dp_shard_range) * contexts);
    int rc = rt_dp_shard_plan(s.n, s.ectl, s.actl, (uint32_t)contexts, tiled ? (RT_DP_F_BOX_TILED | RT_DP_F_TRI_TILED) : 0u, plan);
    if (rc != RT_DP_OK) { fprintf(stderr, "rt_dp_shard_plan: %s\n", rt_dp_last_error()); return 3; }
    worker *w = xmalloc(sizeof(worker) * contexts);
    const double t0 = now();
    for (int r = 0; r < contexts; ++r) {
        w[r].rank = r; w[r].device = r % devices; w[r].s = &s; w[r].r = &parts; w[r].range = plan[r];
        if (pthread_create(&w[r].thread, NULL, work, &w[r]) != 0) { fprintf(stderr, "pthread_create failed\n"); return 3; }
    }
    int ok = 1;
    for (int r = 0; r < contexts; ++r) {
        pthread_join(w[r].thread, NULL);
        if (w[r].rc != RT_DP_OK) { fprintf(stderr, "part %d on GPU %d: %s\n", r, w[r].device, w[r].err); ok = 0; }
    }
    const double wall = now() - t0;
    if (!ok) return 3;
    for (int r = 0; r < contexts; ++r)
        printf("part %d (GPU %d): tri [%llu, %llu) box [%llu, %llu) euclid [%llu, %llu) angular [%llu, %llu)\n", r, w[r].device,
               (unsigned long long)plan[r].start[0], (unsigned long long)plan[r].stop[0], (unsigned long long)plan[r].start[1],
               (unsigned long long)plan[r].stop[1], (unsigned long long)plan[r].start[2], (unsigned long long)plan[r].stop[2],
               (unsigned long long)plan[r].start[3], (unsigned long long)plan[r].stop[3]);

    const uint64_t nb = s.n[RT_DP_OP_QUADBOX], nt = s.n[RT_DP_OP_TRIANGLE], ne = s.n[RT_DP_OP_EUCLIDEAN], na = s.n[RT_DP_OP_ANGULAR];
    if (tiled) {   /* the words behind the last op of the last tile are untouched (zero) in both */
        ok &= same(whole.box_tiled, parts.box_tiled, RT_DP_TILED_WORDS(nb, RT_DP_BOX_TILE_OUT_WORDS) * 4, "box results (tiled)");
        ok &= same(whole.tri_tiled, parts.tri_tiled, RT_DP_TILED_WORDS(nt, RT_DP_TRI_TILE_OUT_WORDS) * 4, "triangle results (tiled)");
    } else {
        for (int k = 0; k < 4; ++k) ok &= same(whole.tmin[k], parts.tmin[k], nb * 4, "tmin_out");
        ok &= same(whole.flags, parts.flags, nb * 4, "box flags");
        ok &= same(whole.t_num, parts.t_num, nt * 4, "t_num") & same(whole.t_denom, parts.t_denom, nt * 4, "t_denom") & same(whole.hit, parts.hit, nt * 4, "triangle_hit");
    }
    ok &= same(whole.acc, parts.acc, ne * 4, "euclidean_accumulator");
    ok &= same(whole.dot, parts.dot, na * 4, "angular_dot_product") & same(whole.norm, parts.norm, na * 4, "angular_norm");
    /* not all zeros: the comparison would be empty otherwise */
    uint64_t nonzero = 0;
    for (uint64_t i = 0; i < ne; ++i) nonzero += whole.acc[i] != 0.0f;
    ok &= nonzero > ne / 2;
    printf("%d context(s) on %d GPU(s), %llu units (%s layout, host arrays): %s; parts took %.3f s wall (staging over the host link included)\n",
           contexts, devices < contexts ? devices : contexts, (unsigned long long)(nb + nt + ne + na), tiled ? "tiled" : "plain",
           ok ? "bit-equal to the whole stream on one context" : "DIFFERENT from the whole stream", wall);
    return ok ? 0 : 1;
}
