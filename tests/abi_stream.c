/* The streaming hand-over (rxr_stream_begin / rxr_stream_begin_pinned / rxr_stream_batch3d, include/rxr.h) from a plain C caller on a
 * machine WITH a GPU.  One frame of NB overlapping quads AT THE SAME DEPTH (the smaller submission index must win every tie, so a
 * batch that lands at the wrong place in the pools shows) is rendered the plain way, and then handed over batch by batch -- in order,
 * in reverse, from four threads at once, out of page-locked memory with the device pulling it -- and, with something wrong each time
 * (a batch never handed over, one handed over twice, a capacity exceeded, a frame that names other arrays, the promise made on a
 * handle that cannot stream), must come out byte-identical: a stream the frame cannot use is abandoned, never trusted.
 * Built and run by tests/test_gpu_abi_stream.py (gcc -std=c11 -Wall -Werror -pthread). */
#define _POSIX_C_SOURCE 200809L /* posix_memalign under -std=c11 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rxr.h"

#define W 160u
#define H 96u
#define NB 12u

typedef struct Arrays {
    float *pv, *uv, *nrm;
    uint32_t *idx;
    rxr_edges *edges;
} Arrays;

static Arrays arrays[NB], copies[NB], pinned[NB];
static rxr_batch3d batches[NB];
static rxr_frame frame;
static uint8_t ref[W * H * 4], out[W * H * 4];
static int failures = 0;

static void identity(float *m) {
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

static void fill(Arrays *a, uint32_t b) {
    /* a quad of two triangles, shifted by 9 pixels per batch: neighbours overlap; all at z = 0.5 */
    const float x0 = 6.0f + 9.0f * (float)b, y0 = 8.0f + 5.0f * (float)(b % 4u), x1 = x0 + 40.0f, y1 = y0 + 60.0f;
    const float v[4][4] = {{x0, y0, 0.5f, 1.0f}, {x1, y0, 0.5f, 1.0f}, {x1, y1, 0.5f, 1.0f}, {x0, y1, 0.5f, 1.0f}};
    memcpy(a->pv, v, sizeof v);
    const float uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
    memcpy(a->uv, uv, sizeof uv);
    for (int i = 0; i < 4; ++i) {
        a->nrm[3 * i] = 0.0f;
        a->nrm[3 * i + 1] = 0.0f;
        a->nrm[3 * i + 2] = 1.0f;
    }
    const uint32_t id[6] = {0, 1, 2, 0, 2, 3};
    memcpy(a->idx, id, sizeof id);
    for (int t = 0; t < 2; ++t) { /* r = c = 1 for every edge: every pixel of the triangle's box passes; the boxes are the quad */
        memset(&a->edges[t], 0, sizeof(rxr_edges));
        a->edges[t].c[0] = a->edges[t].c[1] = a->edges[t].c[2] = 1.0f;
        a->edges[t].visible = 1u;
    }
}

static int alloc_arrays(Arrays *a, int use_pinned) {
    void *(*get)(size_t) = use_pinned ? rxr_alloc_pinned : malloc;
    a->pv = get(4 * 4 * sizeof(float));
    a->uv = get(4 * 2 * sizeof(float));
    a->nrm = get(4 * 3 * sizeof(float));
    a->idx = get(6 * sizeof(uint32_t));
    a->edges = get(2 * sizeof(rxr_edges));
    return a->pv && a->uv && a->nrm && a->idx && a->edges;
}

static void point_at(const Arrays *set) {
    for (uint32_t b = 0; b < NB; ++b) {
        rxr_batch3d *o = &batches[b];
        memset(o, 0, sizeof *o);
        o->projected_vertices = set[b].pv;
        o->clipped_uvs = set[b].uv;
        o->clipped_normals = set[b].nrm;
        o->clipped_indices = set[b].idx;
        o->edges = set[b].edges;
        o->n_vertices = 4;
        o->n_triangles = 2;
        o->has_bounding_box = 1;
        o->bounding_box[0] = 0.0f; o->bounding_box[1] = 0.0f; o->bounding_box[2] = (float)W; o->bounding_box[3] = (float)H;
        o->source.kind = RXR_SOURCE_PIXEL;
        o->source.pixel[0] = (uint8_t)(40u + 17u * b); o->source.pixel[1] = (uint8_t)(250u - 19u * b); o->source.pixel[2] = (uint8_t)(30u + 11u * b); o->source.pixel[3] = 255;
        o->ambient_color[0] = o->ambient_color[1] = o->ambient_color[2] = 1.0f;
        o->shader = -1;
        o->list = RXR_LIST_STATIC;
        o->chunk = -1;
    }
    memset(&frame, 0, sizeof frame);
    frame.abi_version = RXR_ABI_VERSION;
    frame.width = W;
    frame.height = H;
    frame.tile_size = 16;
    identity(frame.inverse_view);
    identity(frame.inverse_projection);
    identity(frame.view);
    identity(frame.projection);
    frame.scaled2 = 1.0f;
    frame.flags = RXR_FLAG_D3_ACTIVE;
    frame.batches3d = batches;
    frame.n_batches3d = NB;
}

static void check(const char *what, int rc) {
    const int same = rc == RXR_OK && memcmp(out, ref, sizeof ref) == 0;
    printf("%-72s rc=%d %s\n", what, rc, same ? "identical" : "DIFFERENT");
    if (!same) ++failures;
}

static uint32_t cap_v[NB], cap_t[NB];
typedef struct Job {
    rxr_ctx *ctx;
    uint32_t first, step;
} Job;
static void *hand_over(void *p) {
    Job *j = (Job *)p;
    for (uint32_t b = j->first; b < NB; b += j->step) (void)rxr_stream_batch3d(j->ctx, b, &batches[b]);
    return NULL;
}

int main(void) {
    rxr_ctx *ctx = NULL;
    if (rxr_create(&ctx, 0) != RXR_OK) {
        printf("rxr_create failed\n");
        return 2;
    }
    int have_pinned = 1;
    for (uint32_t b = 0; b < NB; ++b) {
        if (!alloc_arrays(&arrays[b], 0) || !alloc_arrays(&copies[b], 0)) return 2;
        have_pinned = have_pinned && alloc_arrays(&pinned[b], 1);
        fill(&arrays[b], b);
        fill(&copies[b], b);
        if (have_pinned) fill(&pinned[b], b);
        cap_v[b] = 4u + 4u * 2u;
        cap_t[b] = 3u * 2u;
    }
    point_at(arrays);
    if (rxr_rasterize(ctx, &frame, ref) != RXR_OK) {
        printf("plain frame failed: %s\n", rxr_last_error(ctx));
        return 2;
    }
    {   /* the frame shows every batch, and the ties went to the smaller index */
        const uint8_t *p = &ref[(30u * W + 40u) * 4u];  /* x = 40: inside batches 0..3, batch 0 wins */
        printf("pixel (40, 30): %u %u %u %u (batch 0's colour scaled by its ambient term)\n", p[0], p[1], p[2], p[3]);
        if (p[3] != 255 || p[1] < p[0]) {
            printf("FAILED: the reference frame is not what the test assumes\n");
            return 1;
        }
    }
    int rc;
#define STREAM(begin_call, body, what) do { memset(out, 0, sizeof out); rc = (begin_call); if (rc == RXR_OK) { body; rc = rxr_rasterize(ctx, &frame, out); } check(what, rc); } while (0)
    STREAM(rxr_stream_begin(ctx, NB, cap_v, cap_t), for (uint32_t b = 0; b < NB; ++b) rxr_stream_batch3d(ctx, b, &batches[b]), "streamed in order");
    STREAM(rxr_stream_begin(ctx, NB, cap_v, cap_t), for (uint32_t b = NB; b-- > 0;) rxr_stream_batch3d(ctx, b, &batches[b]), "streamed in reverse order");
    STREAM(rxr_stream_begin(ctx, NB, cap_v, cap_t), {
        pthread_t th[4];
        Job jobs[4];
        for (uint32_t k = 0; k < 4; ++k) {
            jobs[k].ctx = ctx; jobs[k].first = k; jobs[k].step = 4;
            pthread_create(&th[k], NULL, hand_over, &jobs[k]);
        }
        for (uint32_t k = 0; k < 4; ++k) pthread_join(th[k], NULL);
    }, "streamed from four threads");
    STREAM(rxr_stream_begin(ctx, NB, cap_v, cap_t), for (uint32_t b = 0; b < NB; ++b) if (b != 5) rxr_stream_batch3d(ctx, b, &batches[b]), "a batch never handed over: the frame is taken from scratch");
    STREAM(rxr_stream_begin(ctx, NB, cap_v, cap_t), { for (uint32_t b = 0; b < NB; ++b) rxr_stream_batch3d(ctx, b, &batches[b]);
                                                      if (rxr_stream_batch3d(ctx, 3, &batches[3]) == RXR_OK) ++failures; }, "a batch handed over twice: refused, from scratch");
    {
        uint32_t small[NB];
        memcpy(small, cap_v, sizeof small);
        small[7] = 3;  /* batch 7 has 4 vertices */
        STREAM(rxr_stream_begin(ctx, NB, small, cap_t), for (uint32_t b = 0; b < NB; ++b) rxr_stream_batch3d(ctx, b, &batches[b]), "a capacity exceeded: from scratch");
    }
    STREAM(rxr_stream_begin(ctx, NB, cap_v, cap_t), { for (uint32_t b = 0; b < NB; ++b) rxr_stream_batch3d(ctx, b, &batches[b]); point_at(copies); },
           "the frame names other arrays than the ones streamed: from scratch");
    point_at(arrays);
    {
        uint32_t bad[6] = {0, 1, 2, 0, 2, 9};
        rxr_batch3d wrong = batches[2];
        wrong.clipped_indices = bad;
        memset(out, 0, sizeof out);
        rc = rxr_stream_begin(ctx, NB, cap_v, cap_t);
        for (uint32_t b = 0; b < NB && rc == RXR_OK; ++b)
            if (rxr_stream_batch3d(ctx, b, b == 2 ? &wrong : &batches[b]) != RXR_OK && b != 2 && b < 2) ++failures;
        rc = rxr_rasterize(ctx, &frame, out);  /* the FRAME's batch 2 is fine: from scratch */
        check("a batch handed over with an index out of range (the frame's own is fine)", rc);
    }
    if (have_pinned) {
        point_at(pinned);
        STREAM(rxr_stream_begin_pinned(ctx, NB, cap_v, cap_t), for (uint32_t b = NB; b-- > 0;) rxr_stream_batch3d(ctx, b, &batches[b]),
               "page-locked arrays, pulled by the device (reverse order)");
        STREAM(rxr_stream_begin_pinned(ctx, NB, cap_v, cap_t), for (uint32_t b = 0; b < NB; ++b) if (b != 11) rxr_stream_batch3d(ctx, b, &batches[b]),
               "page-locked arrays, the last batch never handed over: from scratch");
        {
            /* ordinary memory locked afterwards (rxr_pin_host_buffer = hipHostRegister), which include/rxr.h allows under the promise as well:
             * the device's address of such pages need not be the host's, and the pull kernel must read through the former (round-3 advisor
             * finding).  One malloc'ed block holds every array of every batch at odd offsets. */
            static Arrays reg[NB];
            const size_t per = 4 * 4 * sizeof(float) + 4 * 2 * sizeof(float) + 4 * 3 * sizeof(float) + 6 * sizeof(uint32_t) + 2 * sizeof(rxr_edges);
            const size_t bytes = ((size_t)NB * (per + 64) + 8192 + 4095) & ~(size_t)4095;
            uint8_t *block = NULL;
            if (posix_memalign((void **)&block, 4096, bytes) == 0 && rxr_pin_host_buffer(ctx, block, bytes) == RXR_OK) {
                uint8_t *p = block + 272;  /* (not page-aligned) */
                for (uint32_t b = 0; b < NB; ++b) {
                    reg[b].pv = (float *)p; p += 4 * 4 * sizeof(float);
                    reg[b].uv = (float *)p; p += 4 * 2 * sizeof(float);
                    reg[b].nrm = (float *)p; p += 4 * 3 * sizeof(float);
                    reg[b].idx = (uint32_t *)p; p += 6 * sizeof(uint32_t);
                    reg[b].edges = (rxr_edges *)p; p += 2 * sizeof(rxr_edges) + 16;
                    fill(&reg[b], b);
                }
                point_at(reg);
                STREAM(rxr_stream_begin_pinned(ctx, NB, cap_v, cap_t), for (uint32_t b = 0; b < NB; ++b) rxr_stream_batch3d(ctx, b, &batches[b]),
                       "malloc'ed arrays locked with rxr_pin_host_buffer, pulled by the device");
                (void)rxr_synchronize(ctx);
                {   /* the frame really came out of the registered block (2 = pulled by the device), not from scratch */
                    extern int rxr_debug_stream_info(rxr_ctx *);
                    const int mode = rxr_debug_stream_info(ctx);
                    printf("%-72s mode=%d %s\n", "... handed over by the pull kernel", mode, mode == 2 ? "yes" : "NO");
                    if (mode != 2) ++failures;
                }
                point_at(pinned);
                memset(out, 0, sizeof out);
                check("... and the frame that follows, from rxr_alloc_pinned arrays again", rxr_rasterize(ctx, &frame, out));
                rxr_unpin_host_buffer(ctx, block);
            } else {
                printf("(hipHostRegister refused the block: the registered-memory case is skipped)\n");
            }
            free(block);
        }
        point_at(arrays);
        /* the promise broken: ordinary memory handed over as page-locked.  The library verifies every array before the device may
         * touch it (a wrong pointer would be a GPU page fault): refused, and the frame is taken from scratch */
        memset(out, 0, sizeof out);
        rc = rxr_stream_begin_pinned(ctx, NB, cap_v, cap_t);
        int refused = 0;
        for (uint32_t b = 0; b < NB && rc == RXR_OK; ++b) refused += rxr_stream_batch3d(ctx, b, &batches[b]) != RXR_OK;
        if (refused != (int)NB) {
            printf("FAILED: %d of %u hand-overs of ordinary memory under the page-locked promise were accepted\n", (int)NB - refused, NB);
            ++failures;
        }
        check("ordinary arrays under the page-locked promise: every hand-over refused", rxr_rasterize(ctx, &frame, out));
    } else {
        printf("(no page-locked memory: the pinned cases are skipped)\n");
    }
    /* without a begin, and on a handle that cannot stream */
    if (rxr_stream_batch3d(ctx, 0, &batches[0]) == RXR_OK) {
        printf("FAILED: rxr_stream_batch3d without rxr_stream_begin must be refused\n");
        ++failures;
    }
    {
        int ids[2] = {0, 0};
        rxr_ctx *multi = NULL;
        if (rxr_create_multi(&multi, ids, 2) == RXR_OK) {
            rc = rxr_stream_begin(multi, NB, cap_v, cap_t);
            printf("%-72s rc=%d %s\n", "rxr_stream_begin on a multi-device handle", rc, rc == RXR_ERR_UNSUPPORTED ? "refused" : "NOT refused");
            if (rc != RXR_ERR_UNSUPPORTED) ++failures;
            memset(out, 0, sizeof out);
            check("... which renders the frame through rxr_rasterize all the same", rxr_rasterize(multi, &frame, out));
            rxr_destroy(multi);
        }
    }
    /* and the context is fine afterwards */
    memset(out, 0, sizeof out);
    check("the plain call after all of that", rxr_rasterize(ctx, &frame, out));
    rxr_destroy(ctx);
    for (uint32_t b = 0; b < NB && have_pinned; ++b) {
        rxr_free_pinned(pinned[b].pv); rxr_free_pinned(pinned[b].uv); rxr_free_pinned(pinned[b].nrm); rxr_free_pinned(pinned[b].idx); rxr_free_pinned(pinned[b].edges);
    }
    printf(failures ? "FAILED: %d case(s)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
