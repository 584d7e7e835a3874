/* Several host threads, each with a context of its own, on ONE GPU at the same time (include/rxr.h: a context is used by one thread at
 * a time; different contexts share nothing a caller must guard -- the library's own process-wide state, the run-time compiler's cache
 * and job table, is locked inside).  Every thread renders its own frame (its own quads and colours) forty times -- plain uploads,
 * streamed hand-overs, a two-member context -- while the others do the same, and every frame must be byte-identical to the one the
 * main thread rendered for it alone, before any thread ran.  Built and run by tests/test_gpu_abi_stream.py. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rxr.h"

#define W 192u
#define H 112u
#define NB 6u
#define NT 4u
#define FRAMES 40u

typedef struct Scene {
    float pv[NB][4][4], uv[NB][4][2], nrm[NB][4][3];
    uint32_t idx[NB][6];
    rxr_edges edges[NB][2];
    rxr_batch3d batches[NB];
    rxr_frame frame;
    uint8_t ref[W * H * 4];
    uint32_t cap_v[NB], cap_t[NB];
    uint32_t thread;
    int failures;
} Scene;

static Scene scenes[NT];

static void identity(float *m) {
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

static void build(Scene *s, uint32_t thread) {
    memset(s, 0, sizeof *s);
    s->thread = thread;
    for (uint32_t b = 0; b < NB; ++b) {
        /* overlapping quads at one depth (ties go to the smaller index), shifted per batch AND per thread */
        const float x0 = 5.0f + 13.0f * (float)b + 3.0f * (float)thread, y0 = 6.0f + 7.0f * (float)((b + thread) % 4u), x1 = x0 + 60.0f, y1 = y0 + 70.0f;
        const float v[4][4] = {{x0, y0, 0.5f, 1.0f}, {x1, y0, 0.5f, 1.0f}, {x1, y1, 0.5f, 1.0f}, {x0, y1, 0.5f, 1.0f}};
        memcpy(s->pv[b], v, sizeof v);
        const float uv[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
        memcpy(s->uv[b], uv, sizeof uv);
        for (int i = 0; i < 4; ++i) s->nrm[b][i][2] = 1.0f;
        const uint32_t id[6] = {0, 1, 2, 0, 2, 3};
        memcpy(s->idx[b], id, sizeof id);
        for (int t = 0; t < 2; ++t) {
            s->edges[b][t].c[0] = s->edges[b][t].c[1] = s->edges[b][t].c[2] = 1.0f;
            s->edges[b][t].visible = 1u;
        }
        rxr_batch3d *o = &s->batches[b];
        o->projected_vertices = &s->pv[b][0][0];
        o->clipped_uvs = &s->uv[b][0][0];
        o->clipped_normals = &s->nrm[b][0][0];
        o->clipped_indices = s->idx[b];
        o->edges = s->edges[b];
        o->n_vertices = 4;
        o->n_triangles = 2;
        o->has_bounding_box = 1;
        o->bounding_box[2] = (float)W;
        o->bounding_box[3] = (float)H;
        o->source.kind = RXR_SOURCE_PIXEL;
        o->source.pixel[0] = (uint8_t)(30u + 35u * b); o->source.pixel[1] = (uint8_t)(240u - 50u * thread); o->source.pixel[2] = (uint8_t)(20u + 60u * thread + 5u * b); o->source.pixel[3] = 255;
        o->ambient_color[0] = o->ambient_color[1] = o->ambient_color[2] = 1.0f;
        o->shader = -1;
        o->list = RXR_LIST_STATIC;
        o->chunk = -1;
        s->cap_v[b] = 4u + 4u * 2u;
        s->cap_t[b] = 3u * 2u;
    }
    s->frame.abi_version = RXR_ABI_VERSION;
    s->frame.width = W;
    s->frame.height = H;
    s->frame.tile_size = 16;
    identity(s->frame.inverse_view);
    identity(s->frame.inverse_projection);
    identity(s->frame.view);
    identity(s->frame.projection);
    s->frame.scaled2 = 1.0f;
    s->frame.flags = RXR_FLAG_D3_ACTIVE;
    s->frame.batches3d = s->batches;
    s->frame.n_batches3d = NB;
}

static void *worker(void *p) {
    Scene *s = (Scene *)p;
    rxr_ctx *ctx = NULL, *multi = NULL;
    static const int ids[2] = {0, 0};
    uint8_t *out = malloc(W * H * 4);
    if (!out || rxr_create(&ctx, 0) != RXR_OK || rxr_create_multi(&multi, ids, 2) != RXR_OK) {
        s->failures = 1000;
        return NULL;
    }
    for (uint32_t f = 0; f < FRAMES; ++f) {
        rxr_ctx *use = (f % 5u == 4u) ? multi : ctx;
        int rc = RXR_OK;
        memset(out, 0, W * H * 4);
        if (use == ctx && f % 3u == 1u) {  /* streamed hand-over, forwards or backwards */
            rc = rxr_stream_begin(ctx, NB, s->cap_v, s->cap_t);
            for (uint32_t k = 0; k < NB && rc == RXR_OK; ++k) (void)rxr_stream_batch3d(ctx, (f & 1u) ? k : NB - 1u - k, &s->batches[(f & 1u) ? k : NB - 1u - k]);
        }
        if (rc == RXR_OK) rc = rxr_rasterize(use, &s->frame, out);
        if (rc != RXR_OK || memcmp(out, s->ref, W * H * 4) != 0) {
            if (s->failures < 3) printf("thread %u frame %u: rc=%d %s (%s)\n", s->thread, f, rc, rc == RXR_OK ? "DIFFERENT" : "failed", rxr_last_error(use));
            ++s->failures;
        }
    }
    rxr_destroy(multi);
    rxr_destroy(ctx);
    free(out);
    return NULL;
}

int main(void) {
    rxr_ctx *ctx = NULL;
    if (rxr_create(&ctx, 0) != RXR_OK) {
        printf("rxr_create failed\n");
        return 2;
    }
    for (uint32_t t = 0; t < NT; ++t) {
        build(&scenes[t], t);
        if (rxr_rasterize(ctx, &scenes[t].frame, scenes[t].ref) != RXR_OK) {
            printf("reference frame %u failed: %s\n", t, rxr_last_error(ctx));
            return 2;
        }
    }
    rxr_destroy(ctx);
    for (uint32_t t = 1; t < NT; ++t)
        if (memcmp(scenes[t].ref, scenes[0].ref, sizeof scenes[0].ref) == 0) {
            printf("FAILED: the threads' frames are not distinct\n");
            return 1;
        }
    pthread_t th[NT];
    for (uint32_t t = 0; t < NT; ++t) pthread_create(&th[t], NULL, worker, &scenes[t]);
    int failures = 0;
    for (uint32_t t = 0; t < NT; ++t) {
        pthread_join(th[t], NULL);
        printf("thread %u: %u frames, %d failures\n", t, FRAMES, scenes[t].failures);
        failures += scenes[t].failures;
    }
    printf(failures ? "FAILED: %d frame(s)\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
