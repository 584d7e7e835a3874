/* The C ABI from a plain C caller's side, on a machine WITH a GPU: one valid frame, then the same frame with one thing wrong at
 * a time -- every malformed frame must come back as an error status with a message (never a crash, never RXR_OK), and the context
 * must render the valid frame again afterwards.  Built and run by tests/test_gpu_abi_negative.py (gcc -std=c11 -Wall -Werror).
 * The reference panics on most of these inputs (index out of range, step_by(0), Option::unwrap); the boundary reports. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rxr.h"

#define W 64u
#define H 48u

static float verts[3][4] = {{4.0f, 4.0f, 0.5f, 1.0f}, {60.0f, 6.0f, 0.5f, 1.0f}, {30.0f, 44.0f, 0.5f, 1.0f}};
static float uvs[3][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {0.0f, 1.0f}};
/* (a batch without normals shades to NaN -> 0 as in the reference, and a normal along y is flipped towards the camera and meets the
 * hemisphere term 0.5 * (n.y + 1) at 0 in the upper half of this frame: z keeps it at 0.5) */
static float normals[3][3] = {{0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, 1.0f}};
static uint32_t idx[3] = {0, 1, 2};
static rxr_edges edges = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}, 1u}; /* r = c = 1: every pixel of the box passes */
static float verts2[4][2] = {{8.0f, 8.0f}, {24.0f, 8.0f}, {24.0f, 20.0f}, {8.0f, 20.0f}};
static float uvs2[4][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {1.0f, 1.0f}, {0.0f, 1.0f}};
static uint32_t idx2[6] = {0, 1, 2, 0, 2, 3};
static rxr_edges edges2[2] = {{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}, 1u}, {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {1.0f, 1.0f, 1.0f}, 1u}};
static rxr_light light;
static rxr_batch3d b3;
static rxr_batch2d b2;
static rxr_frame base;

static void identity(float *m) {
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

static void make_valid(void) {
    memset(&b3, 0, sizeof b3);
    b3.projected_vertices = &verts[0][0];
    b3.clipped_uvs = &uvs[0][0];
    b3.clipped_normals = &normals[0][0];
    b3.clipped_indices = idx;
    b3.edges = &edges;
    b3.n_vertices = 3;
    b3.n_triangles = 1;
    b3.has_bounding_box = 1;
    b3.bounding_box[0] = 4.0f; b3.bounding_box[1] = 4.0f; b3.bounding_box[2] = 56.0f; b3.bounding_box[3] = 40.0f;
    b3.source.kind = RXR_SOURCE_PIXEL;
    b3.source.pixel[0] = 200; b3.source.pixel[1] = 40; b3.source.pixel[2] = 90; b3.source.pixel[3] = 255;
    b3.ambient_color[0] = b3.ambient_color[1] = b3.ambient_color[2] = 1.0f;
    b3.shader = -1;
    b3.list = RXR_LIST_STATIC;
    b3.chunk = -1;
    memset(&b2, 0, sizeof b2);
    b2.projected_vertices = &verts2[0][0];
    b2.uvs = &uvs2[0][0];
    b2.indices = idx2;
    b2.edges = edges2;
    b2.n_vertices = 4;
    b2.n_triangles = 2;
    b2.has_bounding_box = 1;
    b2.bounding_box[0] = 8.0f; b2.bounding_box[1] = 8.0f; b2.bounding_box[2] = 16.0f; b2.bounding_box[3] = 12.0f;
    b2.mode = RXR_MODE_TRIANGLES;
    b2.source.kind = RXR_SOURCE_PIXEL;
    b2.source.pixel[0] = 10; b2.source.pixel[1] = 220; b2.source.pixel[2] = 30; b2.source.pixel[3] = 255;
    b2.shader = -1;
    b2.chunk = -1;
    memset(&light, 0, sizeof light);
    light.emitting = 1;
    light.intensity = 1.0f;
    light.end_distance = 10.0f;
    memset(&base, 0, sizeof base);
    base.abi_version = RXR_ABI_VERSION;
    base.width = W;
    base.height = H;
    base.tile_size = 16;
    identity(base.inverse_view);
    identity(base.inverse_projection);
    identity(base.view);
    identity(base.projection);
    base.scaled2 = 1.0f;
    base.flags = RXR_FLAG_D2_ACTIVE | RXR_FLAG_D3_ACTIVE;
    base.batches3d = &b3;
    base.n_batches3d = 1;
    base.batches2d = &b2;
    base.n_batches2d = 1;
}

static int failures = 0;
static uint8_t pixels[W * H * 4];

/* the frame is wrong: expect an error status, a message, and no crash */
static void expect_error(rxr_ctx *ctx, const char *what, const rxr_frame *f) {
    int rc = rxr_rasterize(ctx, f, pixels);
    const char *msg = rxr_last_error(ctx);
    printf("%-44s rc=%d %s\n", what, rc, rc < 0 && msg ? msg : "");
    if (rc >= 0 || !msg || !msg[0]) {
        printf("  ^^^ FAILED: a malformed frame must be answered with an error status and a message\n");
        ++failures;
    }
}

static void expect_valid(rxr_ctx *ctx, const char *when) {
    make_valid();
    memset(pixels, 0, sizeof pixels);
    int rc = rxr_rasterize(ctx, &base, pixels);
    const uint8_t *p3 = &pixels[(20u * W + 30u) * 4u], *p2 = &pixels[(12u * W + 12u) * 4u];
    /* the 3D fragment: the batch colour under its own ambient term only (darker than 200, 40, 90, same order); the 2D one: as is */
    int ok = rc == RXR_OK && p3[0] > 90 && p3[0] <= 200 && p3[0] > p3[2] && p3[2] > p3[1] && p3[3] == 255 && p2[0] == 10 && p2[1] == 220 && p2[2] == 30;
    printf("%-44s rc=%d pixel3d=%u,%u,%u,%u pixel2d=%u,%u,%u\n", when, rc, p3[0], p3[1], p3[2], p3[3], p2[0], p2[1], p2[2]);
    if (!ok) {
        printf("  ^^^ FAILED: the valid frame must render (%s)\n", rxr_last_error(ctx));
        ++failures;
    }
}

int main(void) {
    rxr_ctx *ctx = NULL;
    int rc = rxr_create(&ctx, 0);
    if (rc != RXR_OK) {
        printf("rxr_create: %d\n", rc);
        return 2;
    }
    expect_valid(ctx, "valid frame");
    rxr_frame f;
    rxr_batch3d m3;
    rxr_batch2d m2;
    uint32_t bad_idx[3] = {0, 1, 3};
    uint32_t bad_idx2[6] = {0, 1, 2, 0, 2, 4};
#define FRAME(stmt, what) do { make_valid(); f = base; stmt; expect_error(ctx, what, &f); } while (0)
#define BATCH3(stmt, what) do { make_valid(); f = base; m3 = b3; stmt; f.batches3d = &m3; expect_error(ctx, what, &f); } while (0)
#define BATCH2(stmt, what) do { make_valid(); f = base; m2 = b2; stmt; f.batches2d = &m2; expect_error(ctx, what, &f); } while (0)
    FRAME(f.abi_version = RXR_ABI_VERSION - 1u, "abi_version of an older header");
    FRAME(f.width = 0, "width 0");
    FRAME(f.height = 0, "height 0");
    FRAME(f.width = 40000, "width 40000");
    FRAME(f.tile_size = 0, "tile_size 0");
    FRAME(f.batches3d = NULL, "n_batches3d without batches3d");
    FRAME(f.batches2d = NULL, "n_batches2d without batches2d");
    FRAME(f.n_lights = 2, "n_lights without lights");
    FRAME(f.n_occluders = 1, "n_occluders without occluders");
    FRAME(f.n_linedefs = 1, "n_linedefs without linedefs");
    FRAME(f.n_chunks = 1, "n_chunks without chunks");
    FRAME(f.background_kind = 9, "unknown background_kind");
    FRAME(f.background_kind = RXR_BG_HOST_PIXELS, "host background without pixels");
    FRAME(f.n_shader_programs = 3, "n_shader_programs beyond rxr_set_shaders");
    FRAME(f.use_meshes = 1, "use_meshes together with batches3d");
    BATCH3(m3.clipped_indices = NULL, "3D batch: triangles without indices");
    BATCH3(m3.edges = NULL, "3D batch: triangles without edges");
    BATCH3(m3.projected_vertices = NULL, "3D batch: vertices without positions");
    BATCH3(m3.clipped_uvs = NULL, "3D batch: vertices without uvs");
    BATCH3(m3.clipped_indices = bad_idx, "3D batch: vertex index out of range");
    BATCH3(m3.chunk = 0, "3D batch: chunk index without chunks");
    BATCH3((m3.source.kind = RXR_SOURCE_STATIC_TILE, m3.source.index = 5), "3D batch: tile index beyond the tile list");
    BATCH3((m3.source.kind = RXR_SOURCE_DYNAMIC_TILE, m3.source.index = 0), "3D batch: dynamic tile without dynamic tiles");
    BATCH3(m3.source.kind = RXR_HOST_SOURCE_ENTITY_TILE, "3D batch: a host-side-only source kind");
    BATCH2(m2.indices = NULL, "2D batch: triangles without indices");
    {
        /* ABI 5: a 2D batch WITHOUT its Edges records is valid -- the library builds Edges::new([v0,v1,v2],[v1,v2,v0], true) of the projected
         * vertices itself (src/batch/batch2d.rs:413-424).  The frame must equal the one rendered from the same records built here. */
        static uint8_t with_null[W * H * 4], with_built[W * H * 4];
        rxr_edges built[2];
        static uint32_t wound[6] = {0, 2, 1, 0, 3, 2};   /* (the winding whose edge functions are >= 0 inside: edge.rs:28-36) */
        for (int t = 0; t < 2; ++t) {
            for (int k = 0; k < 3; ++k) {
                const float *p = verts2[wound[3 * t + k]], *q = verts2[wound[3 * t + (k + 1) % 3]];
                built[t].a[k] = q[1] - p[1];
                built[t].b[k] = p[0] - q[0];
                built[t].c[k] = q[0] * p[1] - q[1] * p[0];
            }
            built[t].visible = 1u;
        }
        make_valid(); f = base; m2 = b2; m2.edges = NULL; m2.indices = wound; f.batches2d = &m2;
        const int r0 = rxr_rasterize(ctx, &f, with_null);
        m2.edges = built;
        const int r1 = rxr_rasterize(ctx, &f, with_built);
        size_t drawn = 0;
        for (size_t i = 0; i < sizeof with_null; i += 4) drawn += with_null[i + 1] == 220;   /* (the 2D batch's green) */
        const int same = r0 == RXR_OK && r1 == RXR_OK && memcmp(with_null, with_built, sizeof with_null) == 0;
        printf("%-44s rc=%d,%d %s, %zu pixels of the batch\n", "2D batch without Edges records (ABI 5)", r0, r1, same ? "identical" : "DIFFERENT", drawn);
        if (!same || drawn == 0) ++failures;
    }
    BATCH2(m2.projected_vertices = NULL, "2D batch: vertices without positions");
    BATCH2(m2.uvs = NULL, "2D batch: vertices without uvs");
    BATCH2(m2.indices = bad_idx2, "2D batch: vertex index out of range");
    BATCH2(m2.mode = 9, "2D batch: unknown mode");
    BATCH2(m2.chunk = 2, "2D batch: chunk index without chunks");
    /* (a 2D batch whose tile index is beyond the tile list is NOT an error: the 2D loop uses tile_list.get(), rasterizer.rs:673-687) */
    {
        make_valid();
        f = base;
        m2 = b2;
        m2.source.kind = RXR_SOURCE_STATIC_TILE;
        m2.source.index = 1;
        f.batches2d = &m2;
        int r0 = rxr_rasterize(ctx, &f, pixels);
        const uint8_t *p2 = &pixels[(12u * W + 12u) * 4u];
        printf("%-44s rc=%d pixel2d=%u,%u,%u (the 3D fragment shows through)\n", "2D batch: tile index beyond the list: [0,0,0,0]", r0, p2[0], p2[1], p2[2]);
        if (r0 != RXR_OK || p2[0] <= 90) ++failures;
    }
    /* calls in the wrong order or with wrong arguments */
    {
        rxr_ctx *fresh = NULL;
        if (rxr_create(&fresh, 0) == RXR_OK) {
            int r1 = rxr_render_rows(fresh, 0, 8), r2 = rxr_render_download(fresh, pixels), r3 = rxr_download_rows(fresh, pixels, 0, 8);
            printf("%-44s rc=%d,%d,%d\n", "render / download before any upload", r1, r2, r3);
            if (r1 >= 0 || r2 >= 0 || r3 >= 0) ++failures;
            rxr_destroy(fresh);
        } else ++failures;
        int r4 = rxr_create(&fresh, 1000);
        printf("%-44s rc=%d\n", "rxr_create on device 1000", r4);
        if (r4 != RXR_ERR_NO_DEVICE) ++failures;
        int r5 = rxr_rasterize(ctx, NULL, pixels), r6 = rxr_rasterize(ctx, &base, NULL), r7 = rxr_rasterize(NULL, &base, pixels);
        printf("%-44s rc=%d,%d,%d\n", "NULL frame / pixels / context", r5, r6, r7);
        if (r5 >= 0 || r6 >= 0 || r7 >= 0) ++failures;
        make_valid();
        if (rxr_upload_frame(ctx, &base) == RXR_OK) {
            int r8 = rxr_render_rows(ctx, 8, 4), r9 = rxr_render_rows(ctx, 0, H + 1u), r10 = rxr_download_rows(ctx, pixels, 0, H + 1u);
            printf("%-44s rc=%d,%d,%d\n", "row ranges: reversed / beyond the frame", r8, r9, r10);
            if (r8 >= 0 || r9 >= 0 || r10 >= 0) ++failures;
        } else ++failures;
    }
    /* the other entry points that take arrays from the caller */
    {
        static uint8_t texel[4] = {1, 2, 3, 255};
        rxr_texture t_ok = {texel, 1, 1}, t_null = {NULL, 1, 1}, t_zero = {texel, 0, 1}, t_huge = {texel, 40000, 1};
        rxr_tile tile = {&t_ok, 1};
        int r[8];
        tile.textures = &t_null; r[0] = rxr_set_textures(ctx, &tile, 1, NULL, 0);
        tile.textures = &t_zero; r[1] = rxr_set_textures(ctx, &tile, 1, NULL, 0);
        tile.textures = &t_huge; r[2] = rxr_set_textures(ctx, &tile, 1, NULL, 0);
        tile.textures = NULL;    r[3] = rxr_set_textures(ctx, &tile, 1, NULL, 0);
        r[4] = rxr_set_textures(ctx, NULL, 2, NULL, 0);
        r[5] = rxr_set_textures(ctx, NULL, 0, NULL, 3);
        tile.textures = &t_ok;   r[6] = rxr_set_textures(ctx, &tile, 1, NULL, 0);  /* (a good one: accepted) */
        r[7] = rxr_set_textures(ctx, NULL, 0, NULL, 0);                           /* (and none at all: accepted) */
        printf("%-44s rc=%d,%d,%d,%d,%d,%d then %d,%d\n", "rxr_set_textures: bad textures / arrays", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        for (int i = 0; i < 6; ++i) if (r[i] >= 0) ++failures;
        if (r[6] != RXR_OK || r[7] != RXR_OK) ++failures;

        static float mv[3][4] = {{0, 0, 0, 1}, {1, 0, 0, 1}, {0, 1, 0, 1}};
        static float mn[3][3] = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}};
        static uint32_t mi[3] = {0, 1, 2}, mi_bad[3] = {0, 1, 7};
        rxr_mesh3d mesh;
        memset(&mesh, 0, sizeof mesh);
        mesh.vertices = &mv[0][0]; mesh.uvs = &uvs[0][0]; mesh.normals = &mn[0][0]; mesh.indices = mi;
        mesh.n_vertices = 3; mesh.n_triangles = 1;
        identity(mesh.transform_3d);
        mesh.shader = -1; mesh.chunk = -1; mesh.list = RXR_LIST_STATIC; mesh.source.kind = RXR_SOURCE_PIXEL;
        rxr_mesh3d m;
        int q[7];
        m = mesh; m.normals = NULL;          q[0] = rxr_set_meshes(ctx, &m, 1);
        m = mesh; m.indices = mi_bad;        q[1] = rxr_set_meshes(ctx, &m, 1);
        m = mesh; m.cull_mode = 5;           q[2] = rxr_set_meshes(ctx, &m, 1);
        m = mesh; m.vertices = NULL;         q[3] = rxr_set_meshes(ctx, &m, 1);
        m = mesh; m.indices = NULL;          q[4] = rxr_set_meshes(ctx, &m, 1);
        q[5] = rxr_set_meshes(ctx, NULL, 2);
        q[6] = rxr_set_meshes(ctx, &mesh, 1);                                      /* (a good one: accepted) */
        printf("%-44s rc=%d,%d,%d,%d,%d,%d then %d\n", "rxr_set_meshes: bad meshes", q[0], q[1], q[2], q[3], q[4], q[5], q[6]);
        for (int i = 0; i < 6; ++i) if (q[i] >= 0) ++failures;
        if (q[6] != RXR_OK) ++failures;
        if (rxr_set_meshes(ctx, NULL, 0) != RXR_OK) ++failures;

        {
            /* ABI 5: 3D batches without Edges records (edges == NULL, edge_visible + cull_mode instead).  Malformed variants: neither array,
             * a cull mode that does not exist, and the two forms mixed in one frame */
            make_valid();
            static uint32_t vis_words[64];
            for (int i = 0; i < 64; ++i) vis_words[i] = 1u;
            rxr_batch3d two3[2];
            rxr_frame f = base;
            two3[0] = base.batches3d[0];
            two3[1] = base.batches3d[0];
            f.batches3d = two3;
            f.n_batches3d = 2;
            two3[0].edges = NULL; two3[0].edge_visible = NULL;
            two3[1].edges = NULL; two3[1].edge_visible = NULL;
            expect_error(ctx, "batch3d: neither edges nor edge_visible", &f);
            two3[0].edge_visible = vis_words; two3[1].edge_visible = vis_words; two3[0].cull_mode = 7;
            expect_error(ctx, "batch3d: edge_visible with a bad cull mode", &f);
            two3[0].cull_mode = RXR_CULL_OFF; two3[1].edges = base.batches3d[0].edges;
            expect_error(ctx, "batch3d: with and without Edges records", &f);
        }
        {
            /* rxr_set_meshes2d: a registration that fails half way (the SECOND mesh has an index out of range) must leave an EMPTY
             * registration, not the first mesh of the failed call beside the device data of the call before (round-3 advisor finding).  A
             * frame that then asks for the registered 2D meshes has no 2D primitives: it renders its 3D part, or is refused -- never faults. */
            static float v2[4][2] = {{4, 4}, {40, 4}, {40, 30}, {4, 30}}, u2[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
            static uint32_t i2[6] = {0, 1, 2, 0, 2, 3}, i2_bad[6] = {0, 1, 2, 0, 2, 9};
            rxr_mesh2d two[2];
            memset(two, 0, sizeof two);
            for (int k = 0; k < 2; ++k) {
                two[k].vertices = &v2[0][0]; two[k].uvs = &u2[0][0]; two[k].indices = i2;
                two[k].n_vertices = 4; two[k].n_triangles = 2; two[k].mode = RXR_MODE_TRIANGLES;
                two[k].source.kind = RXR_SOURCE_PIXEL; two[k].source.pixel[1] = 200; two[k].source.pixel[3] = 255;
                two[k].shader = -1; two[k].chunk = -1;
            }
            const int good = rxr_set_meshes2d(ctx, two, 2);
            two[1].indices = i2_bad;
            const int bad = rxr_set_meshes2d(ctx, two, 2);
            make_valid();
            rxr_frame f = base;
            f.use_meshes = 2;        /* the 2D batches are the registered meshes */
            f.batches2d = NULL;
            f.n_batches2d = 0;
            memset(pixels, 0, sizeof pixels);
            const int rr = rxr_rasterize(ctx, &f, pixels);
            const uint8_t *p2 = &pixels[(12u * W + 12u) * 4u];
            printf("%-44s rc=%d then %d; the frame after it rc=%d pixel2d=%u,%u,%u\n", "rxr_set_meshes2d: second mesh bad", good, bad, rr, p2[0], p2[1], p2[2]);
            if (good != RXR_OK || bad >= 0) ++failures;
            if (rr == RXR_OK && p2[1] == 200) {   /* a mesh of the FAILED registration was drawn */
                printf("  ^^^ FAILED: the failed registration left meshes behind\n");
                ++failures;
            }
            if (rxr_set_meshes2d(ctx, NULL, 0) != RXR_OK) ++failures;
        }
        rxr_ctx *multi = NULL;
        int ids_bad[2] = {0, 1000}, ids_ok[2] = {0, 0};
        int c0 = rxr_create_multi(&multi, NULL, 2), c1 = rxr_create_multi(&multi, ids_ok, 0), c2 = rxr_create_multi(&multi, ids_bad, 2), c3 = rxr_create_multi(NULL, ids_ok, 2);
        printf("%-44s rc=%d,%d,%d,%d\n", "rxr_create_multi: bad device lists", c0, c1, c2, c3);
        if (c0 >= 0 || c1 >= 0 || c2 >= 0 || c3 >= 0) ++failures;
        if (rxr_create_multi(&multi, ids_ok, 2) == RXR_OK) {   /* two logical members on one GPU render the valid frame too */
            make_valid();
            memset(pixels, 0, sizeof pixels);
            int rm = rxr_rasterize(multi, &base, pixels);
            const uint8_t *p2 = &pixels[(12u * W + 12u) * 4u];
            printf("%-44s rc=%d pixel2d=%u,%u,%u\n", "valid frame on a 2-member context", rm, p2[0], p2[1], p2[2]);
            if (rm != RXR_OK || p2[1] != 220) ++failures;
            rxr_destroy(multi);
        } else ++failures;
    }
    expect_valid(ctx, "valid frame again, same context");
    rxr_destroy(ctx);
    printf(failures ? "FAILED: %d case(s)\n" : "ok (%d failures)\n", failures);
    return failures ? 1 : 0;
}
