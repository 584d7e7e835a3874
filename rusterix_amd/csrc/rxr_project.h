// rxr_project.h -- data layout of the device-side projection path:
// Batch3D::clip_and_project + Edges::new + bounding box on the GPU.  See rxr_project.hip.
#pragma once
#include <stdint.h>

#ifdef RXR_JIT
#include "rxr.h"  // (hiprtc: the sources are in-memory headers with plain names, rxr_jit.hip)
#else
#include "../../include/rxr.h"
#endif

// Output layout per mesh (capacity based, so that triangle ids keep the reference's submission order
// without a cross-mesh compaction): vertices [vout_base, vout_base + n_verts + 4*n_tris),
// triangles [tout_base, tout_base + 3*n_tris).  The first n_verts / n_tris slots are the originals
// (copied by clip_and_project at batch3d.rs:566-574), the rest receives what near-plane clipping
// appends (:627-686) -- at most 4 vertices and 2 fan triangles per clipped triangle.
struct DevMesh {
    uint32_t vin_base, tin_base;    // into the object-space pools
    uint32_t n_verts, n_tris;
    uint32_t vout_base, tout_base;  // into the projected pools (the pools k_setup3d reads)
    uint32_t cull_mode;
    uint32_t rejected;              // per frame: the AABB frustum test dropped the batch (:493-552)
    float view_model[16];           // per frame: view * transform_3d (:555)
};  // 96 B

// bounding boxes are accumulated with integer atomics on an order-preserving encoding of f32
// One box per 128-byte line: atomics on one cache line serialise in L2, and with eight boxes to a line the 1 M-triangle grid
// queued 3 456 of them per line -- the whole run time of k_proj_vertices.
struct alignas(128) DevBBox {
    uint32_t min_x, min_y, max_x, max_y;
};

// prefix entry of the append scan: low 32 bits = vertices emitted before this triangle, high 32 bits =
// fan triangles emitted before it (both counted over ALL meshes; per-mesh offsets subtract the value
// at the mesh's first triangle)
typedef unsigned long long AppendCount;

#define RXR_PROJ_SCAN_CHUNK 8192u

struct ProjectParams {
    uint32_t n_meshes, n_verts_in, n_tris_in;  // totals over all meshes (object space)
    uint32_t n_tris_out;                       // total triangle capacity (sum of 3 * n_tris)
    uint32_t edges_in_setup;                   // 1: k_setup3d builds the Edges records itself (RasterParams.pm_meshes): no k_proj_edges launch
    float projection[16];
    float width, height;

    const DevMesh *meshes;
    const uint32_t *vin_prefix;   // n_meshes + 1: first object-space vertex of each mesh
    const uint32_t *tin_prefix;   // n_meshes + 1
    const uint32_t *tout_prefix;  // n_meshes + 1 (== the raster path's batch_tri_base)

    const float4 *obj_verts;
    const uint32_t *obj_idx;      // 3 per triangle, mesh-local
    const float2 *obj_uvs;
    const float *obj_normals;     // 3 per vertex

    float4 *view_verts;           // view space, indexed like the projected pool
    float4 *pv;                   // projected_vertices pool (output layout)
    float2 *uv;                   // clipped_uvs pool
    float *nrm;                   // clipped_normals pool
    uint32_t *idx;                // clipped_indices pool, 3 per triangle slot, mesh-local
    rxr_edges *edges;             // Edges pool, one per triangle slot
    uint8_t *edge_vis;            // per original triangle: edge_visibility (:582-618)
    AppendCount *append;          // per original triangle: this triangle's (verts, tris) then, after the scan, the exclusive prefix
    AppendCount *chunk_tot;       // scan scratch
    AppendCount *chunk_base;
    uint32_t *ticket;             // [0] scan last-block ticket (cleared by the last block); [1] "some triangle of the frame is clipped" (k_proj_init clears, k_clip_count raises)
    DevBBox *bbox;                // per mesh
    uint32_t *mesh_live;          // per mesh and frame: triangle slots in use = originals + appended fans (0 for a rejected mesh);
                                  // the slots behind them are dead: k_proj_edges, k_setup3d and k_fill skip whole workgroups of them
};

#if defined(__HIPCC__) || defined(RXR_JIT)
// Edges for one USED triangle slot of mesh M (batch3d.rs:706-739 + edge.rs:12-24) from its three projected vertices: the front-facing
// test, the winding swap of the cull mode, Edges::new([v0,v1,v2], [v1,v2,v0]).  `evis`: edge_visibility of the slot (appended fans:
// true, :731-732).  Shared by k_proj_edges and, for frames whose set-up builds the records itself, make_setup (rxr_kernels.hip).
__device__ __forceinline__ rxr_edges edges_from_vertices(uint32_t cull_mode, bool evis, float4 v0, float4 v1, float4 v2) {
    rxr_edges E;
    // is_front_facing, :742-746
    const bool front = ((v1.x - v0.x) * (v2.y - v0.y) - (v1.y - v0.y) * (v2.x - v0.x)) > 0.0f;
    bool visible, swap;
    if (cull_mode == RXR_CULL_OFF) {
        swap = front;
        visible = true;
    } else if (cull_mode == RXR_CULL_FRONT) {
        swap = false;
        visible = !front;
    } else {
        swap = front;
        visible = front;
    }
    if (swap) {
        const float4 tmp = v1;
        v1 = v2;
        v2 = tmp;
    }
    // Edges::new([v0,v1,v2], [v1,v2,v0]): a = y1 - y0, b = x0 - x1, c = x1*y0 - y1*x0
    const float px[3] = {v0.x, v1.x, v2.x}, py[3] = {v0.y, v1.y, v2.y};
    const float qx[3] = {v1.x, v2.x, v0.x}, qy[3] = {v1.y, v2.y, v0.y};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        E.a[i] = qy[i] - py[i];
        E.b[i] = px[i] - qx[i];
        E.c[i] = qx[i] * py[i] - qy[i] * px[i];
    }
    E.visible = (evis && visible) ? 1u : 0u;
    return E;
}
#endif

// ---- the 2D half: Batch2D::project (src/batch/batch2d.rs:373-425) + the Prim2D records rxr_upload_frame builds on the host ----------
struct DevMesh2D {
    uint32_t vin_base, n_verts;     // into the object-space 2D pools
    uint32_t mode;                  // RXR_MODE_*
    uint32_t line_color;            // Lines / LineStrip / LineLoop: the packed colour of every segment (:911-915)
};
struct Prim2DSrc {
    uint32_t mesh;                  // registered 2D mesh = batch index of the frame
    uint32_t ia, ib, ic;            // mesh-local vertex indices: a triangle's three, a segment's two
};
struct Project2DParams {
    uint32_t n_meshes, n_verts, n_prims, has_matrix;
    float m[9];                     // vek column-major Mat3 (m[c * 3 + r])
    float width, height;
    uint32_t ref_tile;              // the reference's tile_size (the box test of risky batches, rxr_device.h rxr_ref_tile_span)
    const DevMesh2D *meshes;
    const uint32_t *vin_prefix;     // n_meshes + 1
    const float2 *obj_verts, *obj_uvs;
    const Prim2DSrc *src;
    DevBBox *bbox;                  // per mesh and frame
    struct Prim2D *out;             // the frame blob's Prim2D records
    uint32_t *d2_box;               // min_x, max_x, min_y, max_y of the non-empty pixel boxes (RasterParams.d2_box_dev)
    uint32_t *bad_line;             // pinned status word: a segment end point beyond +-2^30 (the host builder refuses the frame)
};
