/*
 * rxr.h -- C ABI of the MI355X (gfx950) rasterizer back end for Rusterix' tile rasterizer.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own; the boundary it exposes is the
 * Rust call
 *
 *     Rasterizer::setup(m2d, view, proj) .. .rasterize(&mut scene, pixels, w, h, tile_size, &assets)
 *                                                   (reference src/rasterizer.rs:92-152, 185-193)
 *
 * The host side (Rust in a real deployment, the C++ mirror under rusterix_amd/csrc/host here) keeps
 * scene set-up, Scene::project (src/scene.rs:154-200) and the Edges precompute (src/edge.rs:12-24),
 * flattens what the per-tile loops read into the POD structs below and hands them to the entry
 * points declared at the bottom of this file.  Everything that `rasterize` does after
 * `scene.project(..)` (src/rasterizer.rs:256-579) happens behind this ABI on the GPU.
 *
 * Conventions
 *   - plain C, no torch / C++ types; all pointers are HOST pointers unless the name says `dev`.
 *   - every function returns RXR_OK (0) or a negative rxr_status; nothing throws or aborts.
 *   - matrices are column-major exactly as vek stores them: m[c*4 + r] == cols[c][r].
 *   - `usize` indices of the reference are u32 here (the shim asserts < 2^32).
 *   - structs carry no implicit ownership: the caller keeps every buffer alive until the call
 *     returns; the library copies what it needs.
 *   - threads: a context (and a multi-device handle with its members) is used by one thread at a time -- the one exception is
 *     rxr_stream_batch3d, which any number of threads may call on the same context between rxr_stream_begin and
 *     rxr_upload_frame.  Different contexts are independent: threads that each own their contexts need no lock between them, on
 *     one GPU or several (what the library shares process-wide, the run-time compiler's cache and job table, it guards itself;
 *     tests/abi_threads.c).
 */
#ifndef RXR_H
#define RXR_H

#if !defined(__HIPCC_RTC__) /* (hiprtc, the run-time compiler of rxr_jit.hip, has size_t built in and no <stddef.h>) */
#include <stddef.h>
#endif
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RXR_ABI_VERSION 5u

typedef enum rxr_status {
    RXR_OK = 0,
    RXR_ERR_INVALID = -1,     /* bad argument / index out of range (the reference would panic)   */
    RXR_ERR_NO_DEVICE = -2,   /* no HIP device, or device id out of range                         */
    RXR_ERR_HIP = -3,         /* a HIP runtime call failed; see rxr_last_error                    */
    RXR_ERR_UNSUPPORTED = -4, /* scene uses a feature the device path does not implement (yet)    */
    RXR_ERR_OOM = -5,
    RXR_ERR_OVERFLOW = -6     /* rxr_synchronize: a bin list overflowed in a launch that was queued BEFORE the last one (an
                                 asynchronous caller that does not synchronize per frame): that frame was shipped incomplete.
                                 The lists have been grown and the last launch rendered again; render the lost frames again. */
} rxr_status;

/* ---- enums mirrored from the reference ------------------------------------------------------ */

/* SampleMode, src/texture.rs:5-12 */
enum { RXR_SAMPLE_NEAREST = 0, RXR_SAMPLE_LINEAR = 1 };
/* RepeatMode, src/texture.rs:14-25 */
enum { RXR_REPEAT_CLAMP_XY = 0, RXR_REPEAT_REPEAT_XY = 1, RXR_REPEAT_REPEAT_X = 2, RXR_REPEAT_REPEAT_Y = 3 };
/* PrimitiveMode, src/batch/mod.rs:4-15 */
enum { RXR_MODE_TRIANGLES = 0, RXR_MODE_LINES = 1, RXR_MODE_LINE_STRIP = 2, RXR_MODE_LINE_LOOP = 3 };
/* LightType, src/map/light.rs:6-14 */
enum { RXR_LIGHT_POINT = 0, RXR_LIGHT_AMBIENT = 1, RXR_LIGHT_AMBIENT_DAYLIGHT = 2, RXR_LIGHT_SPOT = 3,
       RXR_LIGHT_AREA = 4, RXR_LIGHT_DAYLIGHT = 5 };
/* PixelSource, src/map/pixelsource.rs:23-37.  Only the variants the raster loops distinguish
 * (src/rasterizer.rs:672-758, 1101-1222); every other variant is RXR_SOURCE_OTHER. */
enum { RXR_SOURCE_OTHER = 0,         /* Off / TileId / MaterialId / Sequence / Color / ShapeFXGraphId:
                                        3D texel [0,0,0,255], 2D texel [0,0,0,0]                   */
       RXR_SOURCE_STATIC_TILE = 1,   /* StaticTileIndex(u16) -> assets.tile_list[index]            */
       RXR_SOURCE_DYNAMIC_TILE = 2,  /* DynamicTileIndex(u16) -> scene.dynamic_textures[index]     */
       RXR_SOURCE_PIXEL = 3,         /* Pixel([u8;4])                                              */
       RXR_SOURCE_TERRAIN = 4,       /* Terrain (chunk terrain texture)                            */
       RXR_SOURCE_MISSING = 5 };     /* EntityTile/ItemTile whose lookup failed on the host: [0,0,0,0];
                                        a successful lookup is passed as RXR_SOURCE_DYNAMIC_TILE     */
/* Host-side only -- these two never cross the ABI (rxr_upload_frame answers them with RXR_ERR_INVALID): EntityTile(id, index) /
 * ItemTile(id, index), src/map/pixelsource.rs:29-30.  The raster loops look (id, index) up per fragment in
 * assets.entity_tiles / assets.item_tiles (FxHashMap<u32, IndexMap<String, Tile>>; src/rasterizer.rs:1140-1187, :705-748,
 * :1548-1595); the lookup is frame-constant, so the HOST does it once per batch and hands the device
 * RXR_SOURCE_DYNAMIC_TILE (hit: the tile travels with the dynamic tiles of rxr_set_textures) or RXR_SOURCE_MISSING (unknown
 * id, or no `index`-th sequence: [0,0,0,0] in all three loops).  The host mirror and the oracle use these values for the
 * unresolved variants. */
enum { RXR_HOST_SOURCE_ENTITY_TILE = 64, RXR_HOST_SOURCE_ITEM_TILE = 65 };
/* which Scene list a 3D batch came from; order of the array is submission order
 * (src/rasterizer.rs:314-405) */
enum { RXR_LIST_CHUNK_OPACITY = 0, RXR_LIST_CHUNK = 1, RXR_LIST_CHUNK_TERRAIN = 2, RXR_LIST_STATIC = 3,
       RXR_LIST_DYNAMIC = 4, RXR_LIST_OVERLAY = 5 };
/* background shader kinds (trait Shader, src/shader/mod.rs:9-33) that the device evaluates itself */
enum { RXR_BG_NONE = 0, RXR_BG_VGRADIENT = 1 /* src/shader/vgradient.rs:11-15 */,
       RXR_BG_HOST_PIXELS = 2 /* any other `dyn Shader`, evaluated by the host into background_pixels */,
       RXR_BG_GRID = 3 /* GridShader, src/shader/grid.rs:10-109; parameters in rxr_frame.background_grid */ };

/* rxr_frame.flags */
#define RXR_FLAG_D2_ACTIVE             (1u << 0) /* RenderMode.d2_active, src/rendermode.rs:4-11   */
#define RXR_FLAG_D3_ACTIVE             (1u << 1) /* RenderMode.d3_active                            */
#define RXR_FLAG_IGNORE_BG_SHADER      (1u << 2) /* RenderMode.ignore_background_shader             */
#define RXR_FLAG_PRESERVE_TRANSPARENCY (1u << 3) /* Rasterizer.preserve_transparency, :72           */
#define RXR_FLAG_HAS_BACKGROUND_COLOR  (1u << 4) /* Rasterizer.background_color.is_some(), :59      */
#define RXR_FLAG_HAS_AMBIENT           (1u << 5) /* Rasterizer.ambient_color.is_some(), :62         */
#define RXR_FLAG_HAS_SUN               (1u << 6) /* Rasterizer.sun_dir.is_some(), :86               */

/* ---- PODs ------------------------------------------------------------------------------------ */

/* one Texture (src/texture.rs:46-54): RGBA8 row-major, data[(y*width + x)*4] */
typedef struct rxr_texture {
    const uint8_t *rgba;
    uint32_t width, height;
} rxr_texture;

/* one Tile (src/map/tile.rs `textures: Vec<Texture>`): the animation frames of a tile; the raster
 * loops pick textures[animation_frame % len] (src/rasterizer.rs:1104-1105) */
typedef struct rxr_tile {
    const rxr_texture *textures;
    uint32_t n_textures;
} rxr_tile;

/* CompiledLight, src/map/light.rs:456-477, field for field */
typedef struct rxr_light {
    uint32_t light_type;
    float position[3];
    float color[3];
    float intensity;
    uint32_t emitting;
    float start_distance, end_distance, flicker;
    float direction[3];
    float cone_angle;
    float normal[3];
    float width, height;
    uint32_t from_linedef;
} rxr_light;

/* Edges, src/edge.rs:2-8.  `Edges` is repr(Rust) with private a/b/c; the shim needs the accessor
 * shown in INTEGRATION.md to fill this. */
typedef struct rxr_edges {
    float a[3], b[3], c[3];
    uint32_t visible;
} rxr_edges;

typedef struct rxr_source {
    uint32_t kind;    /* RXR_SOURCE_*            */
    uint32_t index;   /* tile index for *_TILE   */
    uint8_t pixel[4]; /* colour for RXR_SOURCE_PIXEL */
} rxr_source;

/* Batch3D after clip_and_project (src/batch/batch3d.rs:15-78, outputs of :482-740) */
typedef struct rxr_batch3d {
    const float *projected_vertices;  /* [n_vertices][4] = (screen x, screen y, ndc z, clip w), :694-699 */
    const float *clipped_uvs;         /* [n_vertices][2]                                           */
    const float *clipped_normals;     /* [n_vertices][3]; NULL iff batch.normals.is_empty() (:1083) */
    const uint32_t *clipped_indices;  /* [n_triangles][3]                                          */
    const rxr_edges *edges;           /* [n_triangles]                                             */
    uint32_t n_vertices, n_triangles;
    uint32_t has_bounding_box;        /* bounding_box.is_some(); None => batch skipped (:978)      */
    float bounding_box[4];            /* Rect x, y, width, height (src/rect.rs:5-10)               */
    uint32_t repeat_mode;
    rxr_source source;
    float ambient_color[3];
    int32_t shader;                   /* Option<usize>: -1 = None                                  */
    uint32_t has_profile_id, profile_id;
    uint32_t list;                    /* RXR_LIST_*                                                */
    int32_t chunk;                    /* index into rxr_frame.chunks, -1 for the scene-level lists */
    /* ABI 5, optional: edges == NULL.  The Edges records are the largest array of the hand-over (40 of the 124 bytes per triangle that a
     * frame of boxes sends over PCIe) and a function of arrays that travel anyway: the device then builds them itself from the projected
     * vertices -- Edges::new([v0,v1,v2],[v1,v2,v0]) behind the winding swap of `cull_mode` (src/edge.rs:12-24,
     * src/batch/batch3d.rs:706-746; the same operations in the same order: the same floats) -- and the host sends one word per
     * triangle: edge_visible[t] != 0 iff the record's `visible` (what the host would have passed to Edges::new).  Every 3D batch of a
     * frame takes the same form (all with `edges` or all without); rxr_stream_batch3d likewise. */
    const uint32_t *edge_visible;     /* [n_triangles]; read only when edges == NULL               */
    uint32_t cull_mode;               /* RXR_CULL_* of the batch; read only when edges == NULL     */
} rxr_batch3d;

/* CullMode, src/batch/mod.rs:17-26 */
enum { RXR_CULL_OFF = 0, RXR_CULL_FRONT = 1, RXR_CULL_BACK = 2 };

/* Batch3D BEFORE projection: the inputs of Batch3D::clip_and_project (src/batch/batch3d.rs:482-740).
 * Used by the device-side projection path (rxr_set_meshes + rxr_frame.use_meshes): the geometry is
 * uploaded once and every frame sends only matrices; the device then performs the view transform,
 * the near-plane clip (z < -0.1) with its appended fan triangles, the screen mapping (:689-700),
 * Edges::new with the cull-mode rule (:706-739, src/edge.rs:12-24) and the bounding box (:749-768). */
typedef struct rxr_mesh3d {
    const float *vertices;            /* [n_vertices][4]                                            */
    const uint32_t *indices;          /* [n_triangles][3]                                           */
    const float *uvs;                 /* [n_vertices][2]                                            */
    const float *normals;             /* [n_vertices][3]; required when n_triangles > 0 (the
                                         reference indexes self.normals unconditionally, :605-607)   */
    uint32_t n_vertices, n_triangles;
    float transform_3d[16];           /* Batch3D.transform_3d, column-major                         */
    uint32_t cull_mode;               /* RXR_CULL_*                                                 */
    uint32_t repeat_mode;
    rxr_source source;
    float ambient_color[3];
    int32_t shader;
    uint32_t has_profile_id, profile_id;
    uint32_t list;                    /* RXR_LIST_*                                                 */
    int32_t chunk;
} rxr_mesh3d;

/* Batch2D after project (src/batch/batch2d.rs:10-52, outputs of :373-425) */
typedef struct rxr_batch2d {
    const float *projected_vertices;  /* [n_vertices][2] */
    const float *uvs;                 /* [n_vertices][2] */
    const uint32_t *indices;          /* [n_triangles][3]; for Lines only .0/.1 are used (:902)     */
    const rxr_edges *edges;           /* [n_triangles], or NULL (ABI 5): Edges::new([v0,v1,v2],[v1,v2,v0], true) of the projected
                                         vertices, as Batch2D::project builds them (src/batch/batch2d.rs:413-424), is then built by
                                         the library                                                */
    uint32_t n_vertices, n_triangles;
    uint32_t has_bounding_box;
    float bounding_box[4];
    uint32_t mode;                    /* RXR_MODE_* */
    uint32_t repeat_mode;
    rxr_source source;
    uint32_t receives_light;
    int32_t shader;
    int32_t chunk;
} rxr_batch2d;

/* Batch2D BEFORE projection: the inputs of Batch2D::project (src/batch/batch2d.rs:373-425) -- the 2D half of the device-side
 * projection path (rxr_set_meshes2d + rxr_frame.use_meshes bit 1): the geometry is uploaded once, every frame sends the optional
 * Mat3 (rxr_set_projection2d), and the device applies it, accumulates the bounding box (f32::min / f32::max: NaN dropped),
 * builds the Edges (src/edge.rs:12-24) and the clamped pixel boxes (src/rasterizer.rs:615-634, :1777-1821). */
typedef struct rxr_mesh2d {
    const float *vertices;            /* [n_vertices][2]                                            */
    const uint32_t *indices;          /* [n_triangles][3]; Lines read .0/.1 only (:902)             */
    const float *uvs;                 /* [n_vertices][2]                                            */
    uint32_t n_vertices, n_triangles;
    uint32_t mode;                    /* RXR_MODE_* */
    uint32_t repeat_mode;
    rxr_source source;
    uint32_t receives_light;
    int32_t shader;
    int32_t chunk;
} rxr_mesh2d;

/* (BBox, occlusion) entry of MapMini.occluded_sectors / Chunk.occluded_sectors
 * (src/map/mini.rs:58-66, src/chunk.rs:154-161, src/map/bbox.rs:35-40) */
typedef struct rxr_occluder {
    float min[2], max[2];
    float occlusion;
} rxr_occluder;

/* CompiledLinedef start/end as used by MapMini::is_visible (src/map/mini.rs:68-95) */
typedef struct rxr_linedef {
    float start[2], end[2];
} rxr_linedef;

/* the per-chunk data the raster loops read (src/chunk.rs); chunks in the host's iteration order */
typedef struct rxr_chunk {
    const rxr_occluder *occluders;    /* chunk.occluded_sectors (src/chunk.rs:42)                  */
    uint32_t n_occluders;
    /* chunk.shaders (src/chunk.rs:51): programs[program_base .. program_base + n_programs) of the set given to
     * rxr_set_shaders; a chunk batch's `shader` indexes this range (src/rasterizer.rs:763, :1285, :1645) */
    uint32_t program_base, n_programs;
    /* chunk.shader_textures (src/chunk.rs:53): the baked texture of shader i replaces the texel of a 3D
     * opaque-pass batch whose shader == i, and the program does not run (src/rasterizer.rs:1226-1267);
     * rgba == NULL for None */
    const rxr_texture *shader_textures;
    uint32_t n_shader_textures;
    /* chunk.terrain_texture (NULL = None), chunk.origin, chunk.size: what PixelSource::Terrain batches of
     * this chunk sample by world position (src/chunk.rs:133-151) */
    const rxr_texture *terrain_texture;
    int32_t origin[2];
    int32_t size;
} rxr_chunk;

/* everything `rasterize` reads from `self` and `scene` after projection */
typedef struct rxr_frame {
    uint32_t abi_version;             /* RXR_ABI_VERSION                                           */
    uint32_t width, height;           /* framebuffer size in pixels                                */
    uint32_t tile_size;               /* the caller's tile_size; accepted for API parity only: the
                                         result does not depend on it   */
    float inverse_view[16];           /* Rasterizer.inverse_view_matrix, :43, :97                  */
    float inverse_projection[16];     /* Rasterizer.inverse_projection_matrix, :44, :116           */
    float camera_pos[3];              /* :98-102                                                   */
    float translationd2[2];           /* :104-110                                                  */
    float scaled2;
    uint32_t hash_anim;               /* hash_u32(animation_frame), :199-208                       */
    uint64_t animation_frame;         /* scene.animation_frame (usize)                             */
    uint32_t flags;                   /* RXR_FLAG_*                                                */
    uint8_t background_color[4];
    float ambient[4];
    float sun_dir[3];
    float day_factor;
    uint32_t sample_mode;             /* RXR_SAMPLE_*; rasterizer-wide (:1119)                     */
    float time;
    uint32_t background_kind;         /* RXR_BG_*; only read when D3 is off or pixel is... see :292 */
    const uint8_t *background_pixels; /* RXR_BG_HOST_PIXELS: width*height*4                        */

    const rxr_batch3d *batches3d;     /* submission order, src/rasterizer.rs:314-405               */
    uint32_t n_batches3d;
    const rxr_batch2d *batches2d;     /* submission order, :503-552                                */
    uint32_t n_batches2d;
    const rxr_light *lights;          /* scene.lights followed by scene.dynamic_lights (:1373)     */
    uint32_t n_lights;
    const rxr_occluder *occluders;    /* Rasterizer.mapmini.occluded_sectors                       */
    uint32_t n_occluders;
    const rxr_linedef *linedefs;      /* Rasterizer.mapmini.linedefs                               */
    uint32_t n_linedefs;
    const rxr_chunk *chunks;
    uint32_t n_chunks;
    uint32_t n_shader_programs;       /* scene.shaders.len(); a batch whose shader index resolves
                                         to a program makes the call return RXR_ERR_UNSUPPORTED     */
    /* device-side projection: when use_meshes != 0 the 3D batches are the meshes
     * registered with rxr_set_meshes (batches3d must then be empty) and are projected on the device
     * with these matrices, replacing Scene::project's 3D half (src/scene.rs:189-199).
     * Bit 1 (value 2, alone or with bit 0): the 2D batches are the meshes registered with rxr_set_meshes2d (batches2d must then be
     * empty), projected on the device with the matrix of rxr_set_projection2d: Scene::project's 2D half (:163-187) */
    uint32_t use_meshes;
    float view[16];                   /* Rasterizer.view_matrix, :40                               */
    float projection[16];             /* Rasterizer.projection_matrix, :41                         */
    const float *mesh_transforms;     /* optional [n_meshes][16]: this frame's Batch3D.transform_3d of
                                         every registered mesh (moving objects need no re-registration);
                                         NULL = the transforms given to rxr_set_meshes                */
    /* ABI 3 */
    float background_grid[4];         /* RXR_BG_GRID: GridShader.grid_size, .subdivisions, .offset.x, .offset.y
                                         (src/shader/grid.rs:4-8; defaults 30, 2, (0, 0), :12-16)      */
    uint32_t has_brush_preview;       /* Rasterizer.brush_preview.is_some() (src/rasterizer.rs:13-17, :65): the editor's terrain
                                         brush, blended over the pixels no 3D fragment reached (:435-458) and into the terrain
                                         texels of chunk batches (:1192-1213, :1601-1622)               */
    float brush_position[3];
    float brush_radius, brush_falloff;
} rxr_frame;

/* ---- Rusteria shader programs -------------------------------------
 * A program is handed over as the reference's NodeOp tree (rusteria/src/node/nodeop.rs:12-103),
 * serialised depth-first into 32-bit words; opcodes are the NodeOp variants in declaration order:
 *   node := opcode [payload]
 *   LoadGlobal / StoreGlobal / LoadLocal / StoreLocal : index
 *   GetComponents / SetComponents                     : n, then n component indices (one per word)
 *   If                                                : then_len, has_else, else_len, then-block, else-block
 *   For                                               : init_len, cond_len, incr_len, body_len, the four blocks
 *   Push                                              : the f32 bits of x, y, z
 *   FunctionCall                                      : arity, total_locals, function index
 *   every other variant                               : no payload
 * (block lengths in words).  The library flattens the tree into jump code for the device. */
enum {
    RXR_NODE_LOAD_GLOBAL = 0, RXR_NODE_STORE_GLOBAL, RXR_NODE_LOAD_LOCAL, RXR_NODE_STORE_LOCAL, RXR_NODE_SWAP,
    RXR_NODE_GET_COMPONENTS, RXR_NODE_SET_COMPONENTS, RXR_NODE_IF, RXR_NODE_FOR, RXR_NODE_PUSH, RXR_NODE_FUNCTION_CALL,
    RXR_NODE_RETURN, RXR_NODE_DUP, RXR_NODE_CLEAR, RXR_NODE_PACK2, RXR_NODE_PACK3, RXR_NODE_ADD, RXR_NODE_SUB, RXR_NODE_MUL,
    RXR_NODE_DIV, RXR_NODE_LENGTH, RXR_NODE_LENGTH2, RXR_NODE_LENGTH3, RXR_NODE_ABS, RXR_NODE_SIN, RXR_NODE_SIN1,
    RXR_NODE_SIN2, RXR_NODE_COS, RXR_NODE_COS1, RXR_NODE_COS2, RXR_NODE_TAN, RXR_NODE_ATAN, RXR_NODE_ATAN2,
    RXR_NODE_ROTATE2D, RXR_NODE_DOT, RXR_NODE_DOT2, RXR_NODE_DOT3, RXR_NODE_CROSS, RXR_NODE_NORMALIZE, RXR_NODE_FLOOR,
    RXR_NODE_CEIL, RXR_NODE_ROUND, RXR_NODE_FRACT, RXR_NODE_MOD, RXR_NODE_DEGREES, RXR_NODE_RADIANS, RXR_NODE_MIN,
    RXR_NODE_MAX, RXR_NODE_MIX, RXR_NODE_SMOOTHSTEP, RXR_NODE_STEP, RXR_NODE_CLAMP, RXR_NODE_SQRT, RXR_NODE_POW,
    RXR_NODE_LOG, RXR_NODE_PRINT, RXR_NODE_EQ, RXR_NODE_NE, RXR_NODE_LT, RXR_NODE_LE, RXR_NODE_GT, RXR_NODE_GE,
    RXR_NODE_AND, RXR_NODE_OR, RXR_NODE_NOT, RXR_NODE_NEG, RXR_NODE_UV, RXR_NODE_SET_UV, RXR_NODE_NORMAL,
    RXR_NODE_SET_NORMAL, RXR_NODE_HITPOINT, RXR_NODE_TIME, RXR_NODE_SAMPLE, RXR_NODE_SAMPLE_NORMAL, RXR_NODE_COLOR,
    RXR_NODE_SET_COLOR, RXR_NODE_ROUGHNESS, RXR_NODE_SET_ROUGHNESS, RXR_NODE_METALLIC, RXR_NODE_SET_METALLIC,
    RXR_NODE_EMISSIVE, RXR_NODE_SET_EMISSIVE, RXR_NODE_OPACITY, RXR_NODE_SET_OPACITY, RXR_NODE_BUMP, RXR_NODE_SET_BUMP,
    RXR_NODE_ALLOC, RXR_NODE_ITERATE, RXR_NODE_SAVE, RXR_NODE_PALETTE_INDEX, RXR_NODE_COUNT
};

/* one user function: Arc<[NodeOp]> (rusteria/src/node/program.rs:15) */
typedef struct rxr_function {
    const uint32_t *words;
    uint32_t n_words;
} rxr_function;

/* rusteria::Program (rusteria/src/node/program.rs:7-29); `body` and `strings` are not read by the raster path */
typedef struct rxr_program {
    uint32_t n_globals;               /* Program.globals                                            */
    int32_t shade_index;              /* Program.shade_index; -1 = None (the batch then shades as if it had no shader) */
    uint32_t shade_locals;            /* Program.shade_locals                                       */
    const rxr_function *functions;    /* Program.user_functions                                     */
    uint32_t n_functions;
} rxr_program;

/* TexStorage of the global pattern bank (rusteria/src/textures/mod.rs:10-15): width*height RGB f32 */
typedef struct rxr_pattern {
    const float *rgb;
    uint32_t width, height;
} rxr_pattern;

typedef struct rxr_shader_set {
    const rxr_program *programs;      /* scene.shaders (src/scene.rs:43), indexed by rxr_batch*.shader */
    uint32_t n_programs;
    const rxr_pattern *patterns;      /* rusteria::textures::patterns::patterns(), read by NodeOp::Sample */
    uint32_t n_patterns;
    const rxr_pattern *normal_patterns; /* patterns_normal(), read by NodeOp::SampleNormal          */
    uint32_t n_normal_patterns;
    const float *palette_rgb;         /* assets.palette.colors as [n][3] (TheColor::to_vec3)        */
    const uint8_t *palette_present;   /* [n]: 0 where the palette slot is None                      */
    uint32_t n_palette;
} rxr_shader_set;

/* the last rendered frame.  The *_us fields (microseconds) are only measured while profiling is on (rxr_profile_begin with n > 0)
 * and are 0 otherwise: the sum of the set-up kernels' durations and the raster kernel's, each taken from a start / stop HIP event pair
 * bound to the dispatch itself (see rxr_profile_begin). */
typedef struct rxr_stats {
    float setup_us;      /* triangle set-up + binning kernels */
    float raster_us;     /* the tile raster / shade kernel    */
    float total_us;      /* first kernel start .. last kernel end */
    uint32_t n_triangles3d, n_triangles2d, n_bin_entries;
    uint32_t tiles_x, tiles_y;
} rxr_stats;

typedef struct rxr_ctx rxr_ctx;

/* ---- entry points ----------------------------------------------------------------------------
 * Each replaces a slice of Rasterizer::rasterize (src/rasterizer.rs:185-580):                     */

/* creates a context on HIP device `device_id` (no reference counterpart -- the reference uses rayon's global pool,
 * src/rasterizer.rs:273-275) */
int rxr_create(rxr_ctx **out, int device_id);
void rxr_destroy(rxr_ctx *ctx);

/* ABI 4.  A multi-device context: one member context per entry of device_ids (a device may be listed more than once: N
 * logical members on one GPU, which is how the single-GPU tests check byte identity), driven from this one process.  The
 * reference renders tiles independently and concatenates them at the end (src/rasterizer.rs:273-275, :559-579); this does
 * the same across GPUs.  On such a handle
 *   - rxr_set_textures / rxr_set_meshes / rxr_set_shaders / rxr_upload_frame replicate to every member (in parallel, one
 *     host thread and one PCIe link per device);
 *   - rxr_rasterize / rxr_render_download shard the frame by interleaved RXR_STRIPE_ROWS-row stripes (member i of N renders
 *     stripes i, i+N, ...) and EVERY device copies its own stripes straight into the caller's `pixels`;
 *   - rxr_render_gather assembles the frame in the memory of one member's device instead (peer copies over xGMI);
 *   - rxr_synchronize / rxr_get_stats / rxr_last_error cover all members; rxr_profile_* and rxr_selftest_math refer to
 *     member 0; the single-device calls that take device pointers, streams or row ranges (rxr_render_rows*,
 *     rxr_render_stripes_to, rxr_download_rows) return RXR_ERR_UNSUPPORTED -- use them on rxr_member(ctx, i).
 * The result is byte-identical to the single-device frame (tests/test_gpu_multi.py). */
int rxr_create_multi(rxr_ctx **out, const int *device_ids, int n_devices);
/* 1 for a plain context, the number of members for a multi-device one */
int rxr_member_count(const rxr_ctx *ctx);
/* member `index` of a multi-device context (owned by it; never destroy it), or `ctx` itself for index 0 of a plain one */
rxr_ctx *rxr_member(rxr_ctx *ctx, int index);

/* optional: page-locks the caller's pixel buffer (hipHostRegister, visible to every device) so that the downloads of
 * rxr_rasterize / rxr_render_download are direct DMA.  Unpinned buffers work too (the runtime locks them on the fly per copy).
 * Lock WHOLE PAGES of a mapping the buffer has to itself (an allocation of its own from mmap / posix_memalign(4096, ..) / a Vec with
 * page alignment; or take the memory from rxr_alloc_pinned and lock nothing): pages in the middle of the malloc heap are shared with
 * other allocations and reused by them after the buffer is freed, and a copy into such a page shortly after the unlock has been seen to
 * end in a GPU memory access fault on this runtime.  Unlock before the memory is freed. */
int rxr_pin_host_buffer(rxr_ctx *ctx, void *ptr, size_t bytes);
int rxr_unpin_host_buffer(rxr_ctx *ctx, void *ptr);
/* last error text for this context (or for rxr_create when ctx == NULL) */
const char *rxr_last_error(const rxr_ctx *ctx);
/* number of visible HIP devices, 0 if none (never fails) */
int rxr_device_count(void);

/* uploads assets.tile_list (static) and scene.dynamic_textures (dynamic); replaces the texture
 * reads at src/rasterizer.rs:1103-1137 / :674-704.  Call again only when the textures change. */
int rxr_set_textures(rxr_ctx *ctx, const rxr_tile *static_tiles, uint32_t n_static,
                     const rxr_tile *dynamic_tiles, uint32_t n_dynamic);

/* registers the object-space 3D batches of a scene for device-side projection (submission order).
 * Replaces nothing per frame: it is the one-time hand-over of what Batch3D::clip_and_project reads
 * from `self` (src/batch/batch3d.rs:482-740).  Call again only when geometry or materials change. */
int rxr_set_meshes(rxr_ctx *ctx, const rxr_mesh3d *meshes, uint32_t n_meshes);

/* The 2D half of the same: registers the object-space 2D batches of a scene (submission order, src/rasterizer.rs:503-552) for
 * device-side projection; what Batch2D::project reads from `self` (src/batch/batch2d.rs:373-425).  Call again only when geometry
 * or materials change.  A batch whose texture tile does not exist is refused HERE (the reference panics only when the batch is on
 * screen: the device does not know that before it has projected it). */
int rxr_set_meshes2d(rxr_ctx *ctx, const rxr_mesh2d *meshes, uint32_t n_meshes);
/* the `matrix: Option<Mat3<f32>>` of Batch2D::project for the uploads that follow: nine floats in vek's column-major order
 * (m[c * 3 + r]), or NULL for None (vertices are used as they are) */
int rxr_set_projection2d(rxr_ctx *ctx, const float *mat3);

/* debugging / tests: copies the device-projected arrays of mesh `index` (as produced for the last
 * rendered frame) back to the host in the layout of rxr_batch3d.  Any pointer may be NULL.
 * counts[0] = vertices (originals + appended), counts[1] = triangles (originals + appended). */
int rxr_read_projected_mesh(rxr_ctx *ctx, uint32_t index, uint32_t counts[2], float *projected_vertices, float *clipped_uvs,
                            float *clipped_normals, uint32_t *clipped_indices, rxr_edges *edges, float bounding_box[5],
                            uint32_t capacity_vertices, uint32_t capacity_triangles);

/* replaces the context's shader programs, pattern bank and palette (stay resident until the next call;
 * NULL or an empty set removes them).  Programs are restricted to what a per-fragment evaluation can
 * reproduce: RXR_ERR_UNSUPPORTED for Alloc / Iterate / Save, for SetEmissive (the
 * reference leaks it into every later fragment of the tile), for a local of `shade` or a global that is
 * read before the same invocation wrote it (the reference resizes, never clears them) and for Return inside
 * For (the reference unwinds that incorrectly).
 * Replaces: Execution::shade per fragment, src/rasterizer.rs:760-800, :1283-1304, :1642-1667. */
int rxr_set_shaders(rxr_ctx *ctx, const rxr_shader_set *set);
/* the validation half of rxr_set_shaders without a device (no context needed): same status codes; on failure `message`
 * receives the reason; code_words (optional) the length of the flattened jump code.  Lets a host decide up front whether
 * a scene's programs can run on the device or must take the CPU path. */
int rxr_check_shaders(const rxr_shader_set *set, uint32_t *code_words, char *message, uint32_t message_capacity);

/* Arithmetic of the 3D direct-light loop (src/rasterizer.rs:1373-1391 with CompiledLight::radiance_at, src/map/light.rs:491-552,
 * and shade_fast_brdf, :1875-1951), applied from the next rxr_upload_frame on:
 *   RXR_LIGHT_MATH_RELAXED (the default)  in frames with a 3D light loop, point lights are one fused term (fused multiply-adds,
 *       v_rsq_f32 normalisations, the smoothstep divided by a reciprocal), view direction, normal and -- without occluders -- the
 *       world position go through reciprocals, the encode through v_sqrt_f32: every operand within a few ulp of the reference's
 *       correctly rounded value.  Every quantity
 *       of that path is continuous in the fragment's position (range test, smoothstep, Lambert and specular cut-offs all meet
 *       zero), so a channel moves by at most one 8-bit step -- the tolerance BASELINE.json states for lit 3D fragments; measured:
 *       112 of 8 294 400 pixels of the bench frame differ from the oracle, each by 1.  Spot / area / daylight lights (hard
 *       cut-offs), unlit frames, 2D frames and frames that run Rusteria programs are computed exactly in both modes.
 *   RXR_LIGHT_MATH_EXACT  the reference's operations, correctly rounded, throughout (bit-identical to the CPU oracle up to the
 *       libm-dependent pow; 35 % slower on the bench frame).
 * The environment variable RXR_LIGHT_MATH=exact|relaxed, when set, overrides the context's mode (read at every upload).
 * RXR_ERR_INVALID for another mode.  On a multi-device context: every member.  Nothing in the reference corresponds to it. */
#define RXR_LIGHT_MATH_EXACT 0
#define RXR_LIGHT_MATH_RELAXED 1
int rxr_set_light_math(rxr_ctx *ctx, int mode);

/* Streaming hand-over for large scenes (an addition to ABI 4; optional -- rxr_upload_frame alone does the same, later).
 * The reference projects a scene's batches in parallel (Scene::project: rayon par_iter_mut, src/scene.rs:154-200) and only then
 * walks the tiles; with the tiles on the GPU, projecting a million triangles and THEN copying 124 MB to the device takes the host
 * longer than the GPU needs for ten frames.  A host that projects batch by batch hands each 3D batch over as soon as its
 * clip_and_project is done, from whatever thread did it:
 *     rxr_stream_begin(ctx, n, vertex_capacity, triangle_capacity)   n = the frame's number of 3D batches (rxr_frame.batches3d, same
 *                             order); capacity[i] = an upper bound of what projecting batch i can produce (vertices: n + 4 m,
 *                             triangles: 3 m for a batch of n vertices and m triangles covers the near-plane clip, batch3d.rs:627-686).
 *                             Waits for the previous frame's renders.
 *     rxr_stream_batch3d(ctx, i, &projected)      thread-safe, any order; only the five arrays and the two counts are read.  The arrays
 *                             must stay unchanged until rxr_upload_frame has returned.  Batches are retired in index order (a batch's
 *                             place in the pools -- hence the submission order of its triangles, which breaks depth ties -- is the sum
 *                             of its predecessors' sizes), copied into pinned memory and sent to the device in groups, while the
 *                             host projects the batches behind them.
 *     rxr_upload_frame(ctx, &frame)               as always, with frame.batches3d[i] naming the SAME arrays and counts: it then only
 *                             adds what surrounds them.  If anything differs (a batch missing, other pointers, a capacity exceeded)
 *                             the frame is handed over from scratch as if nothing had been streamed: correct either way.
 * Measured (1 M triangles, 7680x4320, 64 host threads, profiles/r03/c5_e2e_breakdown.jsonl): the call Rasterizer::rasterize 14.1 ms
 * (one copy and one transfer after the projection) -> 10.6 (rxr_upload_frame's own pipelined copy) -> 8.7 (streamed, ordinary
 * arrays) -> 6.4 ms (streamed, page-locked arrays: rxr_stream_begin_pinned below).
 * Multi-device handles: RXR_ERR_UNSUPPORTED (use rxr_upload_frame). */
int rxr_stream_begin(rxr_ctx *ctx, uint32_t n_batches3d, const uint32_t *vertex_capacity, const uint32_t *triangle_capacity);
/* The same with a promise: EVERY array handed to rxr_stream_batch3d lies in page-locked host memory that the device can read
 * (rxr_alloc_pinned below, hipHostMalloc, or memory registered with rxr_pin_host_buffer / hipHostRegister -- each array inside ONE
 * registration; the device reads through the device address the runtime reports for it, which for registered memory need not be
 * the host address).  The library then
 * copies nothing on the host: a kernel on the device pulls each group of batches straight out of the caller's arrays over PCIe
 * into the pools.  A pointer that breaks the promise is a device page fault -- only promise what an allocator guarantees. */
int rxr_stream_begin_pinned(rxr_ctx *ctx, uint32_t n_batches3d, const uint32_t *vertex_capacity, const uint32_t *triangle_capacity);
/* page-locked, device-readable host memory for a host's projected arrays (NULL when there is no device or no memory: fall back to
 * ordinary memory and rxr_stream_begin).  No context needed; free with rxr_free_pinned. */
void *rxr_alloc_pinned(size_t bytes);
void rxr_free_pinned(void *ptr);
int rxr_stream_batch3d(rxr_ctx *ctx, uint32_t index, const rxr_batch3d *projected);

/* validates + flattens a projected frame and copies it to HBM (replaces nothing in the reference:
 * it is the host->device hand-over).  The frame stays resident until the next upload. */
int rxr_upload_frame(rxr_ctx *ctx, const rxr_frame *frame);

/* renders rows [row0,row1) of the resident frame into the context's device framebuffer.
 * Replaces the tile list + rayon tile loop, src/rasterizer.rs:256-557.  Asynchronous. */
int rxr_render_rows(rxr_ctx *ctx, uint32_t row0, uint32_t row1);
/* same, but writes the band into caller-owned DEVICE memory (`dev_pixels` points at row `row0`,
 * rows are `width*4` bytes apart) on HIP stream `hip_stream` (hipStream_t; NULL means the context's own
 * non-blocking stream, NOT the legacy default stream -- pass an explicit stream to order with other work).
 * Used by the multi-GPU host, which then gathers the bands with RCCL.
 * Streams: a context has ONE set of scratch buffers, so a render on another stream than the context's previous render is ordered
 * behind that stream -- behind the earlier render AND whatever the caller queued on that stream after it (the event is recorded at the
 * switch, not after every render).  A caller that alternates streams must therefore not make later work of the OLD stream wait for
 * something the NEW stream produces: that wait would be circular.  Callers that stay on one stream, and lanes (one member context per
 * stream, rxr_create_multi), never meet the rule. */
int rxr_render_rows_to(rxr_ctx *ctx, uint32_t row0, uint32_t row1, void *dev_pixels, void *hip_stream);

/* multi-GPU sharding primitive: renders every `stride`-th stripe of RXR_STRIPE_ROWS pixel rows,
 * starting at stripe `first`, into a COMPACT caller-owned DEVICE buffer: local stripe j (frame rows
 * (first + j*stride)*RXR_STRIPE_ROWS ...) lands at dev_pixels + j*RXR_STRIPE_ROWS*width*4.  The
 * buffer must hold ceil((n_stripes - first) / stride) stripes.  Rank r of N renders (first=r,
 * stride=N); the compact buffers are then all-gathered (RCCL) and de-interleaved. */
#define RXR_STRIPE_ROWS 16u
int rxr_render_stripes_to(rxr_ctx *ctx, uint32_t first, uint32_t stride, void *dev_pixels, void *hip_stream);

/* K frames with one host call: renders the resident frame's stripes (first, stride: as rxr_render_stripes_to) `n_frames` times;
 * frame k's compact stripe buffer lands at dev_pixels + k * frame_stride_bytes (frame_stride_bytes >= one compact buffer), and the
 * batch is complete on `hip_stream` (NULL: the context's own stream).  A rank's share of a small frame is tens of microseconds of
 * GPU work -- less than the host spends on a call per frame plus a collective per frame -- so a multi-GPU host renders a bucket of
 * frames per call and exchanges the bucket with one collective (bench.py --bucket).
 * On a multi-device handle whose members ALL sit on one device (rxr_create_multi with the same id L times: "lanes", L frames in
 * flight on that GPU) frame k is rendered by member k mod L on that member's own stream: every member has its own resident frame
 * and scratch, so consecutive frames overlap on the GPU -- the tail of one launch sequence, when most of the chip is already idle,
 * runs under the head of the next (a 1/8 share of the 4K bench frame: 31 us per frame on one stream, 17.5 us with two lanes,
 * profiles/r03/).  The lanes' streams are forked from and joined to `hip_stream` with events.  Members on different devices:
 * RXR_ERR_UNSUPPORTED.  Every frame of the batch is byte-identical to rxr_render_stripes_to's.  Asynchronous; rxr_synchronize
 * waits for it.  The reference has no counterpart: it renders one frame per call (src/rasterizer.rs:185-193). */
int rxr_render_stripes_batch(rxr_ctx *ctx, uint32_t first, uint32_t stride, uint32_t n_frames, void *dev_pixels, size_t frame_stride_bytes,
                             void *hip_stream);

/* per-launch kernel timing for the benchmark: after rxr_profile_begin(ctx, n) every kernel of a render is launched with a start /
 * stop HIP event pair of its own (hipExtLaunchKernel, on the stream the render launches on) into a ring of n slots: the runtime
 * fills the pair with the dispatch's begin / end timestamps, the figures a profiler's kernel trace shows.  rxr_profile_read
 * synchronizes and returns, per sampled render, the SUM of its set-up kernels' durations and its raster kernel's, in microseconds
 * (time between kernels is in neither).  Nothing is recorded on the stream between the launches (rounds 1-3 did: each record is a
 * barrier packet that idles the GPU for microseconds and counts launch latency as kernel time).
 * rxr_profile_begin(ctx, 0) switches the timing off again (the default). */
int rxr_profile_begin(rxr_ctx *ctx, uint32_t max_frames);
/* sample every `stride`-th render only (default 1) */
int rxr_profile_stride(rxr_ctx *ctx, uint32_t stride);
int rxr_profile_read(rxr_ctx *ctx, float *setup_us, float *raster_us, uint32_t capacity, uint32_t *n_out);

/* copies rows [row0,row1) of the context's framebuffer into host `pixels` (full-frame layout:
 * row r goes to pixels + r*width*4).  Replaces the serial tile->framebuffer copy, :559-579.
 * Blocks until the bytes have landed. */
int rxr_download_rows(rxr_ctx *ctx, uint8_t *pixels, uint32_t row0, uint32_t row1);

/* the whole drop-in call: upload + render all rows + download.  `pixels` is width*height*4 bytes
 * of host memory, fully overwritten, as in the reference (src/rasterizer.rs:185-193). */
int rxr_rasterize(rxr_ctx *ctx, const rxr_frame *frame, uint8_t *pixels);
/* the second half of rxr_rasterize for callers that upload separately: renders every row of the uploaded frame and
 * writes width*height*4 bytes to `pixels` (host memory).  Frames that need no per-tile lists (at most 128 3D triangles,
 * few 2D primitives) are rendered in bands whose downloads overlap the rendering of the following bands; the result is
 * byte-identical to rxr_render_rows + rxr_download_rows.  Replaces src/rasterizer.rs:256-579 + the final copy. */
int rxr_render_download(rxr_ctx *ctx, uint8_t *pixels);

/* device-resident consumers: renders the whole uploaded frame and leaves it, in frame layout, in DEVICE memory of member
 * `root` (`dev_pixels` on that device, or that member's own framebuffer when NULL), complete on `hip_stream` (a stream of
 * the root's device; NULL = the root member's own stream).  On a multi-device context every other member renders its
 * stripes and pushes them to the root with peer copies over xGMI (one link per source GPU, all concurrent, each stripe
 * landing at its final place), the root renders its own stripes in place.  On a plain context (root 0) it is
 * rxr_render_rows_to over all rows.  Asynchronous; rxr_synchronize waits for it.
 * Replaces the tile loop + the final tile concatenation, src/rasterizer.rs:256-579. */
int rxr_render_gather(rxr_ctx *ctx, int root, void *dev_pixels, void *hip_stream);

/* Waits for everything the context has queued -- on its own streams AND on the caller's stream of the last
 * rxr_render_rows_to / rxr_render_stripes_to -- and reports what the launches since the previous call left behind:
 *   RXR_OK            every frame since the last call is complete (a bin-list overflow of the LAST launch is repaired here:
 *                     the lists are grown and that launch is rendered again);
 *   RXR_ERR_OVERFLOW  see above: an EARLIER launch overflowed; its frame was incomplete;
 *   RXR_ERR_INVALID   a fragment's Rusteria program did what makes the reference panic;
 *   RXR_ERR_UNSUPPORTED a pixel's opacity staircase overflowed (four nested GROUPS of opacity batches -- runs of opacity batches
 *                       without a profiled opaque batch between them --).
 * Stream contract: every render of a context uses the context's one set of scratch buffers.  A render on a different
 * stream than the previous one is ordered behind it by the library; rxr_upload_frame / rxr_set_* wait (on the host) for
 * all renders, including those on caller streams, before they overwrite anything.  A caller that queues frame after
 * frame without synchronizing must call rxr_synchronize before it declares those frames done. */
int rxr_synchronize(rxr_ctx *ctx);
int rxr_get_stats(rxr_ctx *ctx, rxr_stats *out);
/* device pointer of the context framebuffer (width*height*4 bytes of the last uploaded frame) */
void *rxr_device_framebuffer(rxr_ctx *ctx);

/* diagnostics: the raster kernel replaces hipcc's expansions of f32 division, sqrtf and exp2f(k*log2f(x))
 * by shorter instruction sequences that are bit-identical (rusterix_amd/csrc/rxr_exact_math.h).  This
 * runs both over `n_tuples` seeded operand tuples per operation kind on the device (rounded up to whole
 * workgroups) and returns the number of tuples whose results differ in mismatches[0..RXR_MATH_KINDS-1]:
 * 0 div2, 1 div3, 2 div3_self, 3 normalize3, 4 sqrt, 5 pow, 6 div1, 7 pixel-centre/size and byte/255,
 * 8 normalize3 with exactly-zero components (surface normals), 9 sqrt over a strided sweep of its whole
 * operand window (stride and phase from the seed; the sweep covers n_tuples operands), 10 the saturating float -> u32 conversion
 * (Rust's `as` casts, one v_cvt_u32_f32) against its compare-and-select form over a strided sweep of all bit patterns,
 * 11 the wave maximum of non-negative floats and 12 the wave's inclusive prefix sum through DPP operands against the shuffle forms
 * (at most 64 tuples per lane each).
 * Blocking.  Nothing in the reference corresponds to it. */
#define RXR_MATH_KINDS 13
int rxr_selftest_math(rxr_ctx *ctx, uint64_t n_tuples, uint64_t seed, uint64_t mismatches[RXR_MATH_KINDS]);

#ifdef __cplusplus
}
#endif
#endif /* RXR_H */
