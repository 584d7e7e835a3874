// rusterix_host.hpp -- C++ mirror of the reference's host-side API for the rasterizer path.
//
// In a real deployment this layer is Rusterix itself (Rust): Scene / Batch2D / Batch3D / Assets /
// Rasterizer with `Rasterizer::setup(..).rasterize(&mut scene, pixels, w, h, tile_size, &assets)`.
// No Rust toolchain exists in this image, so the same surface is mirrored in C++ with the same
// names, argument meaning and error behaviour.  What stays on the host is exactly what the
// north-star keeps there: scene set-up, Scene::project (batch projection + near clipping) and the
// Edges precompute.  Everything after `scene.project(..)` crosses the C ABI of include/rxr.h.
//
// Storage is flat (std::vector<float> / uint32_t / rxr_edges) so that a projected batch can be handed
// to rxr_batch3d / rxr_batch2d without repacking.
//
// All citations are file:line under /root/reference/.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <functional>
#include <new>
#include <vector>

#include "../../../include/rusterix_vek.hpp"
#include "../../../include/rxr.h"

namespace rusterix {

// Allocator of the projected arrays: page-locked, device-readable memory from the device library (rxr_alloc_pinned) while that
// works, ordinary memory otherwise (no GPU: the CPU tests project with this mirror too).  Large blocks only -- a page-locked block
// costs a system call, and small scenes are not streamed anyway.  all_pinned() tells Rasterizer::upload whether the promise of
// rxr_stream_begin_pinned can be made: false as soon as ONE block of at least the threshold had to come from malloc.
struct PinnedPool {
    static constexpr size_t threshold = 8u << 10;
    static bool &any_unpinned() {
        static bool v = false;
        return v;
    }
    static void *take(size_t bytes, bool &pinned);
    static void give(void *p, bool pinned);
};
template <class T> struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n) {
        // one header word in front of the block says where it came from
        const size_t bytes = n * sizeof(T) + 64;
        bool pinned = false;
        uint8_t *raw = (uint8_t *)PinnedPool::take(bytes, pinned);
        if (!raw) throw std::bad_alloc();
        *(uint64_t *)raw = pinned ? 1u : 0u;
        return (T *)(raw + 64);
    }
    void deallocate(T *p, size_t) {
        uint8_t *raw = (uint8_t *)p - 64;
        PinnedPool::give(raw, *(uint64_t *)raw != 0u);
    }
    template <class U> bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
template <class T> using PinnedVec = std::vector<T, PinnedAlloc<T>>;
// is the vector's CURRENT buffer page-locked?  (the allocator's header word in front of it)
// (a vector without a buffer -- a batch that was always rejected -- has nothing the device could read: fine)
template <class T> inline bool is_pinned(const PinnedVec<T> &v) { return v.capacity() == 0 || *(const uint64_t *)((const uint8_t *)v.data() - 64) == 1u; }


using rvek::Mat3;
using rvek::Mat4;
using rvek::Vec2;
using rvek::Vec3;
using rvek::Vec4;

using Pixel = uint8_t[4];

enum class CullMode { Off = 0, Front = 1, Back = 2 };  // src/batch/mod.rs:17-26

// src/map/pixelsource.rs:23-37
struct PixelSource {
    uint32_t kind = RXR_SOURCE_OTHER;  // RXR_SOURCE_*, or RXR_HOST_SOURCE_ENTITY_TILE / _ITEM_TILE (index = id, seq = sequence index)
    uint32_t index = 0;
    uint8_t pixel[4] = {0, 0, 0, 0};
    uint32_t seq = 0;
    static PixelSource EntityTile(uint32_t id, uint32_t index) { PixelSource s; s.kind = RXR_HOST_SOURCE_ENTITY_TILE; s.index = id; s.seq = index; return s; }
    static PixelSource ItemTile(uint32_t id, uint32_t index) { PixelSource s; s.kind = RXR_HOST_SOURCE_ITEM_TILE; s.index = id; s.seq = index; return s; }
    static PixelSource Off() { return {}; }
    static PixelSource StaticTileIndex(uint16_t i) { PixelSource s; s.kind = RXR_SOURCE_STATIC_TILE; s.index = i; return s; }
    static PixelSource DynamicTileIndex(uint16_t i) { PixelSource s; s.kind = RXR_SOURCE_DYNAMIC_TILE; s.index = i; return s; }
    static PixelSource FromPixel(const uint8_t p[4]) { PixelSource s; s.kind = RXR_SOURCE_PIXEL; for (int i = 0; i < 4; ++i) s.pixel[i] = p[i]; return s; }
};

// src/texture.rs:46-54
struct Texture {
    std::vector<uint8_t> data;
    uint32_t width = 0, height = 0;
};
// src/map/tile.rs
struct Tile {
    std::vector<Texture> textures;
};
// src/server/assets.rs
uint64_t next_generation();  // process-wide unique stamps (never reused, unlike addresses)
// rusteria TexStorage (rusteria/src/textures/mod.rs:10-15): width * height RGB f32
struct Pattern {
    uint32_t width = 0, height = 0;
    std::vector<float> rgb;
};

struct Assets {
    std::vector<Tile> tile_list;
    // src/server/assets.rs:28, :34 (FxHashMap<u32, IndexMap<String, Tile>>): id -> sequence tiles in insertion order, which is all
    // the raster loops use (`.get(&id)`, `.get_index(i)`, rasterizer.rs:1140-1187).  Rasterizer::upload resolves
    // EntityTile / ItemTile sources against these and ships the tiles with the dynamic textures.
    std::map<uint32_t, std::vector<Tile>> entity_tiles, item_tiles;
    uint64_t generation = next_generation();  // re-stamped on every mutation; lets the rasterizer skip texture re-uploads
    Assets &textures(std::vector<Tile> tiles) { tile_list = std::move(tiles); generation = next_generation(); return *this; }
    // what a Rusteria program reads besides its own code: assets.palette (ThePalette.colors) and rusteria's
    // process-global pattern banks (rusteria/src/textures/patterns.rs), which the caller hands over as data
    std::vector<Pattern> patterns, patterns_normal;
    std::vector<float> palette_rgb;        // [n][3]
    std::vector<uint8_t> palette_present;  // [n]
    uint64_t shader_env_generation = next_generation();
};

// rusteria::Program (rusteria/src/node/program.rs:7-29): user functions as NodeOp trees in the word
// serialisation of include/rxr.h (there is no Rusteria parser / compiler on this side)
struct Program {
    uint32_t globals = 0;
    int32_t shade_index = -1;
    uint32_t shade_locals = 0;
    std::vector<std::vector<uint32_t>> user_functions;
};

// src/rect.rs
struct Rect {
    float x = 0, y = 0, width = 0, height = 0;
};

using CompiledLight = rxr_light;  // src/map/light.rs:456-477

// src/batch/batch3d.rs:15-78
class Batch3D {
public:
    std::vector<float> vertices;           // [n][4]
    std::vector<uint32_t> indices;         // [m][3]
    std::vector<float> uvs;                // [n][2]
    std::vector<float> normals;            // [n][3] or empty
    // what clip_and_project produces lives in page-locked memory when a device is there (PinnedAlloc below): the library's streaming
    // hand-over then lets the GPU read it in place (rxr_stream_begin_pinned) instead of copying it into a staging buffer first
    PinnedVec<float> projected_vertices;    // [n'][4]
    PinnedVec<uint32_t> clipped_indices;    // [m'][3]
    PinnedVec<float> clipped_uvs;           // [n'][2]
    PinnedVec<float> clipped_normals;       // [n'][3]
    PinnedVec<rxr_edges> edges;             // [m']
    PinnedVec<uint32_t> edge_visible;       // [m']: edges[t].visible as a word -- what travels instead of `edges` when the device builds the
                                            // records itself (device_edges(), rxr_batch3d.edges == NULL, ABI 5)
    bool has_bounding_box = false;
    Rect bounding_box;
    uint32_t repeat_mode_ = RXR_REPEAT_CLAMP_XY;
    CullMode cull_mode_ = CullMode::Off;
    PixelSource source_;
    Mat4 transform_3d = Mat4::identity();
    bool receives_light_ = true;
    Vec3 ambient_color_{0, 0, 0};
    int shader_ = -1;
    bool has_profile_id = false;
    uint32_t profile_id_ = 0;
    // geometry stamp for the device-projection cache: re-stamped by every mutator below; code that edits the
    // public vectors directly calls touch().  Copies keep the stamp (same content).
    uint64_t geometry_stamp = next_generation();
    void touch() { geometry_stamp = next_generation(); }

    static Batch3D empty() { return Batch3D(); }
    static Batch3D make(const float *verts4, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2);  // Batch3D::new, :109-137
    static Batch3D from_box(float x, float y, float z, float w, float h, float d);                          // :140-229
    static Batch3D from_obj(const std::string &text);                                                      // :407-419 + src/wavefront.rs
    void add(const float *verts4, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2);          // :238-253
    void compute_vertex_normals();                                                                          // :771-809
    // :482-740.  Returns false where the reference panics (normals shorter than vertices, :605-607).
    bool clip_and_project(const Mat4 &view, const Mat4 &proj, float viewport_width, float viewport_height);

    size_t vertex_count() const { return vertices.size() / 4; }
    size_t triangle_count() const { return indices.size() / 3; }
};

// src/batch/batch2d.rs:10-52
class Batch2D {
public:
    uint32_t mode_ = RXR_MODE_TRIANGLES;
    std::vector<float> vertices;           // [n][2]
    std::vector<uint32_t> indices;         // [m][3]
    std::vector<float> uvs;                // [n][2]
    std::vector<float> projected_vertices; // [n][2]
    std::vector<rxr_edges> edges;          // [m]
    bool has_bounding_box = false;
    Rect bounding_box;
    uint32_t repeat_mode_ = RXR_REPEAT_CLAMP_XY;
    PixelSource source_;
    bool receives_light_ = true;
    int shader_ = -1;

    static Batch2D make(const float *verts2, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2);  // :83-106
    static Batch2D from_rectangle(float x, float y, float w, float h);                                       // :109-127
    void project(const Mat3 *matrix);                                                                         // :373-425
};

// src/map/mini.rs (fields the raster loops read)
struct MapMini {
    std::vector<rxr_occluder> occluded_sectors;
    std::vector<rxr_linedef> linedefs;
};

// src/chunk.rs (fields the raster loops read)
struct Chunk {
    std::vector<Batch3D> batches3d_opacity, batches3d;
    std::vector<Batch2D> batches2d;
    std::vector<CompiledLight> lights;
    std::vector<rxr_occluder> occluded_sectors;
    std::vector<Program> shaders;  // src/chunk.rs:51
    // Option<..> fields as 0- or 1-element vectors / presence flags
    std::vector<Batch2D> terrain_batch2d;           // src/chunk.rs:34
    std::vector<Batch3D> terrain_batch3d;           // :35
    bool has_terrain_texture = false;               // :36
    Texture terrain_texture;
    int32_t origin[2] = {0, 0};                     // :25
    int32_t size = 1;                               // :26
    std::vector<Texture> shader_textures;           // :53, Vec<Option<Texture>>
    std::vector<uint8_t> shader_texture_present;
};

// src/scene.rs:8-50
class Scene {
public:
    uint32_t background = RXR_BG_NONE;  // Option<Box<dyn Shader>>: VGrayGradientShader and GridShader are evaluated on the device
    float background_grid[4] = {30.0f, 2.0f, 0.0f, 0.0f};  // GridShader: grid_size, subdivisions, offset (shader/grid.rs:12-16)
    std::vector<CompiledLight> lights, dynamic_lights;
    std::vector<Batch3D> d3_static, d3_dynamic, d3_overlay;
    std::vector<Batch2D> d2_static, d2_dynamic;
    std::vector<Tile> dynamic_textures;
    uint64_t dynamic_textures_generation = next_generation();
    size_t animation_frame = 1;
    std::vector<Chunk> chunks;
    std::vector<Program> shaders;  // src/scene.rs:43
    uint64_t shaders_generation = next_generation();
    // scene.add_shader (src/scene.rs:104-134) minus parse + compile; returns the shader index
    size_t add_program(Program p) { shaders.push_back(std::move(p)); shaders_generation = next_generation(); return shaders.size() - 1; }

    // src/scene.rs:154-200.  false where the reference panics.
    // on_projected3d (optional): called on the projecting thread right after clip_and_project of 3D batch `index` (index = the batch's
    // position in the submission order of Rasterizer::upload's frame) -- the hook of the streaming hand-over (rxr_stream_batch3d)
    bool project(const Mat3 *m2d, const Mat4 &view, const Mat4 &proj, float width, float height,
                 const std::function<void(size_t index, const Batch3D &)> *on_projected3d = nullptr);
    // the 3D batches in submission order (what project() walks and Rasterizer::upload flattens)
    std::vector<const Batch3D *> batches3d_in_order() const;
    // the 2D half only (src/scene.rs:163-187); used when the 3D half runs on the device
    void project_2d(const Mat3 *m2d);
};

// src/rasterizer.rs:35-193
class Rasterizer {
public:
    bool d2_active = true, d3_active = true, ignore_background_shader = false;  // RenderMode
    bool has_m2d = false;
    Mat3 projection_matrix_2d = Mat3::identity();
    Mat4 view_matrix = Mat4::identity(), projection_matrix = Mat4::identity();
    Mat4 inverse_view_matrix = Mat4::identity(), inverse_projection_matrix = Mat4::identity();
    float width = 0, height = 0;
    Vec3 camera_pos;
    MapMini mapmini;
    uint32_t sample_mode_ = RXR_SAMPLE_NEAREST;
    uint32_t hash_anim = 0;
    bool has_background_color = false;
    uint8_t background_color[4] = {0, 0, 0, 0};
    // brush_preview: Option<BrushPreview> (src/rasterizer.rs:13-17, :65)
    bool has_brush_preview = false;
    float brush_position[3] = {0, 0, 0};
    float brush_radius = 0.0f, brush_falloff = 0.0f;
    bool has_ambient = false;
    Vec4 ambient_color;
    Vec2 translationd2{0, 0};
    float scaled2 = 1.0f;
    bool preserve_transparency = false;
    float time_ = 0.0f;
    bool has_sun = false;
    Vec3 sun_dir;
    float day_factor = 0.0f;

    static Rasterizer setup(const Mat3 *projection_matrix_2d, const Mat4 &view, const Mat4 &proj);  // :92-152

    // :185-580.  Projects the scene on the host, flattens it and renders it on the GPU through the
    // C ABI.  Returns RXR_OK or a negative rxr_status (the reference panics / has no error path).
    int rasterize(Scene &scene, uint8_t *pixels, size_t width, size_t height, size_t tile_size, const Assets &assets);
    // the same up to and including the host->device hand-over (project + flatten + rxr_upload_frame);
    // callers then drive rxr_render_rows / rxr_render_rows_to themselves (bench, multi-GPU host)
    int upload(Scene &scene, size_t width, size_t height, size_t tile_size, const Assets &assets);
};

// the process-wide device context.  Every function that uses it holds one process-wide lock for its whole duration, so
// Rasterizers on several threads are serialised (the reference's are independent values; this layer shares one context).  Device index from RXR_DEVICE, else
// LOCAL_RANK, else 0.
rxr_ctx *context(std::string *error = nullptr);
// device-side projection: when on, Rasterizer::upload registers the object-space
// batches once (rxr_set_meshes) and each frame sends matrices only; Scene::project's 3D half
// (clip_and_project, Edges::new, bounding boxes) then runs on the GPU.  Off by default.
void set_device_projection(bool on);
bool device_projection();
// host-projected 3D batches cross the ABI WITHOUT their Edges records (40 of 124 bytes per triangle over PCIe): a word per triangle says
// `visible`, the device rebuilds a / b / c from the projected vertices it receives anyway (rxr.h, ABI 5).  Default on; RXR_HOST_EDGES=1 or
// set_device_edges(false) sends the records as rounds 1-3 did.
void set_device_edges(bool on);
bool device_edges();
// arithmetic of the 3D light loop (rxr_set_light_math, include/rxr.h): relaxed by default -- point lights within the 1-per-channel
// tolerance of lit 3D fragments; exact = the reference's correctly rounded operations throughout.  Applies from the next frame on.
void set_light_math(bool exact);
bool light_math_exact();
void set_device(int device);
// more than one entry: the context becomes a multi-device one (rxr_create_multi): rasterize() then shards every frame over
// these GPUs inside the library.  A device may be listed more than once (logical members on one GPU).
void set_devices(const int *devices, int n);
const std::string &last_error();

// cameras: src/camera/d3orbit.rs:23-56,186-195 and src/camera/d3firstp.rs:17-42
void orbit_camera(Vec3 center, float distance, float azimuth, float elevation, float fov, float near, float far, float w,
                  float h, Mat4 &view, Mat4 &proj);
void firstp_camera(Vec3 position, Vec3 center, float fov, float near, float far, float w, float h, Mat4 &view, Mat4 &proj);

uint32_t hash_u32(uint32_t seed);  // src/rasterizer.rs:199-207

}  // namespace rusterix
