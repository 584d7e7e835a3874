// rusterix_oracle.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement ("port") of the Rusterix tile rasterizer hot path, written to follow the
// reference source line by line so that it can serve as the parity oracle for the HIP path and as
// the CPU baseline in bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may call into this directory.  The product (rusterix_amd/) never includes or links it.
//
// PARITY PINNING: the reference ships no tests, golden images or known-answer vectors for the
// rasterizer and cannot be compiled here (no rustc/cargo).  This oracle is therefore
// pinned only by (a) the source text it cites, and (b) the analytic known-answer tests derived from
// that text in tests/test_oracle_known_answers.py (SURVEY.md section 8c list).  The arithmetic of the
// third-party crate vek 0.17.2 (un-vendored) is restated in include/rusterix_vek.hpp from its
// published behaviour: "parity unpinned" at that boundary.
//
// All citations are file:line under /root/reference/.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "../include/rusterix_vek.hpp"
#include "../include/rxr.h"  // enum values + rxr_light POD (data layout only)
#include "rusteria_vm.hpp"

namespace orc {

using rvek::Mat3;
using rvek::Mat4;
using rvek::Vec2;
using rvek::Vec3;
using rvek::Vec4;

typedef uint8_t Pixel[4];

// ---- Rust numeric semantics --------------------------------------------
// (rmin / rmax -- f32::min / f32::max with the x86-64 tie rule -- live in rusteria_vm.hpp, which the VM needs them in)
inline float rclamp(float x, float lo, float hi) { return rvek::rclamp(x, lo, hi); }  // keeps NaN
inline uint64_t sat_usize(float x) {  // `x as usize`
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)x;
}
inline int64_t sat_isize(float x) {  // `x as isize`
    if (!(x == x)) return 0;
    if (x <= -9223372036854775808.0f) return INT64_MIN;
    if (x >= 9223372036854775808.0f) return INT64_MAX;
    return (int64_t)x;
}
inline int32_t sat_i32(float x) {
    if (!(x == x)) return 0;
    if (x <= -2147483648.0f) return INT32_MIN;
    if (x >= 2147483648.0f) return INT32_MAX;
    return (int32_t)x;
}
inline uint32_t sat_u32(float x) {
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)x;
}
inline uint8_t sat_u8(float x) {
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

// ---- data model (reference struct -> here) ----------------------------------------------------

// src/edge.rs:2-8
struct Edges {
    float a[3], b[3], c[3];
    bool visible;
};
// src/edge.rs:12-24
Edges edges_new(const float v0[3][2], const float v1[3][2], bool visible);
// src/edge.rs:28-36
inline bool edges_evaluate(const Edges &e, const float p[2]) {
    for (int i = 0; i < 3; ++i) {
        float result = e.a[i] * p[0] + e.b[i] * p[1] + e.c[i];
        if (result < 0.0f) return false;
    }
    return true;
}

// src/rect.rs:5-10
struct Rect {
    float x, y, width, height;
};

// src/map/pixelsource.rs:23-37 (variants the raster loops distinguish; see include/rxr.h)
struct Source {
    uint32_t kind = RXR_SOURCE_OTHER;  // RXR_SOURCE_*, or RXR_HOST_SOURCE_ENTITY_TILE / _ITEM_TILE (index = id, seq = sequence index)
    uint32_t index = 0;
    uint8_t pixel[4] = {0, 0, 0, 0};
    uint32_t seq = 0;
};

// src/texture.rs:46-54
struct Texture {
    std::vector<uint8_t> data;
    size_t width = 0, height = 0;
};
// src/map/tile.rs (only `textures`)
struct Tile {
    std::vector<Texture> textures;
};
// src/server/assets.rs (only `tile_list` and `palette`); `vm_env` also carries rusteria's process-global
// pattern banks (rusteria/src/textures/patterns.rs) so that tests can supply their own
struct Assets {
    std::vector<Tile> tile_list;
    // src/server/assets.rs:28, :34: FxHashMap<u32, IndexMap<String, Tile>>.  The raster loops only use `.get(&id)` and
    // `.get_index(i)` (rasterizer.rs:1140-1187), so an id -> ordered list of tiles carries everything they read
    std::map<uint32_t, std::vector<Tile>> entity_tiles, item_tiles;
    vm::Env vm_env;
};

enum CullMode { CullOff = 0, CullFront = 1, CullBack = 2 };  // src/batch/mod.rs:17-26

// src/batch/batch3d.rs:15-78
struct Batch3D {
    int mode = RXR_MODE_TRIANGLES;
    std::vector<std::array<float, 4>> vertices;
    std::vector<std::array<size_t, 3>> indices;
    std::vector<std::array<float, 2>> uvs;
    std::vector<std::array<float, 4>> projected_vertices;
    bool has_bounding_box = false;
    Rect bounding_box{0, 0, 0, 0};
    std::vector<Edges> edges;
    int repeat_mode = RXR_REPEAT_CLAMP_XY;
    int cull_mode = CullOff;
    Source source;
    std::vector<std::array<size_t, 3>> clipped_indices;
    std::vector<std::array<float, 2>> clipped_uvs;
    Mat4 transform_3d = Mat4::identity();
    bool receives_light = true;
    std::vector<Vec3> normals;
    std::vector<Vec3> clipped_normals;
    Vec3 ambient_color{0, 0, 0};
    int shader = -1;
    bool has_profile_id = false;
    uint32_t profile_id = 0;
};

// src/batch/batch2d.rs:10-52
struct Batch2D {
    int mode = RXR_MODE_TRIANGLES;
    std::vector<std::array<float, 2>> vertices;
    std::vector<std::array<size_t, 3>> indices;
    std::vector<std::array<float, 2>> uvs;
    std::vector<std::array<float, 2>> projected_vertices;
    bool has_bounding_box = false;
    Rect bounding_box{0, 0, 0, 0};
    std::vector<Edges> edges;
    int repeat_mode = RXR_REPEAT_CLAMP_XY;
    Source source;
    bool receives_light = true;
    int shader = -1;
};

typedef rxr_light CompiledLight;  // src/map/light.rs:456-477, same fields

// src/map/bbox.rs + occlusion value
struct Occluder {
    Vec2 min, max;
    float occlusion;
};
struct Linedef {
    Vec2 start, end;
};
// src/map/mini.rs (fields the path reads)
struct MapMini {
    std::vector<Occluder> occluded_sectors;
    std::vector<Linedef> linedefs;
};

// src/chunk.rs (fields the path reads)
struct Chunk {
    std::vector<Batch3D> batches3d_opacity;
    std::vector<Batch3D> batches3d;
    std::vector<Batch2D> batches2d;
    std::vector<CompiledLight> lights;
    std::vector<Occluder> occluded_sectors;
    std::vector<vm::Program> shaders;  // src/chunk.rs:51
    // Option<..> fields as 0- or 1-element vectors / presence flags
    std::vector<Batch2D> terrain_batch2d;        // src/chunk.rs:34
    std::vector<Batch3D> terrain_batch3d;        // :35
    bool has_terrain_texture = false;            // :36
    Texture terrain_texture;
    int origin[2] = {0, 0};                      // :25
    int size = 1;                                // :26
    std::vector<Texture> shader_textures;        // :53: Vec<Option<Texture>>
    std::vector<uint8_t> shader_texture_present;
};
// src/chunk.rs:133-151; false where the reference panics (size == 0: integer division by zero)
bool chunk_sample_terrain_texture(const Chunk &c, Vec2 world_pos, Vec2 scale, uint8_t out[4]);

// src/scene.rs:8-50
struct Scene {
    int background = RXR_BG_NONE;  // Option<Box<dyn Shader>>: none | VGrayGradientShader | GridShader
    float background_grid[4] = {30.0f, 2.0f, 0.0f, 0.0f};  // GridShader: grid_size, subdivisions, offset (shader/grid.rs:12-16)
    std::vector<CompiledLight> lights;
    std::vector<CompiledLight> dynamic_lights;
    std::vector<Batch3D> d3_static, d3_dynamic, d3_overlay;
    std::vector<Batch2D> d2_static, d2_dynamic;
    std::vector<Tile> dynamic_textures;
    size_t animation_frame = 1;  // src/scene.rs:72
    std::vector<Chunk> chunks;   // FxHashMap in the reference; here: the host's iteration order
    std::vector<vm::Program> shaders;  // src/scene.rs:43
};

// src/rasterizer.rs:35-88
struct Rasterizer {
    bool d2_active = true, d3_active = true, ignore_background_shader = false;  // RenderMode
    bool has_m2d = false;
    Mat3 projection_matrix_2d = Mat3::identity();
    Mat4 view_matrix = Mat4::identity(), projection_matrix = Mat4::identity();
    Mat4 inverse_view_matrix = Mat4::identity(), inverse_projection_matrix = Mat4::identity();
    float width = 0, height = 0;
    Vec3 camera_pos;
    MapMini mapmini;
    int sample_mode = RXR_SAMPLE_NEAREST;
    uint32_t hash_anim = 0;
    bool has_background_color = false;
    uint8_t background_color[4] = {0, 0, 0, 0};
    bool has_ambient = false;
    Vec4 ambient_color;
    Vec2 translationd2{0, 0};
    float scaled2 = 1.0f;
    bool preserve_transparency = false;
    float time = 0.0f;
    bool has_sun = false;
    Vec3 sun_dir;
    float day_factor = 0.0f;
    // Rasterizer.brush_preview: Option<BrushPreview> (src/rasterizer.rs:13-17, :65)
    bool has_brush_preview = false;
    Vec3 brush_position;
    float brush_radius = 0.0f, brush_falloff = 0.0f;
};

// ---- restated functions --------------------------------------------------------------------------
uint32_t hash_u32(uint32_t seed);                                     // src/rasterizer.rs:199-207
void pixel_to_vec4(const uint8_t p[4], float out[4]);                 // src/lib.rs:55-62
uint8_t f32_to_u8_saturated(float x);                                 // src/lib.rs:64-68
void vec4_to_pixel(const float v[4], uint8_t out[4]);                 // src/lib.rs:71-79
float srgb_to_linear_fast(float x);                                   // src/rasterizer.rs:19-25
float linear_to_srgb_fast(float x);                                   // src/rasterizer.rs:27-33
void texture_sample(const Texture &t, float u, float v, int sample_mode, int repeat_mode, uint8_t out[4]);  // src/texture.rs:203-232
void texture_sample_nearest(const Texture &t, float u, float v, uint8_t out[4]);  // :307-323
void texture_sample_linear(const Texture &t, float u, float v, uint8_t out[4]);   // :414-460
bool light_color_at(const CompiledLight &l, Vec3 point, uint32_t hash, bool d2, float out[3]);        // src/map/light.rs:491-502
bool light_radiance_at(const CompiledLight &l, Vec3 point, bool has_n, Vec3 n, uint32_t hash, Vec3 &out);  // :504-533
float mapmini_get_occlusion(const std::vector<Occluder> &occ, Vec2 at);  // src/map/mini.rs:58-66
bool mapmini_is_visible(const MapMini &m, Vec2 from, Vec2 to);            // src/map/mini.rs:88-95

Batch3D batch3d_from_box(float x, float y, float z, float w, float h, float d);  // src/batch/batch3d.rs:140-229
void batch3d_add(Batch3D &b, const float *verts4, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2);  // :238-253
void batch3d_compute_vertex_normals(Batch3D &b);                                  // :771-809
// returns false where the reference would panic (empty normals, :605-607)
bool batch3d_clip_and_project(Batch3D &b, const Mat4 &view, const Mat4 &proj, float vw, float vh);  // :482-740
Batch2D batch2d_from_rectangle(float x, float y, float w, float h);               // src/batch/batch2d.rs:109-127
void batch2d_project(Batch2D &b, const Mat3 *matrix);                             // :373-425
Batch3D batch3d_from_obj(const char *text);                                       // src/wavefront.rs:34-102

Rasterizer rasterizer_setup(const Mat3 *m2d, const Mat4 &view, const Mat4 &proj);  // src/rasterizer.rs:92-152
bool scene_project(Scene &s, const Mat3 *m2d, const Mat4 &view, const Mat4 &proj, float w, float h);  // src/scene.rs:154-200
// src/rasterizer.rs:185-580; n_threads plays rayon's pool.  Returns 0 or a negative rxr_status
// where the reference would panic.
int rasterize(Rasterizer &r, Scene &scene, uint8_t *pixels, size_t width, size_t height, size_t tile_size,
              const Assets &assets, int n_threads);

// cameras, src/camera/d3orbit.rs:23-56,186-195 and src/camera/d3firstp.rs:17-42
void orbit_camera(Vec3 center, float distance, float azimuth, float elevation, float fov, float near, float far,
                  float w, float h, Mat4 &view, Mat4 &proj);
void firstp_camera(Vec3 position, Vec3 center, float fov, float near, float far, float w, float h, Mat4 &view,
                   Mat4 &proj);

}  // namespace orc
