// host_capi.cpp -- extern "C" handle API over the C++ host mirror (rusterix_host.hpp), bound from
// Python by rusterix_amd/binding.py (prefix `rxh_`).  Product code: rasterize() goes through the
// C ABI of include/rxr.h to the HIP kernels; there is no CPU path here.
#include <cstring>
#include <string>

#include "rusterix_host.hpp"

using namespace rusterix;

namespace {
Mat4 mat4_from(const float *m) {
    Mat4 o{};
    memcpy(o.m, m, sizeof(o.m));
    return o;
}
Tile make_tile(const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    Tile t;
    for (uint32_t i = 0; i < n; ++i) {
        Texture x;
        x.width = ws[i];
        x.height = hs[i];
        x.data.assign(frames[i], frames[i] + (size_t)ws[i] * hs[i] * 4);
        t.textures.push_back(std::move(x));
    }
    return t;
}
std::vector<Batch3D> *list3d(Scene *s, int list, int chunk) {
    const bool chunk_ok = chunk >= 0 && (size_t)chunk < s->chunks.size();
    switch (list) {
        case RXR_LIST_CHUNK_OPACITY: return chunk_ok ? &s->chunks[chunk].batches3d_opacity : nullptr;
        case RXR_LIST_CHUNK: return chunk_ok ? &s->chunks[chunk].batches3d : nullptr;
        case RXR_LIST_CHUNK_TERRAIN: return chunk_ok ? &s->chunks[chunk].terrain_batch3d : nullptr;
        case RXR_LIST_STATIC: return &s->d3_static;
        case RXR_LIST_DYNAMIC: return &s->d3_dynamic;
        case RXR_LIST_OVERLAY: return &s->d3_overlay;
    }
    return nullptr;
}
void set_source(PixelSource &s, uint32_t kind, uint32_t index, const uint8_t *pixel) {
    s.kind = kind;
    s.index = index;
    if (pixel) memcpy(s.pixel, pixel, 4);
}
}  // namespace

extern "C" {

const char *rxh_last_error() { return last_error().c_str(); }
void rxh_set_device(int device) { set_device(device); }
// more than one device: the context becomes a multi-device one (rxr_create_multi); a device may repeat (logical members)
void rxh_set_devices(const int *devices, int n) { set_devices(devices, n); }
// the process-wide rxr_ctx (NULL + rxh_last_error() when no GPU): lets callers drive the split-phase
// ABI (rxr_render_rows_to / rxr_get_stats) after rxh_rasterizer_upload
void *rxh_context() { return context(); }
// device-side projection on/off (rusterix::set_device_projection)
void rxh_set_device_projection(int on) { set_device_projection(on != 0); }
// host-projected batches with (0) or without (1, the default) their Edges records across the ABI (rusterix::set_device_edges)
void rxh_set_device_edges(int on) { set_device_edges(on != 0); }
int rxh_get_device_edges() { return device_edges() ? 1 : 0; }
int rxh_get_device_projection() { return device_projection() ? 1 : 0; }
// light-loop arithmetic (rusterix::set_light_math): 1 = exact, 0 = relaxed (the default)
void rxh_set_light_math_exact(int exact) { set_light_math(exact != 0); }
int rxh_get_light_math_exact() { return light_math_exact() ? 1 : 0; }

// ---- scene ----------------------------------------------------------------------------------------
void *rxh_scene_new() { return new Scene(); }
void rxh_scene_free(void *s) { delete (Scene *)s; }
void rxh_scene_set_animation_frame(void *s, uint64_t f) { ((Scene *)s)->animation_frame = (size_t)f; }
void rxh_scene_set_background(void *s, int kind) { ((Scene *)s)->background = (uint32_t)kind; }
// GridShader::set_parameter_f32 / set_parameter_vec2 (shader/grid.rs:19-34)
void rxh_scene_set_background_grid(void *s, float grid_size, float subdivisions, float offset_x, float offset_y) {
    float *g = ((Scene *)s)->background_grid;
    g[0] = grid_size;
    g[1] = subdivisions;
    g[2] = offset_x;
    g[3] = offset_y;
}
void rxh_scene_add_light(void *s, const rxr_light *l, int dynamic) {
    (dynamic ? ((Scene *)s)->dynamic_lights : ((Scene *)s)->lights).push_back(*l);
}
void rxh_scene_add_dynamic_tile(void *s, const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    Scene *sc = (Scene *)s;
    sc->dynamic_textures.push_back(make_tile(frames, ws, hs, n));
    sc->dynamic_textures_generation = next_generation();
}
int rxh_scene_add_chunk(void *s) {
    ((Scene *)s)->chunks.emplace_back();
    return (int)((Scene *)s)->chunks.size() - 1;
}
void rxh_chunk_add_occluder(void *s, int chunk, float minx, float miny, float maxx, float maxy, float occ) {
    ((Scene *)s)->chunks[chunk].occluded_sectors.push_back(rxr_occluder{{minx, miny}, {maxx, maxy}, occ});
}
// chunk.terrain_texture / origin / size (src/chunk.rs:25-36); rgba == NULL: no texture
void rxh_chunk_set_terrain(void *s, int chunk, const uint8_t *rgba, uint32_t w, uint32_t h, int ox, int oy, int size) {
    Chunk &c = ((Scene *)s)->chunks[chunk];
    c.origin[0] = ox;
    c.origin[1] = oy;
    c.size = size;
    c.has_terrain_texture = rgba != nullptr;
    if (rgba) {
        c.terrain_texture.width = w;
        c.terrain_texture.height = h;
        c.terrain_texture.data.assign(rgba, rgba + (size_t)w * h * 4);
    }
}
void rxh_chunk_set_terrain_batch2d(void *s, int chunk, void *b) {
    Chunk &c = ((Scene *)s)->chunks[chunk];
    c.terrain_batch2d.clear();
    c.terrain_batch2d.push_back(*(Batch2D *)b);
}
// chunk.shader_textures.push(texture) (src/chunk.rs:129); rgba == NULL pushes None
void rxh_chunk_add_shader_texture(void *s, int chunk, const uint8_t *rgba, uint32_t w, uint32_t h) {
    Chunk &c = ((Scene *)s)->chunks[chunk];
    Texture t;
    if (rgba) {
        t.width = w;
        t.height = h;
        t.data.assign(rgba, rgba + (size_t)w * h * 4);
    }
    c.shader_textures.push_back(std::move(t));
    c.shader_texture_present.push_back(rgba ? 1 : 0);
}
void rxh_chunk_add_light(void *s, int chunk, const rxr_light *l) { ((Scene *)s)->chunks[chunk].lights.push_back(*l); }
// scene.add_shader (src/scene.rs:104-134) minus the parser / compiler; chunk >= 0: that chunk's shaders
int rxh_scene_add_program(void *s, int chunk, uint32_t n_globals, int32_t shade_index, uint32_t shade_locals,
                          const uint32_t *const *fn_words, const uint32_t *fn_lens, uint32_t n_functions) {
    Scene *sc = (Scene *)s;
    Program p;
    p.globals = n_globals;
    p.shade_index = shade_index;
    p.shade_locals = shade_locals;
    for (uint32_t i = 0; i < n_functions; ++i) p.user_functions.emplace_back(fn_words[i], fn_words[i] + fn_lens[i]);
    if (chunk >= 0) {
        if ((size_t)chunk >= sc->chunks.size()) return RXR_ERR_INVALID;
        sc->chunks[chunk].shaders.push_back(std::move(p));
        sc->shaders_generation = next_generation();
        return (int)sc->chunks[chunk].shaders.size() - 1;
    }
    return (int)sc->add_program(std::move(p));
}
uint32_t rxh_scene_num_dynamic_lights(void *s) { return (uint32_t)((Scene *)s)->dynamic_lights.size(); }

// ---- Batch3D --------------------------------------------------------------------------------------
void *rxh_batch3d_new(const float *v4, uint32_t nv, const uint32_t *idx, uint32_t nt, const float *uv2) {
    return new Batch3D(Batch3D::make(v4, nv, idx, nt, uv2));
}
void *rxh_batch3d_from_box(float x, float y, float z, float w, float h, float d) { return new Batch3D(Batch3D::from_box(x, y, z, w, h, d)); }
void *rxh_batch3d_from_obj(const char *text) { return new Batch3D(Batch3D::from_obj(text)); }
void rxh_batch3d_free(void *b) { delete (Batch3D *)b; }
void rxh_batch3d_add(void *b, const float *v4, uint32_t nv, const uint32_t *idx, uint32_t nt, const float *uv2) {
    ((Batch3D *)b)->add(v4, nv, idx, nt, uv2);
}
void rxh_batch3d_set_normals(void *b, const float *n3, uint32_t n) {
    ((Batch3D *)b)->normals.assign(n3, n3 + (size_t)n * 3);
    ((Batch3D *)b)->touch();
}
void rxh_batch3d_compute_vertex_normals(void *b) { ((Batch3D *)b)->compute_vertex_normals(); }
void rxh_batch3d_set_source(void *b, uint32_t kind, uint32_t index, const uint8_t *pixel) { set_source(((Batch3D *)b)->source_, kind, index, pixel); }
// PixelSource::EntityTile(id, seq) / ItemTile(id, seq)
void rxh_batch3d_set_source_seq(void *b, int is_item, uint32_t id, uint32_t seq) {
    ((Batch3D *)b)->source_ = is_item ? PixelSource::ItemTile(id, seq) : PixelSource::EntityTile(id, seq);
}
void rxh_batch3d_set_repeat_mode(void *b, int m) { ((Batch3D *)b)->repeat_mode_ = (uint32_t)m; }
void rxh_batch3d_set_cull_mode(void *b, int m) { ((Batch3D *)b)->cull_mode_ = (CullMode)m; }
void rxh_batch3d_set_ambient_color(void *b, float r, float g, float bl) { ((Batch3D *)b)->ambient_color_ = Vec3{r, g, bl}; }
void rxh_batch3d_set_transform(void *b, const float *m16) { ((Batch3D *)b)->transform_3d = mat4_from(m16); }
void rxh_batch3d_set_profile_id(void *b, int has, uint32_t id) {
    ((Batch3D *)b)->has_profile_id = has != 0;
    ((Batch3D *)b)->profile_id_ = id;
}
void rxh_batch3d_set_shader(void *b, int shader) { ((Batch3D *)b)->shader_ = shader; }
void rxh_batch3d_counts(void *b, uint32_t *nv, uint32_t *nt) {
    *nv = (uint32_t)((Batch3D *)b)->vertex_count();
    *nt = (uint32_t)((Batch3D *)b)->triangle_count();
}
uint32_t rxh_batch3d_num_normals(void *b) { return (uint32_t)(((Batch3D *)b)->normals.size() / 3); }
void rxh_batch3d_get_geometry(void *b, float *v4, uint32_t *idx, float *uv2, float *n3) {
    Batch3D *p = (Batch3D *)b;
    memcpy(v4, p->vertices.data(), p->vertices.size() * 4);
    memcpy(idx, p->indices.data(), p->indices.size() * 4);
    memcpy(uv2, p->uvs.data(), p->uvs.size() * 4);
    if (n3) memcpy(n3, p->normals.data(), p->normals.size() * 4);
}
int rxh_scene_push_batch3d(void *s, void *b, int list, int chunk) {
    auto *l = list3d((Scene *)s, list, chunk);
    if (!l) return RXR_ERR_INVALID;
    if (list == RXR_LIST_CHUNK_TERRAIN) l->clear();  // Option<Batch3D>: the latest one wins
    l->push_back(*(Batch3D *)b);
    return 0;
}

// ---- Batch2D --------------------------------------------------------------------------------------
void *rxh_batch2d_new(const float *v2, uint32_t nv, const uint32_t *idx, uint32_t nt, const float *uv2) {
    return new Batch2D(Batch2D::make(v2, nv, idx, nt, uv2));
}
void *rxh_batch2d_from_rectangle(float x, float y, float w, float h) { return new Batch2D(Batch2D::from_rectangle(x, y, w, h)); }
void rxh_batch2d_free(void *b) { delete (Batch2D *)b; }
void rxh_batch2d_set_mode(void *b, int m) { ((Batch2D *)b)->mode_ = (uint32_t)m; }
void rxh_batch2d_set_repeat_mode(void *b, int m) { ((Batch2D *)b)->repeat_mode_ = (uint32_t)m; }
void rxh_batch2d_set_source(void *b, uint32_t kind, uint32_t index, const uint8_t *pixel) { set_source(((Batch2D *)b)->source_, kind, index, pixel); }
void rxh_batch2d_set_source_seq(void *b, int is_item, uint32_t id, uint32_t seq) {
    ((Batch2D *)b)->source_ = is_item ? PixelSource::ItemTile(id, seq) : PixelSource::EntityTile(id, seq);
}
void rxh_batch2d_set_receives_light(void *b, int v) { ((Batch2D *)b)->receives_light_ = v != 0; }
void rxh_batch2d_set_shader(void *b, int shader) { ((Batch2D *)b)->shader_ = shader; }
int rxh_scene_push_batch2d(void *s, void *b, int dynamic, int chunk) {
    Scene *sc = (Scene *)s;
    if (chunk >= 0) {
        if ((size_t)chunk >= sc->chunks.size()) return RXR_ERR_INVALID;
        sc->chunks[chunk].batches2d.push_back(*(Batch2D *)b);
    } else {
        (dynamic ? sc->d2_dynamic : sc->d2_static).push_back(*(Batch2D *)b);
    }
    return 0;
}

// ---- Assets ---------------------------------------------------------------------------------------
void *rxh_assets_new() { return new Assets(); }
void rxh_assets_free(void *a) { delete (Assets *)a; }
void rxh_assets_add_tile(void *a, const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    Assets *as = (Assets *)a;
    as->tile_list.push_back(make_tile(frames, ws, hs, n));
    as->generation = next_generation();
}

// assets.entity_tiles / item_tiles: makes `id` known (an entry without sequences) ...
void rxh_assets_add_sequence_id(void *a, int is_item, uint32_t id) {
    Assets *as = (Assets *)a;
    (is_item ? as->item_tiles : as->entity_tiles)[id];
    as->generation = next_generation();
}
// ... and appends one sequence tile to it (IndexMap insertion order = get_index order)
void rxh_assets_add_sequence_tile(void *a, int is_item, uint32_t id, const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    Assets *as = (Assets *)a;
    (is_item ? as->item_tiles : as->entity_tiles)[id].push_back(make_tile(frames, ws, hs, n));
    as->generation = next_generation();
}

void rxh_assets_set_patterns(void *a, int normal, const float *const *rgb, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    Assets *as = (Assets *)a;
    std::vector<Pattern> &dst = normal ? as->patterns_normal : as->patterns;
    dst.clear();
    for (uint32_t i = 0; i < n; ++i) {
        Pattern t;
        t.width = ws[i];
        t.height = hs[i];
        t.rgb.assign(rgb[i], rgb[i] + (size_t)3 * ws[i] * hs[i]);
        dst.push_back(std::move(t));
    }
    as->shader_env_generation = next_generation();
}
void rxh_assets_set_palette(void *a, const float *rgb3, const uint8_t *present, uint32_t n) {
    Assets *as = (Assets *)a;
    as->palette_rgb.assign(rgb3, rgb3 + (size_t)3 * n);
    as->palette_present.assign(n, 1);
    if (present) as->palette_present.assign(present, present + n);
    as->shader_env_generation = next_generation();
}

// ---- Rasterizer -----------------------------------------------------------------------------------
void *rxh_rasterizer_setup(const float *m2d9, const float *view16, const float *proj16) {
    Mat3 m2d{};
    if (m2d9) memcpy(m2d.m, m2d9, sizeof(m2d.m));
    return new Rasterizer(Rasterizer::setup(m2d9 ? &m2d : nullptr, mat4_from(view16), mat4_from(proj16)));
}
void rxh_rasterizer_free(void *r) { delete (Rasterizer *)r; }
void rxh_rasterizer_render_mode(void *r, int d2, int d3, int ignore_bg) {
    Rasterizer *x = (Rasterizer *)r;
    x->d2_active = d2 != 0;
    x->d3_active = d3 != 0;
    x->ignore_background_shader = ignore_bg != 0;
}
void rxh_rasterizer_sample_mode(void *r, int m) { ((Rasterizer *)r)->sample_mode_ = (uint32_t)m; }
void rxh_rasterizer_brush_preview(void *r, int on, float px, float py, float pz, float radius, float falloff) {
    Rasterizer *x = (Rasterizer *)r;
    x->has_brush_preview = on != 0;
    x->brush_position[0] = px;
    x->brush_position[1] = py;
    x->brush_position[2] = pz;
    x->brush_radius = radius;
    x->brush_falloff = falloff;
}
void rxh_rasterizer_background(void *r, const uint8_t *px) {
    Rasterizer *x = (Rasterizer *)r;
    x->has_background_color = px != nullptr;
    if (px) memcpy(x->background_color, px, 4);
}
void rxh_rasterizer_ambient(void *r, const float *a4) {
    Rasterizer *x = (Rasterizer *)r;
    x->has_ambient = a4 != nullptr;
    if (a4) x->ambient_color = Vec4{a4[0], a4[1], a4[2], a4[3]};
}
void rxh_rasterizer_time(void *r, float t) { ((Rasterizer *)r)->time_ = t; }
void rxh_rasterizer_preserve_transparency(void *r, int v) { ((Rasterizer *)r)->preserve_transparency = v != 0; }
void rxh_rasterizer_sun(void *r, const float *dir3, float day_factor) {
    Rasterizer *x = (Rasterizer *)r;
    x->has_sun = dir3 != nullptr;
    if (dir3) x->sun_dir = Vec3{dir3[0], dir3[1], dir3[2]};
    x->day_factor = day_factor;
}
void rxh_rasterizer_mapmini_add_occluder(void *r, float minx, float miny, float maxx, float maxy, float occ) {
    ((Rasterizer *)r)->mapmini.occluded_sectors.push_back(rxr_occluder{{minx, miny}, {maxx, maxy}, occ});
}
void rxh_rasterizer_mapmini_add_linedef(void *r, float x0, float y0, float x1, float y1) {
    ((Rasterizer *)r)->mapmini.linedefs.push_back(rxr_linedef{{x0, y0}, {x1, y1}});
}
void rxh_rasterizer_get_derived(void *r, float *inv_view16, float *inv_proj16, float *camera_pos3) {
    Rasterizer *x = (Rasterizer *)r;
    memcpy(inv_view16, x->inverse_view_matrix.m, 64);
    memcpy(inv_proj16, x->inverse_projection_matrix.m, 64);
    camera_pos3[0] = x->camera_pos.x; camera_pos3[1] = x->camera_pos.y; camera_pos3[2] = x->camera_pos.z;
}
int rxh_rasterizer_rasterize(void *r, void *scene, uint8_t *pixels, uint32_t w, uint32_t h, uint32_t tile_size, void *assets) {
    return ((Rasterizer *)r)->rasterize(*(Scene *)scene, pixels, w, h, tile_size, *(Assets *)assets);
}
// project + flatten + host->device hand-over only (no render)
int rxh_rasterizer_upload(void *r, void *scene, uint32_t w, uint32_t h, uint32_t tile_size, void *assets) {
    return ((Rasterizer *)r)->upload(*(Scene *)scene, w, h, tile_size, *(Assets *)assets);
}
// host-side projection only (what stays on the host); used by the CPU tests
int rxh_scene_project(void *r, void *scene, uint32_t w, uint32_t h) {
    Rasterizer *x = (Rasterizer *)r;
    return ((Scene *)scene)->project(x->has_m2d ? &x->projection_matrix_2d : nullptr, x->view_matrix, x->projection_matrix, (float)w, (float)h)
               ? 0
               : RXR_ERR_INVALID;
}

// ---- introspection of projected batches -------------------------------------------------------------
int rxh_scene_batch3d_counts(void *s, int list, int chunk, uint32_t i, uint32_t *nv, uint32_t *nt, uint32_t *has_normals) {
    auto *l = list3d((Scene *)s, list, chunk);
    if (!l || i >= l->size()) return RXR_ERR_INVALID;
    const Batch3D &b = (*l)[i];
    *nv = (uint32_t)(b.projected_vertices.size() / 4);
    *nt = (uint32_t)b.edges.size();
    *has_normals = b.normals.empty() ? 0 : 1;
    return 0;
}
int rxh_scene_batch3d_copy(void *s, int list, int chunk, uint32_t i, float *pv4, float *uv2, float *n3, uint32_t *idx3,
                           float *edges10, float *bbox5) {
    auto *l = list3d((Scene *)s, list, chunk);
    if (!l || i >= l->size()) return RXR_ERR_INVALID;
    const Batch3D &b = (*l)[i];
    const size_t nv = b.projected_vertices.size() / 4;
    memcpy(pv4, b.projected_vertices.data(), b.projected_vertices.size() * 4);
    memcpy(uv2, b.clipped_uvs.data(), std::min(b.clipped_uvs.size(), nv * 2) * 4);
    memcpy(n3, b.clipped_normals.data(), std::min(b.clipped_normals.size(), nv * 3) * 4);
    for (size_t k = 0; k < b.edges.size(); ++k) {
        for (int j = 0; j < 3; ++j) idx3[k * 3 + j] = b.clipped_indices[k * 3 + j];
        for (int j = 0; j < 3; ++j) {
            edges10[k * 10 + j] = b.edges[k].a[j];
            edges10[k * 10 + 3 + j] = b.edges[k].b[j];
            edges10[k * 10 + 6 + j] = b.edges[k].c[j];
        }
        edges10[k * 10 + 9] = b.edges[k].visible ? 1.0f : 0.0f;
    }
    bbox5[0] = b.has_bounding_box ? 1.0f : 0.0f;
    bbox5[1] = b.bounding_box.x; bbox5[2] = b.bounding_box.y; bbox5[3] = b.bounding_box.width; bbox5[4] = b.bounding_box.height;
    return 0;
}

// ---- cameras ----------------------------------------------------------------------------------------
void rxh_camera_orbit(const float *center3, float distance, float azimuth, float elevation, float fov, float near, float far,
                      float w, float h, float *view16, float *proj16) {
    Mat4 v, p;
    orbit_camera(Vec3{center3[0], center3[1], center3[2]}, distance, azimuth, elevation, fov, near, far, w, h, v, p);
    memcpy(view16, v.m, 64);
    memcpy(proj16, p.m, 64);
}
void rxh_camera_firstp(const float *pos3, const float *center3, float fov, float near, float far, float w, float h,
                       float *view16, float *proj16) {
    Mat4 v, p;
    firstp_camera(Vec3{pos3[0], pos3[1], pos3[2]}, Vec3{center3[0], center3[1], center3[2]}, fov, near, far, w, h, v, p);
    memcpy(view16, v.m, 64);
    memcpy(proj16, p.m, 64);
}

}  // extern "C"
