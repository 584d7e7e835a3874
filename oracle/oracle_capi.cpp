// oracle_capi.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// extern "C" builder API over the oracle so that tests/ can drive it with ctypes.  The function
// set deliberately has the same shape as the product host library's `rxh_*` API
// (rusterix_amd/csrc/host/host_capi.cpp) so one Python scene description can be replayed on both.
#include <array>
#include <cstdio>
#include <cstring>
#include <thread>

#include "rusterix_oracle.hpp"

using namespace orc;

namespace {
Mat4 mat4_from(const float *m) {
    Mat4 o{};
    memcpy(o.m, m, sizeof(o.m));
    return o;
}
Mat3 mat3_from(const float *m) {
    Mat3 o{};
    memcpy(o.m, m, sizeof(o.m));
    return o;
}
Tile make_tile(const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    Tile t;
    for (uint32_t i = 0; i < n; ++i) {
        Texture tex;
        tex.width = ws[i];
        tex.height = hs[i];
        tex.data.assign(frames[i], frames[i] + (size_t)ws[i] * hs[i] * 4);
        t.textures.push_back(std::move(tex));
    }
    return t;
}
struct RasterizerBox {
    Rasterizer r;
    int n_threads = 1;
};
std::vector<Batch3D> *list3d(Scene *s, int list, int chunk) {
    switch (list) {
        case RXR_LIST_CHUNK_OPACITY: return (chunk >= 0 && (size_t)chunk < s->chunks.size()) ? &s->chunks[chunk].batches3d_opacity : nullptr;
        case RXR_LIST_CHUNK: return (chunk >= 0 && (size_t)chunk < s->chunks.size()) ? &s->chunks[chunk].batches3d : nullptr;
        case RXR_LIST_CHUNK_TERRAIN: return (chunk >= 0 && (size_t)chunk < s->chunks.size()) ? &s->chunks[chunk].terrain_batch3d : nullptr;
        case RXR_LIST_STATIC: return &s->d3_static;
        case RXR_LIST_DYNAMIC: return &s->d3_dynamic;
        case RXR_LIST_OVERLAY: return &s->d3_overlay;
    }
    return nullptr;
}
}  // namespace

extern "C" {

// ---- scene ----------------------------------------------------------------------------------------
void *orc_scene_new() { return new Scene(); }
void orc_scene_free(void *s) { delete (Scene *)s; }
void orc_scene_set_animation_frame(void *s, uint64_t f) { ((Scene *)s)->animation_frame = (size_t)f; }
void orc_scene_set_background(void *s, int kind) { ((Scene *)s)->background = kind; }
void orc_scene_set_background_grid(void *s, float grid_size, float subdivisions, float offset_x, float offset_y) {
    float *g = ((Scene *)s)->background_grid;
    g[0] = grid_size;
    g[1] = subdivisions;
    g[2] = offset_x;
    g[3] = offset_y;
}
void orc_scene_add_light(void *s, const rxr_light *l, int dynamic) {
    if (dynamic) ((Scene *)s)->dynamic_lights.push_back(*l);
    else ((Scene *)s)->lights.push_back(*l);
}
void orc_scene_add_dynamic_tile(void *s, const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    ((Scene *)s)->dynamic_textures.push_back(make_tile(frames, ws, hs, n));
}
int orc_scene_add_chunk(void *s) {
    ((Scene *)s)->chunks.emplace_back();
    return (int)((Scene *)s)->chunks.size() - 1;
}
void orc_chunk_add_occluder(void *s, int chunk, float minx, float miny, float maxx, float maxy, float occ) {
    ((Scene *)s)->chunks[chunk].occluded_sectors.push_back(Occluder{{minx, miny}, {maxx, maxy}, occ});
}
// chunk.terrain_texture / origin / size (src/chunk.rs:25-36); rgba == NULL: no texture
void orc_chunk_set_terrain(void *s, int chunk, const uint8_t *rgba, uint32_t w, uint32_t h, int ox, int oy, int size) {
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
void orc_chunk_set_terrain_batch2d(void *s, int chunk, void *b) {
    Chunk &c = ((Scene *)s)->chunks[chunk];
    c.terrain_batch2d.clear();
    c.terrain_batch2d.push_back(*(Batch2D *)b);
}
// chunk.shader_textures.push(texture) (src/chunk.rs:129); rgba == NULL pushes None
void orc_chunk_add_shader_texture(void *s, int chunk, const uint8_t *rgba, uint32_t w, uint32_t h) {
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
void orc_chunk_add_light(void *s, int chunk, const rxr_light *l) { ((Scene *)s)->chunks[chunk].lights.push_back(*l); }
// scene.add_shader (src/scene.rs:104-134) minus the parser / compiler: the program arrives as NodeOp trees
// in the word serialisation of include/rxr.h.  chunk < 0: scene.shaders, else that chunk's shaders.
int orc_scene_add_program(void *s, int chunk, uint32_t n_globals, int32_t shade_index, uint32_t shade_locals,
                          const uint32_t *const *fn_words, const uint32_t *fn_lens, uint32_t n_functions) {
    Scene *sc = (Scene *)s;
    vm::Program p;
    p.globals = n_globals;
    p.shade_index = shade_index;
    p.shade_locals = shade_locals;
    for (uint32_t i = 0; i < n_functions; ++i) {
        std::vector<vm::NodeOp> code;
        if (!vm::parse_block(fn_words[i], fn_lens[i], code)) return RXR_ERR_INVALID;
        p.user_functions.push_back(std::move(code));
    }
    std::vector<vm::Program> *dst = &sc->shaders;
    if (chunk >= 0) {
        if ((size_t)chunk >= sc->chunks.size()) return RXR_ERR_INVALID;
        dst = &sc->chunks[chunk].shaders;
    }
    dst->push_back(std::move(p));
    return (int)dst->size() - 1;
}
uint32_t orc_scene_num_dynamic_lights(void *s) { return (uint32_t)((Scene *)s)->dynamic_lights.size(); }

// ---- Batch3D --------------------------------------------------------------------------------------
void *orc_batch3d_new(const float *v4, uint32_t nv, const uint32_t *idx, uint32_t nt, const float *uv2) {
    Batch3D *b = new Batch3D();
    batch3d_add(*b, v4, nv, idx, nt, uv2);
    return b;
}
void *orc_batch3d_from_box(float x, float y, float z, float w, float h, float d) { return new Batch3D(batch3d_from_box(x, y, z, w, h, d)); }
void *orc_batch3d_from_obj(const char *text) { return new Batch3D(batch3d_from_obj(text)); }
void orc_batch3d_free(void *b) { delete (Batch3D *)b; }
void orc_batch3d_add(void *b, const float *v4, uint32_t nv, const uint32_t *idx, uint32_t nt, const float *uv2) {
    batch3d_add(*(Batch3D *)b, v4, nv, idx, nt, uv2);
}
void orc_batch3d_set_normals(void *b, const float *n3, uint32_t n) {
    auto &v = ((Batch3D *)b)->normals;
    v.clear();
    for (uint32_t i = 0; i < n; ++i) v.push_back(Vec3{n3[i * 3], n3[i * 3 + 1], n3[i * 3 + 2]});
}
void orc_batch3d_compute_vertex_normals(void *b) { batch3d_compute_vertex_normals(*(Batch3D *)b); }
void orc_batch3d_set_source(void *b, uint32_t kind, uint32_t index, const uint8_t *pixel) {
    Source &s = ((Batch3D *)b)->source;
    s.kind = kind;
    s.index = index;
    if (pixel) memcpy(s.pixel, pixel, 4);
}
// PixelSource::EntityTile(id, seq) / ItemTile(id, seq)
void orc_batch3d_set_source_seq(void *b, int is_item, uint32_t id, uint32_t seq) {
    Source &s = ((Batch3D *)b)->source;
    s.kind = is_item ? RXR_HOST_SOURCE_ITEM_TILE : RXR_HOST_SOURCE_ENTITY_TILE;
    s.index = id;
    s.seq = seq;
}
void orc_batch3d_set_repeat_mode(void *b, int m) { ((Batch3D *)b)->repeat_mode = m; }
void orc_batch3d_set_cull_mode(void *b, int m) { ((Batch3D *)b)->cull_mode = m; }
void orc_batch3d_set_ambient_color(void *b, float r, float g, float bl) { ((Batch3D *)b)->ambient_color = Vec3{r, g, bl}; }
void orc_batch3d_set_transform(void *b, const float *m16) { ((Batch3D *)b)->transform_3d = mat4_from(m16); }
void orc_batch3d_set_profile_id(void *b, int has, uint32_t id) {
    ((Batch3D *)b)->has_profile_id = has != 0;
    ((Batch3D *)b)->profile_id = id;
}
void orc_batch3d_set_shader(void *b, int shader) { ((Batch3D *)b)->shader = shader; }
void orc_batch3d_counts(void *b, uint32_t *nv, uint32_t *nt) {
    *nv = (uint32_t)((Batch3D *)b)->vertices.size();
    *nt = (uint32_t)((Batch3D *)b)->indices.size();
}
void orc_batch3d_get_geometry(void *b, float *v4, uint32_t *idx, float *uv2, float *n3) {
    Batch3D *p = (Batch3D *)b;
    for (size_t i = 0; i < p->vertices.size(); ++i) memcpy(v4 + i * 4, p->vertices[i].data(), 16);
    for (size_t i = 0; i < p->indices.size(); ++i)
        for (int k = 0; k < 3; ++k) idx[i * 3 + k] = (uint32_t)p->indices[i][k];
    for (size_t i = 0; i < p->uvs.size(); ++i) memcpy(uv2 + i * 2, p->uvs[i].data(), 8);
    if (n3)
        for (size_t i = 0; i < p->normals.size(); ++i) {
            n3[i * 3] = p->normals[i].x; n3[i * 3 + 1] = p->normals[i].y; n3[i * 3 + 2] = p->normals[i].z;
        }
}
uint32_t orc_batch3d_num_normals(void *b) { return (uint32_t)((Batch3D *)b)->normals.size(); }
// copies the batch into the scene list
int orc_scene_push_batch3d(void *s, void *b, int list, int chunk) {
    auto *l = list3d((Scene *)s, list, chunk);
    if (!l) return RXR_ERR_INVALID;
    if (list == RXR_LIST_CHUNK_TERRAIN) l->clear();  // Option<Batch3D>: the latest one wins
    l->push_back(*(Batch3D *)b);
    return 0;
}

// ---- Batch2D --------------------------------------------------------------------------------------
void *orc_batch2d_new(const float *v2, uint32_t nv, const uint32_t *idx, uint32_t nt, const float *uv2) {
    Batch2D *b = new Batch2D();
    for (uint32_t i = 0; i < nv; ++i) b->vertices.push_back({v2[i * 2], v2[i * 2 + 1]});
    for (uint32_t i = 0; i < nv; ++i) b->uvs.push_back({uv2[i * 2], uv2[i * 2 + 1]});
    for (uint32_t i = 0; i < nt; ++i) b->indices.push_back({idx[i * 3], idx[i * 3 + 1], idx[i * 3 + 2]});
    return b;
}
void *orc_batch2d_from_rectangle(float x, float y, float w, float h) { return new Batch2D(batch2d_from_rectangle(x, y, w, h)); }
void orc_batch2d_free(void *b) { delete (Batch2D *)b; }
void orc_batch2d_set_mode(void *b, int m) { ((Batch2D *)b)->mode = m; }
void orc_batch2d_set_repeat_mode(void *b, int m) { ((Batch2D *)b)->repeat_mode = m; }
void orc_batch2d_set_source(void *b, uint32_t kind, uint32_t index, const uint8_t *pixel) {
    Source &s = ((Batch2D *)b)->source;
    s.kind = kind;
    s.index = index;
    if (pixel) memcpy(s.pixel, pixel, 4);
}
void orc_batch2d_set_source_seq(void *b, int is_item, uint32_t id, uint32_t seq) {
    Source &s = ((Batch2D *)b)->source;
    s.kind = is_item ? RXR_HOST_SOURCE_ITEM_TILE : RXR_HOST_SOURCE_ENTITY_TILE;
    s.index = id;
    s.seq = seq;
}
void orc_batch2d_set_receives_light(void *b, int v) { ((Batch2D *)b)->receives_light = v != 0; }
void orc_batch2d_set_shader(void *b, int shader) { ((Batch2D *)b)->shader = shader; }
int orc_scene_push_batch2d(void *s, void *b, int dynamic, int chunk) {
    Scene *sc = (Scene *)s;
    if (chunk >= 0) {
        if ((size_t)chunk >= sc->chunks.size()) return RXR_ERR_INVALID;
        sc->chunks[chunk].batches2d.push_back(*(Batch2D *)b);
    } else if (dynamic) {
        sc->d2_dynamic.push_back(*(Batch2D *)b);
    } else {
        sc->d2_static.push_back(*(Batch2D *)b);
    }
    return 0;
}

// ---- Assets ---------------------------------------------------------------------------------------
void *orc_assets_new() { return new Assets(); }
void orc_assets_free(void *a) { delete (Assets *)a; }
void orc_assets_add_tile(void *a, const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    ((Assets *)a)->tile_list.push_back(make_tile(frames, ws, hs, n));
}
// assets.entity_tiles / item_tiles: makes `id` known (an entry without sequences) ...
void orc_assets_add_sequence_id(void *a, int is_item, uint32_t id) { (is_item ? ((Assets *)a)->item_tiles : ((Assets *)a)->entity_tiles)[id]; }
// ... and appends one sequence tile to it (IndexMap insertion order = get_index order)
void orc_assets_add_sequence_tile(void *a, int is_item, uint32_t id, const uint8_t *const *frames, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    (is_item ? ((Assets *)a)->item_tiles : ((Assets *)a)->entity_tiles)[id].push_back(make_tile(frames, ws, hs, n));
}

// rusteria's global pattern banks (textures/patterns.rs) and assets.palette, as data
void orc_assets_set_patterns(void *a, int normal, const float *const *rgb, const uint32_t *ws, const uint32_t *hs, uint32_t n) {
    std::vector<vm::TexStorage> &dst = normal ? ((Assets *)a)->vm_env.patterns_normal : ((Assets *)a)->vm_env.patterns;
    dst.clear();
    for (uint32_t i = 0; i < n; ++i) {
        vm::TexStorage t;
        t.width = ws[i];
        t.height = hs[i];
        t.data.resize((size_t)ws[i] * hs[i]);
        for (size_t k = 0; k < t.data.size(); ++k) t.data[k] = Vec3{rgb[i][3 * k], rgb[i][3 * k + 1], rgb[i][3 * k + 2]};
        dst.push_back(std::move(t));
    }
}
void orc_assets_set_palette(void *a, const float *rgb3, const uint8_t *present, uint32_t n) {
    vm::Env &e = ((Assets *)a)->vm_env;
    e.palette_rgb.clear();
    e.palette_present.clear();
    for (uint32_t i = 0; i < n; ++i) {
        e.palette_rgb.push_back(Vec3{rgb3[3 * i], rgb3[3 * i + 1], rgb3[3 * i + 2]});
        e.palette_present.push_back(present ? present[i] : 1);
    }
}
static const char *g_last_vm_fault = "";
const char *orc_last_vm_fault() { return g_last_vm_fault; }
// one invocation of Execution::shade on a fresh Execution with the given inputs (tests/test_oracle_vm.py):
// in/out = uv, color, roughness, metallic, emissive, opacity, bump, normal, hitpoint, time (10 x 3 floats)
int orc_vm_shade(void *s, void *a, int program, float *fields30) {
    Scene *sc = (Scene *)s;
    if (program < 0 || (size_t)program >= sc->shaders.size()) return RXR_ERR_INVALID;
    const vm::Program &p = sc->shaders[program];
    if (p.shade_index < 0) return RXR_ERR_INVALID;
    vm::Execution ex(0);
    Vec3 *f[10] = {&ex.uv, &ex.color, &ex.roughness, &ex.metallic, &ex.emissive, &ex.opacity, &ex.bump, &ex.normal, &ex.hitpoint, &ex.time};
    for (int i = 0; i < 10; ++i) *f[i] = Vec3{fields30[3 * i], fields30[3 * i + 1], fields30[3 * i + 2]};
    ex.reset(p.globals);
    try {
        ex.shade((size_t)p.shade_index, p, ((Assets *)a)->vm_env);
    } catch (const vm::Fault &f) {
        g_last_vm_fault = f.what;
        return RXR_ERR_INVALID;
    }
    for (int i = 0; i < 10; ++i) {
        fields30[3 * i] = f[i]->x;
        fields30[3 * i + 1] = f[i]->y;
        fields30[3 * i + 2] = f[i]->z;
    }
    return 0;
}

// ---- Rasterizer -----------------------------------------------------------------------------------
void *orc_rasterizer_setup(const float *m2d9, const float *view16, const float *proj16) {
    RasterizerBox *rb = new RasterizerBox();
    Mat3 m2d{};
    if (m2d9) m2d = mat3_from(m2d9);
    rb->r = rasterizer_setup(m2d9 ? &m2d : nullptr, mat4_from(view16), mat4_from(proj16));
    rb->n_threads = (int)std::thread::hardware_concurrency();
    if (rb->n_threads < 1) rb->n_threads = 1;
    return rb;
}
void orc_rasterizer_free(void *r) { delete (RasterizerBox *)r; }
void orc_rasterizer_set_threads(void *r, int n) { ((RasterizerBox *)r)->n_threads = n < 1 ? 1 : n; }
int orc_rasterizer_get_threads(void *r) { return ((RasterizerBox *)r)->n_threads; }
void orc_rasterizer_render_mode(void *r, int d2, int d3, int ignore_bg) {
    Rasterizer &x = ((RasterizerBox *)r)->r;
    x.d2_active = d2 != 0;
    x.d3_active = d3 != 0;
    x.ignore_background_shader = ignore_bg != 0;
}
void orc_rasterizer_sample_mode(void *r, int m) { ((RasterizerBox *)r)->r.sample_mode = m; }
void orc_rasterizer_brush_preview(void *r, int on, float px, float py, float pz, float radius, float falloff) {
    Rasterizer &x = *(Rasterizer *)r;
    x.has_brush_preview = on != 0;
    x.brush_position = Vec3{px, py, pz};
    x.brush_radius = radius;
    x.brush_falloff = falloff;
}
void orc_rasterizer_background(void *r, const uint8_t *px) {
    Rasterizer &x = ((RasterizerBox *)r)->r;
    x.has_background_color = px != nullptr;
    if (px) memcpy(x.background_color, px, 4);
}
void orc_rasterizer_ambient(void *r, const float *a4) {
    Rasterizer &x = ((RasterizerBox *)r)->r;
    x.has_ambient = a4 != nullptr;
    if (a4) x.ambient_color = Vec4{a4[0], a4[1], a4[2], a4[3]};
}
void orc_rasterizer_time(void *r, float t) { ((RasterizerBox *)r)->r.time = t; }
void orc_rasterizer_preserve_transparency(void *r, int v) { ((RasterizerBox *)r)->r.preserve_transparency = v != 0; }
void orc_rasterizer_sun(void *r, const float *dir3, float day_factor) {
    Rasterizer &x = ((RasterizerBox *)r)->r;
    x.has_sun = dir3 != nullptr;
    if (dir3) x.sun_dir = Vec3{dir3[0], dir3[1], dir3[2]};
    x.day_factor = day_factor;
}
void orc_rasterizer_mapmini_add_occluder(void *r, float minx, float miny, float maxx, float maxy, float occ) {
    ((RasterizerBox *)r)->r.mapmini.occluded_sectors.push_back(Occluder{{minx, miny}, {maxx, maxy}, occ});
}
void orc_rasterizer_mapmini_add_linedef(void *r, float x0, float y0, float x1, float y1) {
    ((RasterizerBox *)r)->r.mapmini.linedefs.push_back(Linedef{{x0, y0}, {x1, y1}});
}
void orc_rasterizer_get_derived(void *r, float *inv_view16, float *inv_proj16, float *camera_pos3) {
    Rasterizer &x = ((RasterizerBox *)r)->r;
    memcpy(inv_view16, x.inverse_view_matrix.m, 64);
    memcpy(inv_proj16, x.inverse_projection_matrix.m, 64);
    camera_pos3[0] = x.camera_pos.x; camera_pos3[1] = x.camera_pos.y; camera_pos3[2] = x.camera_pos.z;
}
int orc_rasterizer_rasterize(void *r, void *scene, uint8_t *pixels, uint32_t w, uint32_t h, uint32_t tile_size, void *assets) {
    RasterizerBox *rb = (RasterizerBox *)r;
    return rasterize(rb->r, *(Scene *)scene, pixels, w, h, tile_size, *(Assets *)assets, rb->n_threads);
}

// host-side projection only (Scene::project with the rasterizer's matrices)
int orc_scene_project(void *r, void *scene, uint32_t w, uint32_t h) {
    Rasterizer &x = ((RasterizerBox *)r)->r;
    return scene_project(*(Scene *)scene, x.has_m2d ? &x.projection_matrix_2d : nullptr, x.view_matrix, x.projection_matrix, (float)w, (float)h)
               ? 0
               : RXR_ERR_INVALID;
}

// ---- introspection of projected batches (after rasterize) --------------------------------------------
int orc_scene_batch3d_counts(void *s, int list, int chunk, uint32_t i, uint32_t *nv, uint32_t *nt, uint32_t *has_normals) {
    auto *l = list3d((Scene *)s, list, chunk);
    if (!l || i >= l->size()) return RXR_ERR_INVALID;
    const Batch3D &b = (*l)[i];
    *nv = (uint32_t)b.projected_vertices.size();
    *nt = (uint32_t)b.edges.size();
    *has_normals = b.normals.empty() ? 0 : 1;
    return 0;
}
// edges10: a[3] b[3] c[3] visible(0/1) as float; bbox5: has,x,y,w,h
int orc_scene_batch3d_copy(void *s, int list, int chunk, uint32_t i, float *pv4, float *uv2, float *n3, uint32_t *idx3,
                           float *edges10, float *bbox5) {
    auto *l = list3d((Scene *)s, list, chunk);
    if (!l || i >= l->size()) return RXR_ERR_INVALID;
    const Batch3D &b = (*l)[i];
    for (size_t k = 0; k < b.projected_vertices.size(); ++k) memcpy(pv4 + k * 4, b.projected_vertices[k].data(), 16);
    for (size_t k = 0; k < b.clipped_uvs.size() && k < b.projected_vertices.size(); ++k) memcpy(uv2 + k * 2, b.clipped_uvs[k].data(), 8);
    for (size_t k = 0; k < b.clipped_normals.size() && k < b.projected_vertices.size(); ++k) {
        n3[k * 3] = b.clipped_normals[k].x; n3[k * 3 + 1] = b.clipped_normals[k].y; n3[k * 3 + 2] = b.clipped_normals[k].z;
    }
    for (size_t k = 0; k < b.edges.size(); ++k) {
        for (int j = 0; j < 3; ++j) idx3[k * 3 + j] = (uint32_t)b.clipped_indices[k][j];
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
void orc_camera_orbit(const float *center3, float distance, float azimuth, float elevation, float fov, float near, float far,
                      float w, float h, float *view16, float *proj16) {
    Mat4 v, p;
    orbit_camera(Vec3{center3[0], center3[1], center3[2]}, distance, azimuth, elevation, fov, near, far, w, h, v, p);
    memcpy(view16, v.m, 64);
    memcpy(proj16, p.m, 64);
}
void orc_camera_firstp(const float *pos3, const float *center3, float fov, float near, float far, float w, float h,
                       float *view16, float *proj16) {
    Mat4 v, p;
    firstp_camera(Vec3{pos3[0], pos3[1], pos3[2]}, Vec3{center3[0], center3[1], center3[2]}, fov, near, far, w, h, v, p);
    memcpy(view16, v.m, 64);
    memcpy(proj16, p.m, 64);
}

// ---- leaf functions for the known-answer tests -------------------------------------------------------
uint32_t orc_hash_u32(uint32_t seed) { return hash_u32(seed); }
void orc_pixel_to_vec4(const uint8_t *p, float *out) { pixel_to_vec4(p, out); }
void orc_vec4_to_pixel(const float *v, uint8_t *out) { vec4_to_pixel(v, out); }
float orc_srgb_to_linear_fast(float x) { return srgb_to_linear_fast(x); }
float orc_linear_to_srgb_fast(float x) { return linear_to_srgb_fast(x); }
void orc_texture_sample(const uint8_t *rgba, uint32_t w, uint32_t h, float u, float v, int sample_mode, int repeat_mode, uint8_t *out) {
    Texture t;
    t.width = w;
    t.height = h;
    t.data.assign(rgba, rgba + (size_t)w * h * 4);
    texture_sample(t, u, v, sample_mode, repeat_mode, out);
}
int orc_light_color_at(const rxr_light *l, const float *p3, uint32_t hash, int d2, float *out3) {
    return light_color_at(*l, Vec3{p3[0], p3[1], p3[2]}, hash, d2 != 0, out3) ? 1 : 0;
}
int orc_light_radiance_at(const rxr_light *l, const float *p3, const float *n3, uint32_t hash, float *out3) {
    Vec3 o;
    bool ok = light_radiance_at(*l, Vec3{p3[0], p3[1], p3[2]}, n3 != nullptr, n3 ? Vec3{n3[0], n3[1], n3[2]} : Vec3{}, hash, o);
    out3[0] = o.x; out3[1] = o.y; out3[2] = o.z;
    return ok ? 1 : 0;
}
void orc_edges_new(const float *v0xy3, const float *v1xy3, float *abc9) {
    float a[3][2], b[3][2];
    memcpy(a, v0xy3, sizeof(a));
    memcpy(b, v1xy3, sizeof(b));
    Edges e = edges_new(a, b, true);
    memcpy(abc9, e.a, 12);
    memcpy(abc9 + 3, e.b, 12);
    memcpy(abc9 + 6, e.c, 12);
}
int orc_edges_evaluate(const float *abc9, float px, float py) {
    Edges e{};
    memcpy(e.a, abc9, 12);
    memcpy(e.b, abc9 + 3, 12);
    memcpy(e.c, abc9 + 6, 12);
    e.visible = true;
    float p[2] = {px, py};
    return edges_evaluate(e, p) ? 1 : 0;
}
void orc_mat4_inverted(const float *m16, float *out16) {
    Mat4 r = rvek::inverted(mat4_from(m16));
    memcpy(out16, r.m, 64);
}
void orc_mat4_mul_vec4(const float *m16, const float *v4, float *out4) {
    Vec4 r = mat4_from(m16) * Vec4{v4[0], v4[1], v4[2], v4[3]};
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
const char *orc_build_info() { return "rusterix oracle: C++ restatement of the reference algorithm; vek matvec fused="
#if RXR_VEK_FUSED_MATVEC
    "1";
#else
    "0";
#endif
}

}  // extern "C"
