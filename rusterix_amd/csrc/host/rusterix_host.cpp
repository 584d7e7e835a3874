// rusterix_host.cpp -- see rusterix_host.hpp.  Build with -ffp-contract=off (Rust never fuses).
#include "rusterix_host.hpp"
#include "../rxr_parallel.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace rusterix {

void *PinnedPool::take(size_t bytes, bool &pinned) {
    pinned = false;
    if (bytes >= threshold) {
        const char *pa = getenv("RXR_PINNED_ARRAYS");  // (0: ordinary memory -- tests and A-B runs)
        if (!(pa && pa[0] == '0'))
            if (void *p = rxr_alloc_pinned(bytes)) {
                pinned = true;
                return p;
            }
        any_unpinned() = true;
    }
    return malloc(bytes);
}
void PinnedPool::give(void *p, bool pinned) {
    if (pinned) rxr_free_pinned(p);
    else free(p);
}


namespace {

// f32::min / f32::max return the non-NaN operand
inline float fmin_rs(float a, float b) { return std::fmin(a, b); }
inline float fmax_rs(float a, float b) { return std::fmax(a, b); }

// Edges::new, src/edge.rs:12-24: p[i] -> q[i] for the three directed edges
inline rxr_edges make_edges(const float p[3][2], const float q[3][2], bool visible) {
    rxr_edges e{};
    for (int i = 0; i < 3; ++i) {
        e.a[i] = q[i][1] - p[i][1];
        e.b[i] = p[i][0] - q[i][0];
        e.c[i] = q[i][0] * p[i][1] - q[i][1] * p[i][0];
    }
    e.visible = visible ? 1u : 0u;
    return e;
}

inline rxr_edges triangle_edges(const float *v0, const float *v1, const float *v2, bool visible) {
    const float p[3][2] = {{v0[0], v0[1]}, {v1[0], v1[1]}, {v2[0], v2[1]}};
    const float q[3][2] = {{v1[0], v1[1]}, {v2[0], v2[1]}, {v0[0], v0[1]}};
    return make_edges(p, q, visible);
}

// src/batch/batch3d.rs:742-746
inline bool front_facing(const float *v0, const float *v1, const float *v2) {
    float orientation = (v1[0] - v0[0]) * (v2[1] - v0[1]) - (v1[1] - v0[1]) * (v2[0] - v0[0]);
    return orientation > 0.0f;
}

struct ClipVertex {
    float pos[4];
    float uv[2];
    Vec3 n;
};

}  // namespace

uint64_t next_generation() {
    static std::atomic<uint64_t> g{1};
    return g.fetch_add(1);
}

uint32_t hash_u32(uint32_t seed) {
    uint32_t state = seed;
    state = (state ^ 61u) ^ (state >> 16);
    state += state << 3;
    state ^= state >> 4;
    state *= 0x27d4eb2du;
    state ^= state >> 15;
    return state;
}

// ---- Batch3D ------------------------------------------------------------------------------------
Batch3D Batch3D::make(const float *verts4, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2) {
    Batch3D b;
    b.add(verts4, nv, idx3, nt, uvs2);
    return b;
}

void Batch3D::add(const float *verts4, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2) {
    const uint32_t base_index = (uint32_t)vertex_count();
    vertices.insert(vertices.end(), verts4, verts4 + nv * 4);
    uvs.insert(uvs.end(), uvs2, uvs2 + nv * 2);
    indices.reserve(indices.size() + nt * 3);
    for (size_t i = 0; i < nt * 3; ++i) indices.push_back(idx3[i] + base_index);
    touch();
}

Batch3D Batch3D::from_box(float x, float y, float z, float w, float h, float d) {
    const float X = x + w, Y = y + h, Z = z + d;
    // face order and winding as src/batch/batch3d.rs:141-193
    const float v[24][4] = {
        {x, y, z, 1}, {X, y, z, 1}, {X, Y, z, 1}, {x, Y, z, 1},  // front
        {x, y, Z, 1}, {X, y, Z, 1}, {X, Y, Z, 1}, {x, Y, Z, 1},  // back
        {x, y, z, 1}, {x, Y, z, 1}, {x, Y, Z, 1}, {x, y, Z, 1},  // left
        {X, y, z, 1}, {X, Y, z, 1}, {X, Y, Z, 1}, {X, y, Z, 1},  // right
        {x, Y, z, 1}, {X, Y, z, 1}, {X, Y, Z, 1}, {x, Y, Z, 1},  // top
        {x, y, z, 1}, {X, y, z, 1}, {X, y, Z, 1}, {x, y, Z, 1},  // bottom
    };
    const uint32_t idx[36] = {0, 1, 2, 0, 2, 3, 4, 6, 5, 4, 7, 6, 8, 9, 10, 8, 10, 11, 12, 14, 13, 12, 15, 14,
                              16, 17, 18, 16, 18, 19, 20, 23, 22, 20, 22, 21};
    float uv[24][2];
    for (int f = 0; f < 6; ++f) {
        const float q[4][2] = {{0, 1}, {1, 1}, {1, 0}, {0, 0}};
        memcpy(uv[f * 4], q, sizeof(q));
    }
    return make(&v[0][0], 24, idx, 12, &uv[0][0]);
}

Batch3D Batch3D::from_obj(const std::string &text) {
    // src/wavefront.rs:34-102: `v`, `vt`, `f` (first three corners, index before the first '/')
    Batch3D b;
    std::vector<float> tc;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        size_t s = line.find_first_not_of(" \t\r");
        if (s == std::string::npos) continue;
        line = line.substr(s, line.find_last_not_of(" \t\r") - s + 1);
        if (line[0] == '#') continue;
        if (line.compare(0, 2, "v ") == 0) {
            float x = 0, y = 0, z = 0;
            sscanf(line.c_str() + 2, "%f %f %f", &x, &y, &z);
            const float v[4] = {x, y, z, 1.0f};
            b.vertices.insert(b.vertices.end(), v, v + 4);
        } else if (line.compare(0, 3, "vt ") == 0) {
            float u = 0, v = 0;
            sscanf(line.c_str() + 3, "%f %f", &u, &v);
            tc.push_back(u);
            tc.push_back(v);
        } else if (line.compare(0, 2, "f ") == 0) {
            char c0[64], c1[64], c2[64];
            if (sscanf(line.c_str() + 2, "%63s %63s %63s", c0, c1, c2) == 3) {
                b.indices.push_back((uint32_t)strtoul(c0, nullptr, 10) - 1u);
                b.indices.push_back((uint32_t)strtoul(c1, nullptr, 10) - 1u);
                b.indices.push_back((uint32_t)strtoul(c2, nullptr, 10) - 1u);
            }
        }
    }
    if (tc.empty()) {
        for (size_t i = 0; i < b.vertex_count(); ++i) {  // uv = (x, y), :92-95
            b.uvs.push_back(b.vertices[i * 4]);
            b.uvs.push_back(b.vertices[i * 4 + 1]);
        }
    } else {
        b.uvs = tc;
    }
    return b;
}

void Batch3D::compute_vertex_normals() {
    const size_t nv = vertex_count();
    std::vector<Vec3> acc(nv);
    std::vector<uint32_t> counts(nv, 0);
    for (size_t t = 0; t < triangle_count(); ++t) {
        const uint32_t i0 = indices[3 * t], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
        Vec3 p0{vertices[4 * i0], vertices[4 * i0 + 1], vertices[4 * i0 + 2]};
        Vec3 p1{vertices[4 * i1], vertices[4 * i1 + 1], vertices[4 * i1 + 2]};
        Vec3 p2{vertices[4 * i2], vertices[4 * i2 + 1], vertices[4 * i2 + 2]};
        Vec3 fn = rvek::normalized(rvek::cross(p1 - p0, p2 - p0));
        acc[i0] += fn;
        acc[i1] += fn;
        acc[i2] += fn;
        ++counts[i0];
        ++counts[i1];
        ++counts[i2];
    }
    normals.assign(nv * 3, 0.0f);
    for (size_t i = 0; i < nv; ++i) {
        Vec3 n = acc[i];
        if (counts[i] > 0) {
            n = n / (float)counts[i];
            n = rvek::normalized(n);
        }
        normals[3 * i] = n.x;
        normals[3 * i + 1] = n.y;
        normals[3 * i + 2] = n.z;
    }
    touch();
}

bool Batch3D::clip_and_project(const Mat4 &view_matrix, const Mat4 &projection_matrix, float viewport_width,
                               float viewport_height) {
    const size_t nv = vertex_count(), nt = triangle_count();
    const Mat4 mvp = (projection_matrix * view_matrix) * transform_3d;

    if (nv > 0) {  // object-space AABB against the clip planes, :493-552
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < nv; ++i)
            for (int k = 0; k < 3; ++k) {
                lo[k] = fmin_rs(lo[k], vertices[4 * i + k]);
                hi[k] = fmax_rs(hi[k], vertices[4 * i + k]);
            }
        bool out_l = true, out_r = true, out_b = true, out_t = true, out_n = true, out_f = true;
        for (int c = 0; c < 8; ++c) {
            Vec4 corner{(c & 4) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 1) ? hi[2] : lo[2], 1.0f};
            Vec4 v = mvp * corner;
            const float w = v.w;
            out_l &= v.x < -w;
            out_r &= v.x > w;
            out_b &= v.y < -w;
            out_t &= v.y > w;
            out_n &= v.z < -w;
            out_f &= v.z > w;
        }
        if (out_l || out_r || out_b || out_t || out_n || out_f) {
            projected_vertices.clear();
            clipped_indices.clear();
            clipped_uvs.clear();
            clipped_normals.clear();
            edges.clear();
            edge_visible.clear();
            has_bounding_box = false;
            return true;
        }
    }

    const Mat4 view_model = view_matrix * transform_3d;
    std::vector<float> vs(nv * 4);  // view space
    for (size_t i = 0; i < nv; ++i) {
        Vec4 r = view_model * Vec4{vertices[4 * i], vertices[4 * i + 1], vertices[4 * i + 2], vertices[4 * i + 3]};
        vs[4 * i] = r.x; vs[4 * i + 1] = r.y; vs[4 * i + 2] = r.z; vs[4 * i + 3] = r.w;
    }

    const float near_plane = 0.1f;
    clipped_indices.assign(indices.begin(), indices.end());
    clipped_uvs.assign(uvs.begin(), uvs.end());
    clipped_normals.assign(normals.begin(), normals.end());
    std::vector<uint8_t> edge_visibility(nt, 1);
    std::vector<ClipVertex> fresh;  // vertices created by clipping, appended after the originals

    for (size_t t = 0; t < nt; ++t) {
        const uint32_t ix[3] = {indices[3 * t], indices[3 * t + 1], indices[3 * t + 2]};
        const float *v[3] = {&vs[4 * ix[0]], &vs[4 * ix[1]], &vs[4 * ix[2]]};

        if (cull_mode_ != CullMode::Off) {  // :592-600
            float orient = (v[1][0] - v[0][0]) * (v[2][1] - v[0][1]) - (v[1][1] - v[0][1]) * (v[2][0] - v[0][0]);
            bool is_front = orient > 0.0f;
            if (cull_mode_ == CullMode::Back && is_front) continue;
            if (cull_mode_ == CullMode::Front && !is_front) continue;
        }
        // the reference indexes self.normals[i] unconditionally here (:605-607)
        if (normals.size() / 3 <= ix[0] || normals.size() / 3 <= ix[1] || normals.size() / 3 <= ix[2]) return false;

        const bool inside[3] = {v[0][2] < -near_plane, v[1][2] < -near_plane, v[2][2] < -near_plane};
        if (inside[0] && inside[1] && inside[2]) continue;
        edge_visibility[t] = 0;
        if (!inside[0] && !inside[1] && !inside[2]) continue;

        // Sutherland-Hodgman against z = -near_plane, :626-669
        uint32_t poly[4];
        int np = 0;
        size_t emitted = 0;
        for (int i = 0; i < 3; ++i) {
            const int j = (i + 1) % 3;
            const float *cur = v[i], *nxt = v[j];
            const float *uvc = &uvs[2 * ix[i]], *uvn = &uvs[2 * ix[j]];
            Vec3 nc{normals[3 * ix[i]], normals[3 * ix[i] + 1], normals[3 * ix[i] + 2]};
            Vec3 nn{normals[3 * ix[j]], normals[3 * ix[j] + 1], normals[3 * ix[j] + 2]};
            if (cur[2] < -near_plane) {
                ClipVertex cv{{cur[0], cur[1], cur[2], cur[3]}, {uvc[0], uvc[1]}, nc};
                fresh.push_back(cv);
                poly[np++] = (uint32_t)(nv + fresh.size() - 1);
                ++emitted;
            }
            if ((cur[2] < -near_plane) != (nxt[2] < -near_plane)) {
                const float s = (-near_plane - cur[2]) / (nxt[2] - cur[2]);
                ClipVertex cv{};
                for (int k = 0; k < 4; ++k) cv.pos[k] = cur[k] + s * (nxt[k] - cur[k]);
                cv.uv[0] = uvc[0] + s * (uvn[0] - uvc[0]);
                cv.uv[1] = uvc[1] + s * (uvn[1] - uvc[1]);
                cv.n = rvek::normalized(nc * (1.0f - s) + nn * s);
                fresh.push_back(cv);
                poly[np++] = (uint32_t)(nv + fresh.size() - 1);
                ++emitted;
            }
        }
        for (int i = 1; i + 1 < np; ++i) {  // fan, :672-678
            clipped_indices.push_back(poly[0]);
            clipped_indices.push_back(poly[i]);
            clipped_indices.push_back(poly[i + 1]);
        }
        edge_visibility.insert(edge_visibility.end(), emitted, 1);  // one `true` per emitted vertex, :680
    }

    for (const ClipVertex &cv : fresh) {  // :684-686
        vs.insert(vs.end(), cv.pos, cv.pos + 4);
        clipped_uvs.push_back(cv.uv[0]);
        clipped_uvs.push_back(cv.uv[1]);
        clipped_normals.push_back(cv.n.x);
        clipped_normals.push_back(cv.n.y);
        clipped_normals.push_back(cv.n.z);
    }

    const size_t nvp = vs.size() / 4;
    projected_vertices.resize(nvp * 4);
    float min_x = INFINITY, max_x = -INFINITY, min_y = INFINITY, max_y = -INFINITY;
    for (size_t i = 0; i < nvp; ++i) {  // :689-700 and :749-768
        Vec4 r = projection_matrix * Vec4{vs[4 * i], vs[4 * i + 1], vs[4 * i + 2], vs[4 * i + 3]};
        const float w = r.w;
        float *o = &projected_vertices[4 * i];
        o[0] = ((r.x / w) * 0.5f + 0.5f) * viewport_width;
        o[1] = ((-r.y / w) * 0.5f + 0.5f) * viewport_height;
        o[2] = r.z / w;
        o[3] = w;
        min_x = fmin_rs(min_x, o[0]);
        max_x = fmax_rs(max_x, o[0]);
        min_y = fmin_rs(min_y, o[1]);
        max_y = fmax_rs(max_y, o[1]);
    }
    has_bounding_box = true;
    bounding_box = Rect{min_x, min_y, max_x - min_x, max_y - min_y};

    const size_t ntc = clipped_indices.size() / 3;
    edges.resize(ntc);
    edge_visible.resize(ntc);
    for (size_t t = 0; t < ntc; ++t) {  // :706-739
        const float *v0 = &projected_vertices[4 * clipped_indices[3 * t]];
        const float *v1 = &projected_vertices[4 * clipped_indices[3 * t + 1]];
        const float *v2 = &projected_vertices[4 * clipped_indices[3 * t + 2]];
        const bool front = front_facing(v0, v1, v2);
        bool visible, swap;
        switch (cull_mode_) {
            case CullMode::Off: swap = front; visible = true; break;
            case CullMode::Front: swap = false; visible = !front; break;
            default: swap = front; visible = front; break;  // Back
        }
        const bool ev = t < edge_visibility.size() ? edge_visibility[t] != 0 : true;
        edges[t] = swap ? triangle_edges(v0, v2, v1, ev && visible) : triangle_edges(v0, v1, v2, ev && visible);
        edge_visible[t] = (ev && visible) ? 1u : 0u;
    }
    return true;
}

// ---- Batch2D ------------------------------------------------------------------------------------
Batch2D Batch2D::make(const float *verts2, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2) {
    Batch2D b;
    b.vertices.assign(verts2, verts2 + nv * 2);
    b.uvs.assign(uvs2, uvs2 + nv * 2);
    b.indices.assign(idx3, idx3 + nt * 3);
    return b;
}

Batch2D Batch2D::from_rectangle(float x, float y, float w, float h) {
    const float v[8] = {x, y, x, y + h, x + w, y + h, x + w, y};
    const uint32_t idx[6] = {0, 1, 2, 0, 2, 3};
    const float uv[8] = {0, 0, 0, 1, 1, 1, 1, 0};
    return make(v, 4, idx, 2, uv);
}

void Batch2D::project(const Mat3 *matrix) {
    const size_t nv = vertices.size() / 2;
    projected_vertices.resize(nv * 2);
    float min_x = INFINITY, max_x = -INFINITY, min_y = INFINITY, max_y = -INFINITY;
    for (size_t i = 0; i < nv; ++i) {
        float px = vertices[2 * i], py = vertices[2 * i + 1];
        if (matrix) {
            Vec3 r = (*matrix) * Vec3{px, py, 1.0f};
            px = r.x;
            py = r.y;
        }
        min_x = fmin_rs(min_x, px);
        max_x = fmax_rs(max_x, px);
        min_y = fmin_rs(min_y, py);
        max_y = fmax_rs(max_y, py);
        projected_vertices[2 * i] = px;
        projected_vertices[2 * i + 1] = py;
    }
    has_bounding_box = true;
    bounding_box = Rect{min_x, min_y, max_x - min_x, max_y - min_y};
    const size_t nt = indices.size() / 3;
    edges.resize(nt);
    // Lines batches index only .0/.1 meaningfully; the reference still builds Edges from all three
    for (size_t t = 0; t < nt; ++t) {
        const uint32_t i0 = indices[3 * t], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
        if (i0 >= nv || i1 >= nv || i2 >= nv) {
            edges[t] = rxr_edges{};
            continue;
        }
        edges[t] = triangle_edges(&projected_vertices[2 * i0], &projected_vertices[2 * i1], &projected_vertices[2 * i2], true);
    }
}

// ---- Scene --------------------------------------------------------------------------------------
// src/scene.rs:155-215: every batch list is projected with rayon's par_iter_mut -- the batches are independent.  Here: one job per
// batch through the worker pool (rxr_parallel.h), largest lists first come out of the atomic cursor in submission order; frames with
// little geometry run inline.
namespace {
struct ProjectJobs {
    std::vector<Batch2D *> d2;
    std::vector<Batch3D *> d3;
    size_t weight = 0;
    void add(std::vector<Batch2D> &l) {
        for (Batch2D &b : l) {
            d2.push_back(&b);
            weight += b.vertices.size() / 2 + b.indices.size();
        }
    }
    void add(std::vector<Batch3D> &l) {
        for (Batch3D &b : l) {
            d3.push_back(&b);
            weight += 4 * (b.vertices.size() / 4 + b.indices.size());  // clipping, Edges::new, bounding box
        }
    }
};
}  // namespace

std::vector<const Batch3D *> Scene::batches3d_in_order() const {
    std::vector<const Batch3D *> out;
    for (const Chunk &c : chunks) {
        for (const Batch3D &b : c.batches3d_opacity) out.push_back(&b);
        for (const Batch3D &b : c.batches3d) out.push_back(&b);
        for (const Batch3D &b : c.terrain_batch3d) out.push_back(&b);
    }
    for (const Batch3D &b : d3_static) out.push_back(&b);
    for (const Batch3D &b : d3_dynamic) out.push_back(&b);
    for (const Batch3D &b : d3_overlay) out.push_back(&b);
    return out;
}

bool Scene::project(const Mat3 *m2d, const Mat4 &view, const Mat4 &proj, float w, float h,
                    const std::function<void(size_t, const Batch3D &)> *on_projected3d) {
    ProjectJobs jobs;
    for (Chunk &c : chunks) {
        jobs.add(c.batches2d);
        jobs.add(c.terrain_batch2d);
        jobs.add(c.batches3d_opacity);
        jobs.add(c.batches3d);
        jobs.add(c.terrain_batch3d);
    }
    jobs.add(d2_static);
    jobs.add(d2_dynamic);
    jobs.add(d3_static);
    jobs.add(d3_dynamic);
    jobs.add(d3_overlay);
    std::atomic<bool> ok{true};
    const size_t n3 = jobs.d3.size();
    rxr_parallel::run(n3 + jobs.d2.size(), jobs.weight, [&](size_t i) {
        if (i < n3) {
            if (!jobs.d3[i]->clip_and_project(view, proj, w, h)) ok.store(false, std::memory_order_relaxed);
            else if (on_projected3d) (*on_projected3d)(i, *jobs.d3[i]);  // (jobs.d3 is in submission order: batches3d_in_order())
        } else {
            jobs.d2[i - n3]->project(m2d);
        }
    });
    return ok.load();
}

void Scene::project_2d(const Mat3 *m2d) {
    ProjectJobs jobs;
    for (Chunk &c : chunks) {
        jobs.add(c.batches2d);
        jobs.add(c.terrain_batch2d);
    }
    jobs.add(d2_static);
    jobs.add(d2_dynamic);
    rxr_parallel::run(jobs.d2.size(), jobs.weight, [&](size_t i) { jobs.d2[i]->project(m2d); });
}

// ---- device context -----------------------------------------------------------------------------
// The host layer keeps ONE device context (plain or multi-device) and its upload caches per process.  In the reference
// every Rasterizer is an independent value; here Rasterizers on different threads share that context, so every entry point
// that touches it (context, set_device(s), Rasterizer::upload / rasterize) holds g_mu for its whole duration: calls from
// several threads are serialised, never interleaved.
namespace {
std::recursive_mutex g_mu;
rxr_ctx *g_ctx = nullptr;
int g_device = -1;
std::vector<int> g_devices;  // more than one entry: rxr_create_multi
std::string g_error;
uint64_t g_tex_static_gen = 0, g_tex_dynamic_gen = 0;
uint64_t g_shaders_gen = 0, g_shader_env_gen = 0;
// (RXR_DEVICE_PROJECTION=1 in the environment makes device projection the initial choice, as in the Rust shim: shim/.../lib.rs)
bool g_device_projection = [] { const char *e = getenv("RXR_DEVICE_PROJECTION"); return e && e[0] == '1'; }();
bool g_device_edges = !(getenv("RXR_HOST_EDGES") && atoi(getenv("RXR_HOST_EDGES")) != 0);
// the form this thread's current frame takes across the ABI (every 3D batch of a frame the same: rxr.h): without Edges records, unless the
// setting says otherwise or the page-locked streaming hand-over can only be promised for the records (see Rasterizer::upload)
thread_local bool t_frame_edgeless = true;
int g_light_math = RXR_LIGHT_MATH_RELAXED;  // the library's default
uint64_t g_mesh_fingerprint = 0;
uint64_t g_mesh2d_fingerprint = 0;
}  // namespace

void set_device_projection(bool on) { g_device_projection = on; }
bool device_projection() { return g_device_projection; }
void set_device_edges(bool on) { g_device_edges = on; }
bool device_edges() { return g_device_edges; }
void set_light_math(bool exact) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_light_math = exact ? RXR_LIGHT_MATH_EXACT : RXR_LIGHT_MATH_RELAXED;
    if (g_ctx) (void)rxr_set_light_math(g_ctx, g_light_math);
}
bool light_math_exact() { return g_light_math == RXR_LIGHT_MATH_EXACT; }

const std::string &last_error() { return g_error; }

namespace {
void drop_context_locked() {
    if (g_ctx) rxr_destroy(g_ctx);
    g_ctx = nullptr;
    g_mesh_fingerprint = 0;
    g_mesh2d_fingerprint = 0;
    g_tex_static_gen = g_tex_dynamic_gen = 0;
    g_shaders_gen = g_shader_env_gen = 0;
}
}  // namespace

void set_device(int device) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g_ctx && (device != g_device || g_devices.size() > 1)) drop_context_locked();
    g_device = device;
    g_devices.clear();
}

void set_devices(const int *devices, int n) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    std::vector<int> want(devices, devices + (n > 0 ? n : 0));
    if (want.size() <= 1) {
        set_device(want.empty() ? -1 : want[0]);
        return;
    }
    if (g_ctx && want != g_devices) drop_context_locked();
    if (g_ctx && g_devices.empty()) drop_context_locked();
    g_devices = want;
    g_device = want[0];
}

rxr_ctx *context(std::string *error) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g_ctx) {
        int dev = g_device;
        if (dev < 0) {
            const char *e = getenv("RXR_DEVICE");
            if (!e) e = getenv("LOCAL_RANK");
            dev = e ? atoi(e) : 0;
            int n = rxr_device_count();
            if (n > 0) dev %= n;
        }
        int rc = g_devices.size() > 1 ? rxr_create_multi(&g_ctx, g_devices.data(), (int)g_devices.size()) : rxr_create(&g_ctx, dev);
        if (rc != RXR_OK) {
            g_error = rxr_last_error(nullptr);
            if (error) *error = g_error;
            g_ctx = nullptr;
            return nullptr;
        }
        g_device = dev;
        (void)rxr_set_light_math(g_ctx, g_light_math);
    }
    return g_ctx;
}

// ---- Rasterizer ---------------------------------------------------------------------------------
Rasterizer Rasterizer::setup(const Mat3 *m2d, const Mat4 &view, const Mat4 &proj) {
    Rasterizer r;
    r.inverse_view_matrix = rvek::inverted(view);
    r.camera_pos = Vec3{r.inverse_view_matrix.m[12], r.inverse_view_matrix.m[13], r.inverse_view_matrix.m[14]};
    if (m2d) {
        r.has_m2d = true;
        r.projection_matrix_2d = *m2d;
        r.translationd2 = Vec2{m2d->at(0, 2), m2d->at(1, 2)};
        r.scaled2 = m2d->at(0, 0);
    }
    r.inverse_projection_matrix = rvek::inverted(proj);
    r.view_matrix = view;
    r.projection_matrix = proj;
    return r;
}

namespace {

void tile_views(const std::vector<Tile> &tiles, std::vector<rxr_texture> &texs, std::vector<rxr_tile> &out) {
    size_t n = 0;
    for (const Tile &t : tiles) n += t.textures.size();
    texs.clear();
    texs.reserve(n);
    out.clear();
    for (const Tile &t : tiles) {
        rxr_tile rt{};
        rt.textures = texs.data() + texs.size();
        rt.n_textures = (uint32_t)t.textures.size();
        for (const Texture &x : t.textures) texs.push_back(rxr_texture{x.data.data(), x.width, x.height});
        out.push_back(rt);
    }
}

// EntityTile(id, index) / ItemTile(id, index) -> what the device gets: the tile's slot among the dynamic tiles, or
// RXR_SOURCE_MISSING where the reference's two lookups fail (rasterizer.rs:1140-1187, :705-748, :1548-1595)
struct SequenceSlots {
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> entity, item;
    rxr_source resolve(const PixelSource &p) const {
        rxr_source o{};
        o.kind = p.kind;
        o.index = p.index;
        memcpy(o.pixel, p.pixel, 4);
        if (p.kind == RXR_HOST_SOURCE_ENTITY_TILE || p.kind == RXR_HOST_SOURCE_ITEM_TILE) {
            const auto &m = p.kind == RXR_HOST_SOURCE_ENTITY_TILE ? entity : item;
            const auto it = m.find({p.index, p.seq});
            o.kind = it == m.end() ? (uint32_t)RXR_SOURCE_MISSING : (uint32_t)RXR_SOURCE_DYNAMIC_TILE;
            o.index = it == m.end() ? 0u : it->second;
        }
        return o;
    }
};

rxr_batch3d view3d(const Batch3D &b, uint32_t list, int chunk, const SequenceSlots &slots) {
    rxr_batch3d o{};
    o.projected_vertices = b.projected_vertices.data();
    o.clipped_uvs = b.clipped_uvs.data();
    o.clipped_normals = b.normals.empty() ? nullptr : b.clipped_normals.data();
    o.clipped_indices = b.clipped_indices.data();
    o.edges = t_frame_edgeless ? nullptr : b.edges.data();
    o.edge_visible = b.edge_visible.data();
    o.cull_mode = (uint32_t)b.cull_mode_;
    o.n_vertices = (uint32_t)(b.projected_vertices.size() / 4);
    o.n_triangles = (uint32_t)b.edges.size();
    o.has_bounding_box = b.has_bounding_box ? 1u : 0u;
    o.bounding_box[0] = b.bounding_box.x; o.bounding_box[1] = b.bounding_box.y;
    o.bounding_box[2] = b.bounding_box.width; o.bounding_box[3] = b.bounding_box.height;
    o.repeat_mode = b.repeat_mode_;
    o.source = slots.resolve(b.source_);
    o.ambient_color[0] = b.ambient_color_.x; o.ambient_color[1] = b.ambient_color_.y; o.ambient_color[2] = b.ambient_color_.z;
    o.shader = b.shader_;
    o.has_profile_id = b.has_profile_id ? 1u : 0u;
    o.profile_id = b.profile_id_;
    o.list = list;
    o.chunk = chunk;
    return o;
}

rxr_batch2d view2d(const Batch2D &b, int chunk, const SequenceSlots &slots) {
    rxr_batch2d o{};
    o.projected_vertices = b.projected_vertices.data();
    o.uvs = b.uvs.data();
    o.indices = b.indices.data();
    o.edges = t_frame_edgeless ? nullptr : b.edges.data();  // (ABI 5: the library builds Batch2D::project's records itself)
    o.n_vertices = (uint32_t)(b.projected_vertices.size() / 2);
    o.n_triangles = (uint32_t)(b.indices.size() / 3);
    o.has_bounding_box = b.has_bounding_box ? 1u : 0u;
    o.bounding_box[0] = b.bounding_box.x; o.bounding_box[1] = b.bounding_box.y;
    o.bounding_box[2] = b.bounding_box.width; o.bounding_box[3] = b.bounding_box.height;
    o.mode = b.mode_;
    o.repeat_mode = b.repeat_mode_;
    o.source = slots.resolve(b.source_);
    o.receives_light = b.receives_light_ ? 1u : 0u;
    o.shader = b.shader_;
    o.chunk = chunk;
    return o;
}

}  // namespace

int Rasterizer::upload(Scene &scene, size_t w, size_t h, size_t tile_size, const Assets &assets) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    std::string err;
    rxr_ctx *ctx = context(&err);
    if (!ctx) return RXR_ERR_NO_DEVICE;

    width = (float)w;
    height = (float)h;
    hash_anim = hash_u32((uint32_t)scene.animation_frame);  // :208

    const bool on_device = g_device_projection;
    t_frame_edgeless = g_device_edges;
    if (on_device) {
        // both halves of Scene::project run on the GPU (rxr_set_meshes / rxr_set_meshes2d below)
    } else {
        const auto tp = std::chrono::steady_clock::now();
        // Large scenes on a plain context: every 3D batch is handed to the device as soon as it is projected (rxr_stream_batch3d) --
        // the copy into pinned memory and the PCIe transfer of the batches that are done run under the projection of the rest.
        // RXR_STREAM_UPLOAD=0 keeps the plain sequence (project everything, then rxr_upload_frame copies everything).
        std::function<void(size_t, const Batch3D &)> hand_over;
        const char *su = getenv("RXR_STREAM_UPLOAD");  // (read per frame: tests switch it)
        const bool stream_off = su && su[0] == '0', stream_forced = su && su[0] == 'f';
        if (!stream_off && rxr_member_count(ctx) == 1) {
            const std::vector<const Batch3D *> order = scene.batches3d_in_order();
            size_t elements = 0;
            std::vector<uint32_t> cap_v(order.size()), cap_t(order.size());
            for (size_t i = 0; i < order.size(); ++i) {
                const size_t nv = order[i]->vertex_count(), nt = order[i]->triangle_count();
                cap_v[i] = (uint32_t)std::min<size_t>(nv + 4 * nt, 0x7FFFFFFFu);  // the near-plane clip appends at most 4 vertices and
                cap_t[i] = (uint32_t)std::min<size_t>(3 * nt, 0x7FFFFFFFu);       // 2 triangles per triangle (batch3d.rs:627-686)
                elements += nv + nt;
            }
            // the promise of rxr_stream_begin_pinned is made when the buffers every batch projected into LAST frame are page-locked
            // (clip_and_project refills the same vectors; the first frame of a scene, whose vectors do not exist yet, copies).  It is
            // checked again per batch at hand-over: a vector that had to grow into ordinary memory is not handed over, and the frame
            // then goes the plain way.
            // (the `visible` words are a tenth of the records: in batches of fewer than 2048 triangles they stay below the size from which
            // the vectors are page-locked at all.  The frame then promises -- and sends -- the records, as rounds 1-3 did)
            auto arrays_pinned_with = [](const Batch3D &b, bool edgeless) {
                return is_pinned(b.projected_vertices) && is_pinned(b.clipped_uvs) && is_pinned(b.clipped_indices) &&
                       (edgeless ? is_pinned(b.edge_visible) : is_pinned(b.edges)) && (b.normals.empty() || is_pinned(b.clipped_normals));
            };
            bool pinned = true, pinned_records = true;
            for (size_t i = 0; i < order.size() && (pinned || pinned_records); ++i) {
                pinned = pinned && arrays_pinned_with(*order[i], t_frame_edgeless);
                pinned_records = pinned_records && arrays_pinned_with(*order[i], false);
            }
            if (!pinned && pinned_records && t_frame_edgeless) {
                t_frame_edgeless = false;
                pinned = true;
            }
            const bool edgeless = t_frame_edgeless;
            auto arrays_pinned = [arrays_pinned_with, edgeless](const Batch3D &b) { return arrays_pinned_with(b, edgeless); };
            const bool large = (order.size() >= 8 && elements >= (1u << 20)) || (stream_forced && order.size() >= 2);
            const int began = !large ? RXR_ERR_UNSUPPORTED
                              : pinned ? rxr_stream_begin_pinned(ctx, (uint32_t)order.size(), cap_v.data(), cap_t.data())
                                       : rxr_stream_begin(ctx, (uint32_t)order.size(), cap_v.data(), cap_t.data());
            if (began == RXR_OK) {
                hand_over = [ctx, pinned, arrays_pinned, edgeless](size_t i, const Batch3D &b) {
                    if (pinned && b.edges.size() && !arrays_pinned(b)) return;  // (the promise does not hold for this batch: not handed over)
                    rxr_batch3d v{};
                    v.projected_vertices = b.projected_vertices.data();
                    v.clipped_uvs = b.clipped_uvs.data();
                    v.clipped_normals = b.normals.empty() ? nullptr : b.clipped_normals.data();
                    v.clipped_indices = b.clipped_indices.data();
                    v.edges = edgeless ? nullptr : b.edges.data();
                    v.edge_visible = b.edge_visible.data();
                    v.cull_mode = (uint32_t)b.cull_mode_;
                    v.n_vertices = (uint32_t)(b.projected_vertices.size() / 4);
                    v.n_triangles = (uint32_t)b.edges.size();
                    (void)rxr_stream_batch3d(ctx, (uint32_t)i, &v);  // (a refusal makes rxr_upload_frame hand the frame over from scratch)
                };
            }
        }
        static const bool timing0 = getenv("RXR_E2E_TIMING") != nullptr;
        if (timing0) fprintf(stderr, "rxr_e2e_timing before project (stream set-up) %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count());
        if (!scene.project(has_m2d ? &projection_matrix_2d : nullptr, view_matrix, projection_matrix, width, height, hand_over ? &hand_over : nullptr)) {  // :210
            g_error = "clip_and_project: batch without normals (the reference panics at batch3d.rs:605)";
            return RXR_ERR_INVALID;
        }
        static const bool timing = getenv("RXR_E2E_TIMING") != nullptr;
        if (timing) fprintf(stderr, "rxr_e2e_timing project_ms=%.3f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count());
    }
    for (const Chunk &c : scene.chunks)  // :219-223
        for (const CompiledLight &l : c.lights) scene.dynamic_lights.push_back(l);

    // the dynamic tiles the device sees: scene.dynamic_textures, then every sequence tile of assets.entity_tiles and
    // assets.item_tiles (ids ascending, sequences in insertion order); EntityTile / ItemTile sources are resolved to these slots
    SequenceSlots slots;
    std::vector<const Tile *> dyn_tiles;
    for (const Tile &t : scene.dynamic_textures) dyn_tiles.push_back(&t);
    for (int pass = 0; pass < 2; ++pass)
        for (const auto &kv : pass == 0 ? assets.entity_tiles : assets.item_tiles)
            for (size_t k = 0; k < kv.second.size(); ++k) {
                (pass == 0 ? slots.entity : slots.item)[{kv.first, (uint32_t)k}] = (uint32_t)dyn_tiles.size();
                dyn_tiles.push_back(&kv.second[k]);
            }

    // textures: re-upload only when the asset set changed
    if (g_tex_static_gen != assets.generation || g_tex_dynamic_gen != scene.dynamic_textures_generation) {
        std::vector<rxr_texture> ts, td;
        std::vector<rxr_tile> tiles_s, tiles_d;
        tile_views(assets.tile_list, ts, tiles_s);
        {
            size_t n = 0;
            for (const Tile *t : dyn_tiles) n += t->textures.size();
            td.reserve(n);
            for (const Tile *t : dyn_tiles) {
                rxr_tile rt{};
                rt.textures = td.data() + td.size();
                rt.n_textures = (uint32_t)t->textures.size();
                for (const Texture &x : t->textures) td.push_back(rxr_texture{x.data.data(), x.width, x.height});
                tiles_d.push_back(rt);
            }
        }
        int rc = rxr_set_textures(ctx, tiles_s.data(), (uint32_t)tiles_s.size(), tiles_d.data(), (uint32_t)tiles_d.size());
        if (rc != RXR_OK) {
            g_error = rxr_last_error(ctx);
            return rc;
        }
        g_tex_static_gen = assets.generation;
        g_tex_dynamic_gen = scene.dynamic_textures_generation;
    }

    // Rusteria programs, patterns, palette: re-sent only when they changed
    // one table: scene.shaders first, then every chunk's shaders (rxr_chunk.program_base points into it)
    std::vector<const Program *> all_programs;
    for (const Program &p : scene.shaders) all_programs.push_back(&p);
    std::vector<uint32_t> chunk_program_base(scene.chunks.size(), 0);
    for (size_t c = 0; c < scene.chunks.size(); ++c) {
        chunk_program_base[c] = (uint32_t)all_programs.size();
        for (const Program &p : scene.chunks[c].shaders) all_programs.push_back(&p);
    }
    if (g_shaders_gen != scene.shaders_generation || g_shader_env_gen != assets.shader_env_generation) {
        std::vector<std::vector<rxr_function>> fns(all_programs.size());
        std::vector<rxr_program> progs(all_programs.size());
        for (size_t i = 0; i < all_programs.size(); ++i) {
            const Program &p = *all_programs[i];
            for (const auto &f : p.user_functions) fns[i].push_back(rxr_function{f.data(), (uint32_t)f.size()});
            progs[i] = rxr_program{p.globals, p.shade_index, p.shade_locals, fns[i].data(), (uint32_t)fns[i].size()};
        }
        auto views = [](const std::vector<Pattern> &src) {
            std::vector<rxr_pattern> v;
            for (const Pattern &t : src) v.push_back(rxr_pattern{t.rgb.data(), t.width, t.height});
            return v;
        };
        std::vector<rxr_pattern> pv = views(assets.patterns), pn = views(assets.patterns_normal);
        rxr_shader_set set{};
        set.programs = progs.data();
        set.n_programs = (uint32_t)progs.size();
        set.patterns = pv.data();
        set.n_patterns = (uint32_t)pv.size();
        set.normal_patterns = pn.data();
        set.n_normal_patterns = (uint32_t)pn.size();
        set.palette_rgb = assets.palette_rgb.data();
        set.palette_present = assets.palette_present.data();
        set.n_palette = (uint32_t)assets.palette_present.size();
        int rc = rxr_set_shaders(ctx, &set);
        if (rc != RXR_OK) {
            g_error = rxr_last_error(ctx);
            g_shaders_gen = 0;
            return rc;
        }
        g_shaders_gen = scene.shaders_generation;
        g_shader_env_gen = assets.shader_env_generation;
    }

    // flatten in submission order (:314-405, :503-552)
    std::vector<rxr_batch3d> b3;
    std::vector<rxr_batch2d> b2;
    std::vector<rxr_chunk> chunks;
    std::vector<std::vector<rxr_texture>> chunk_shader_textures(scene.chunks.size());
    std::vector<rxr_texture> chunk_terrain_textures(scene.chunks.size());
    for (size_t c = 0; c < scene.chunks.size(); ++c) {
        const Chunk &ch = scene.chunks[c];
        for (const Batch3D &b : ch.batches3d_opacity) b3.push_back(view3d(b, RXR_LIST_CHUNK_OPACITY, (int)c, slots));
        for (const Batch3D &b : ch.batches3d) b3.push_back(view3d(b, RXR_LIST_CHUNK, (int)c, slots));
        for (const Batch3D &b : ch.terrain_batch3d) b3.push_back(view3d(b, RXR_LIST_CHUNK_TERRAIN, (int)c, slots));  // :343-356
        for (const Batch2D &b : ch.batches2d) b2.push_back(view2d(b, (int)c, slots));
        for (const Batch2D &b : ch.terrain_batch2d) b2.push_back(view2d(b, (int)c, slots));  // :515-525
        rxr_chunk rc{};
        rc.occluders = ch.occluded_sectors.data();
        rc.n_occluders = (uint32_t)ch.occluded_sectors.size();
        rc.program_base = chunk_program_base[c];
        rc.n_programs = (uint32_t)ch.shaders.size();
        for (size_t k = 0; k < ch.shader_textures.size(); ++k) {
            const Texture &t = ch.shader_textures[k];
            chunk_shader_textures[c].push_back(rxr_texture{ch.shader_texture_present[k] ? t.data.data() : nullptr, (uint32_t)t.width, (uint32_t)t.height});
        }
        rc.shader_textures = chunk_shader_textures[c].data();
        rc.n_shader_textures = (uint32_t)chunk_shader_textures[c].size();
        if (ch.has_terrain_texture) {
            chunk_terrain_textures[c] = rxr_texture{ch.terrain_texture.data.data(), (uint32_t)ch.terrain_texture.width, (uint32_t)ch.terrain_texture.height};
            rc.terrain_texture = &chunk_terrain_textures[c];
        }
        rc.origin[0] = ch.origin[0];
        rc.origin[1] = ch.origin[1];
        rc.size = ch.size;
        chunks.push_back(rc);
    }
    for (const Batch3D &b : scene.d3_static) b3.push_back(view3d(b, RXR_LIST_STATIC, -1, slots));
    for (const Batch3D &b : scene.d3_dynamic) b3.push_back(view3d(b, RXR_LIST_DYNAMIC, -1, slots));
    for (const Batch3D &b : scene.d3_overlay) b3.push_back(view3d(b, RXR_LIST_OVERLAY, -1, slots));

    // device-side projection: the same batches in the same order, as object-space meshes
    std::vector<rxr_mesh3d> meshes;
    std::vector<float> mesh_transforms;
    if (on_device) {
        uint64_t fp = 1469598103934665603ull;  // FNV-1a over what rxr_set_meshes copies (transforms travel per frame)
        auto mix = [&](const void *p, size_t n) {
            const uint8_t *q = (const uint8_t *)p;
            for (size_t i = 0; i < n; ++i) fp = (fp ^ q[i]) * 1099511628211ull;
        };
        auto add = [&](const Batch3D &b, uint32_t list, int chunk) -> bool {
            if (!b.indices.empty() && b.normals.size() / 3 < b.vertex_count()) return false;  // batch3d.rs:605 panics
            rxr_mesh3d m{};
            m.vertices = b.vertices.data();
            m.indices = b.indices.data();
            m.uvs = b.uvs.data();
            m.normals = b.normals.empty() ? nullptr : b.normals.data();
            m.n_vertices = (uint32_t)b.vertex_count();
            m.n_triangles = (uint32_t)b.triangle_count();
            memcpy(m.transform_3d, b.transform_3d.m, 64);
            m.cull_mode = (uint32_t)b.cull_mode_;
            m.repeat_mode = b.repeat_mode_;
            m.source = slots.resolve(b.source_);
            m.ambient_color[0] = b.ambient_color_.x; m.ambient_color[1] = b.ambient_color_.y; m.ambient_color[2] = b.ambient_color_.z;
            m.shader = b.shader_;
            m.has_profile_id = b.has_profile_id ? 1u : 0u;
            m.profile_id = b.profile_id_;
            m.list = list;
            m.chunk = chunk;
            meshes.push_back(m);
            mesh_transforms.insert(mesh_transforms.end(), b.transform_3d.m, b.transform_3d.m + 16);
            const uint32_t meta[12] = {m.n_vertices, m.n_triangles, m.cull_mode, m.repeat_mode, m.source.kind, m.source.index,
                                       (uint32_t)m.shader, m.has_profile_id, m.profile_id, m.list, (uint32_t)m.chunk,
                                       (uint32_t)b.normals.size()};
            mix(meta, sizeof(meta));
            mix(m.source.pixel, 4);
            mix(m.ambient_color, 12);
            mix(&b.geometry_stamp, 8);  // geometry identity (Batch3D::touch)
            return true;
        };
        bool ok = true;
        for (size_t c = 0; c < scene.chunks.size(); ++c) {
            for (const Batch3D &b : scene.chunks[c].batches3d_opacity) ok = ok && add(b, RXR_LIST_CHUNK_OPACITY, (int)c);
            for (const Batch3D &b : scene.chunks[c].batches3d) ok = ok && add(b, RXR_LIST_CHUNK, (int)c);
            for (const Batch3D &b : scene.chunks[c].terrain_batch3d) ok = ok && add(b, RXR_LIST_CHUNK_TERRAIN, (int)c);
        }
        for (const Batch3D &b : scene.d3_static) ok = ok && add(b, RXR_LIST_STATIC, -1);
        for (const Batch3D &b : scene.d3_dynamic) ok = ok && add(b, RXR_LIST_DYNAMIC, -1);
        for (const Batch3D &b : scene.d3_overlay) ok = ok && add(b, RXR_LIST_OVERLAY, -1);
        if (!ok) {
            g_error = "clip_and_project: batch without normals (the reference panics at batch3d.rs:605)";
            return RXR_ERR_INVALID;
        }
        if (fp != g_mesh_fingerprint) {
            int rc = rxr_set_meshes(ctx, meshes.data(), (uint32_t)meshes.size());
            if (rc != RXR_OK) {
                g_error = rxr_last_error(ctx);
                g_mesh_fingerprint = 0;
                return rc;
            }
            g_mesh_fingerprint = fp;
        }
        b3.clear();
    }
    for (const Batch2D &b : scene.d2_static) b2.push_back(view2d(b, -1, slots));
    for (const Batch2D &b : scene.d2_dynamic) b2.push_back(view2d(b, -1, slots));
    if (on_device) {
        // the 2D batches in the same order, as object-space meshes; registered again when anything Batch2D::project reads has changed
        // (2D batches are small: the fingerprint covers their contents)
        std::vector<rxr_mesh2d> m2;
        uint64_t fp = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t n) {
            const uint8_t *q = (const uint8_t *)p;
            for (size_t i = 0; i < n; ++i) fp = (fp ^ q[i]) * 1099511628211ull;
        };
        auto add2 = [&](const Batch2D &b, int chunk) {
            rxr_mesh2d m{};
            m.vertices = b.vertices.data();
            m.indices = b.indices.data();
            m.uvs = b.uvs.data();
            m.n_vertices = (uint32_t)(b.vertices.size() / 2);
            m.n_triangles = (uint32_t)(b.indices.size() / 3);
            m.mode = b.mode_;
            m.repeat_mode = b.repeat_mode_;
            m.source = slots.resolve(b.source_);
            m.receives_light = b.receives_light_ ? 1u : 0u;
            m.shader = b.shader_;
            m.chunk = chunk;
            m2.push_back(m);
            const uint32_t meta[9] = {m.n_vertices, m.n_triangles, m.mode, m.repeat_mode, m.source.kind, m.source.index, m.receives_light, (uint32_t)m.shader, (uint32_t)m.chunk};
            mix(meta, sizeof(meta));
            mix(m.source.pixel, 4);
            mix(b.vertices.data(), b.vertices.size() * 4);
            mix(b.uvs.data(), b.uvs.size() * 4);
            mix(b.indices.data(), b.indices.size() * 4);
        };
        for (size_t c = 0; c < scene.chunks.size(); ++c) {
            for (const Batch2D &b : scene.chunks[c].batches2d) add2(b, (int)c);
            for (const Batch2D &b : scene.chunks[c].terrain_batch2d) add2(b, (int)c);
        }
        for (const Batch2D &b : scene.d2_static) add2(b, -1);
        for (const Batch2D &b : scene.d2_dynamic) add2(b, -1);
        if (fp != g_mesh2d_fingerprint) {
            int rc = rxr_set_meshes2d(ctx, m2.data(), (uint32_t)m2.size());
            if (rc != RXR_OK) {
                g_error = rxr_last_error(ctx);
                g_mesh2d_fingerprint = 0;
                return rc;
            }
            g_mesh2d_fingerprint = fp;
        }
        (void)rxr_set_projection2d(ctx, has_m2d ? projection_matrix_2d.m : nullptr);
        b2.clear();
    }

    std::vector<rxr_light> lights(scene.lights);
    lights.insert(lights.end(), scene.dynamic_lights.begin(), scene.dynamic_lights.end());

    rxr_frame f{};
    f.abi_version = RXR_ABI_VERSION;
    f.width = (uint32_t)w;
    f.height = (uint32_t)h;
    f.tile_size = (uint32_t)tile_size;
    memcpy(f.inverse_view, inverse_view_matrix.m, 64);
    memcpy(f.inverse_projection, inverse_projection_matrix.m, 64);
    f.camera_pos[0] = camera_pos.x; f.camera_pos[1] = camera_pos.y; f.camera_pos[2] = camera_pos.z;
    f.translationd2[0] = translationd2.x; f.translationd2[1] = translationd2.y;
    f.scaled2 = scaled2;
    f.hash_anim = hash_anim;
    f.animation_frame = scene.animation_frame;
    f.flags = (d2_active ? RXR_FLAG_D2_ACTIVE : 0u) | (d3_active ? RXR_FLAG_D3_ACTIVE : 0u) |
              (ignore_background_shader ? RXR_FLAG_IGNORE_BG_SHADER : 0u) |
              (preserve_transparency ? RXR_FLAG_PRESERVE_TRANSPARENCY : 0u) |
              (has_background_color ? RXR_FLAG_HAS_BACKGROUND_COLOR : 0u) | (has_ambient ? RXR_FLAG_HAS_AMBIENT : 0u) |
              (has_sun ? RXR_FLAG_HAS_SUN : 0u);
    memcpy(f.background_color, background_color, 4);
    f.ambient[0] = ambient_color.x; f.ambient[1] = ambient_color.y; f.ambient[2] = ambient_color.z; f.ambient[3] = ambient_color.w;
    f.sun_dir[0] = sun_dir.x; f.sun_dir[1] = sun_dir.y; f.sun_dir[2] = sun_dir.z;
    f.day_factor = day_factor;
    f.sample_mode = sample_mode_;
    f.time = time_;
    f.background_kind = scene.background;
    memcpy(f.background_grid, scene.background_grid, 16);
    f.has_brush_preview = has_brush_preview ? 1u : 0u;
    memcpy(f.brush_position, brush_position, 12);
    f.brush_radius = brush_radius;
    f.brush_falloff = brush_falloff;
    f.batches3d = b3.data();
    f.n_batches3d = (uint32_t)b3.size();
    f.batches2d = b2.data();
    f.n_batches2d = (uint32_t)b2.size();
    f.lights = lights.data();
    f.n_lights = (uint32_t)lights.size();
    f.occluders = mapmini.occluded_sectors.data();
    f.n_occluders = (uint32_t)mapmini.occluded_sectors.size();
    f.linedefs = mapmini.linedefs.data();
    f.n_linedefs = (uint32_t)mapmini.linedefs.size();
    f.chunks = chunks.data();
    f.n_chunks = (uint32_t)chunks.size();
    f.n_shader_programs = (uint32_t)scene.shaders.size();
    if (on_device) {
        f.use_meshes = 3;  // both halves
        memcpy(f.view, view_matrix.m, 64);
        memcpy(f.projection, projection_matrix.m, 64);
        f.mesh_transforms = mesh_transforms.empty() ? nullptr : mesh_transforms.data();
    }

    int rc = rxr_upload_frame(ctx, &f);
    if (rc != RXR_OK) g_error = rxr_last_error(ctx);
    return rc;
}

int Rasterizer::rasterize(Scene &scene, uint8_t *pixels, size_t w, size_t h, size_t tile_size, const Assets &assets) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (tile_size == 0) {
        g_error = "tile_size 0 (step_by(0) panics in the reference)";
        return RXR_ERR_INVALID;
    }
    static const bool timing = getenv("RXR_E2E_TIMING") != nullptr;  // diagnostics (tools/e2e_probe.py)
    const auto t0 = std::chrono::steady_clock::now();
    int rc = upload(scene, w, h, tile_size, assets);
    if (rc != RXR_OK) return rc;
    if (timing) fprintf(stderr, "rxr_e2e_timing project_and_handover_ms=%.3f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    rxr_ctx *ctx = context();
    rc = rxr_render_download(ctx, pixels);
    if (rc != RXR_OK) g_error = rxr_last_error(ctx);
    return rc;
}

// ---- cameras ------------------------------------------------------------------------------------
void orbit_camera(Vec3 center, float distance, float azimuth, float elevation, float fov, float near, float far, float w,
                  float h, Mat4 &view, Mat4 &proj) {
    const float x = distance * std::cos(azimuth) * std::cos(elevation);
    const float y = distance * std::sin(elevation);
    const float z = distance * std::sin(azimuth) * std::cos(elevation);
    const Vec3 eye = Vec3{x, y, z} + center;
    view = rvek::look_at_rh(eye, center, Vec3{0, 1, 0});
    proj = rvek::perspective_fov_rh_zo(fov * (3.14159265358979323846f / 180.0f), w, h, near, far);
}

void firstp_camera(Vec3 position, Vec3 center, float fov, float near, float far, float w, float h, Mat4 &view, Mat4 &proj) {
    view = rvek::look_at_rh(position, center, Vec3{0, 1, 0});
    proj = rvek::perspective_fov_rh_zo(fov * (3.14159265358979323846f / 180.0f), w, h, near, far);
}

}  // namespace rusterix
