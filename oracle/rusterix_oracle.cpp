// rusterix_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rusterix_oracle.hpp.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
// (-ffp-contract=off is mandatory: Rust never fuses a*b+c; the one fused op, vec4_to_pixel's
//  mul_add, is spelled fmaf below.)
#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "rusterix_oracle.hpp"

namespace orc {

// ---- src/edge.rs:12-24 -----------------------------------------------------------------------
Edges edges_new(const float v0[3][2], const float v1[3][2], bool visible) {
    Edges e{};
    for (int i = 0; i < 3; ++i) {
        e.a[i] = v1[i][1] - v0[i][1];                            // dy
        e.b[i] = v0[i][0] - v1[i][0];                            // -dx
        e.c[i] = v1[i][0] * v0[i][1] - v1[i][1] * v0[i][0];      // x1*y0 - y1*x0
    }
    e.visible = visible;
    return e;
}

// ---- src/rasterizer.rs:199-207 ---------------------------------------------------------------
uint32_t hash_u32(uint32_t seed) {
    uint32_t state = seed;
    state = (state ^ 61u) ^ (state >> 16);
    state = state + (state << 3);
    state ^= state >> 4;
    state = state * 0x27d4eb2du;
    state ^= state >> 15;
    return state;
}

// ---- src/lib.rs:50-79 ------------------------------------------------------------------------
static const float INV_255 = 1.0f / 255.0f;
void pixel_to_vec4(const uint8_t p[4], float out[4]) {
    out[0] = (float)p[0] * INV_255;
    out[1] = (float)p[1] * INV_255;
    out[2] = (float)p[2] * INV_255;
    out[3] = (float)p[3] * INV_255;
}
uint8_t f32_to_u8_saturated(float x) {
    float y = std::fmaf(rmin(rmax(x, 0.0f), 1.0f), 255.0f, 0.5f);  // x.max(0).min(1).mul_add(255, 0.5)
    return (uint8_t)sat_i32(y);                                     // `as i32 as u8`
}
void vec4_to_pixel(const float v[4], uint8_t out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = f32_to_u8_saturated(v[i]);
}

// ---- src/rasterizer.rs:19-33 -----------------------------------------------------------------
float srgb_to_linear_fast(float x) {
    float x2 = x * x;
    return (0.6975f * x2 + 0.3025f) * x;
}
float linear_to_srgb_fast(float x) {
    float sqrt_x = std::sqrt(x);
    return 1.055f * sqrt_x - 0.055f * sqrt_x * sqrt_x;
}

// ---- src/texture.rs:307-323 ------------------------------------------------------------------
void texture_sample_nearest(const Texture &t, float u, float v, uint8_t out[4]) {
    uint64_t tx = sat_usize(std::round(u * ((float)t.width - 1.0f)));
    uint64_t ty = sat_usize(std::round(v * ((float)t.height - 1.0f)));
    if (tx > t.width - 1) tx = t.width - 1;
    if (ty > t.height - 1) ty = t.height - 1;
    size_t idx = (ty * t.width + tx) * 4;
    out[0] = t.data[idx];
    out[1] = t.data[idx + 1];
    out[2] = t.data[idx + 2];
    out[3] = t.data[idx + 3];
}

// ---- src/texture.rs:414-460 ------------------------------------------------------------------
void texture_sample_linear(const Texture &t, float u, float v, uint8_t out[4]) {
    float x = u * ((float)t.width - 1.0f);
    float y = v * ((float)t.height - 1.0f);
    uint64_t x0 = sat_usize(std::floor(x));
    uint64_t x1 = (x0 + 1 < t.width - 1) ? x0 + 1 : t.width - 1;
    uint64_t y0 = sat_usize(std::floor(y));
    uint64_t y1 = (y0 + 1 < t.height - 1) ? y0 + 1 : t.height - 1;
    float dx = x - std::floor(x);
    float dy = y - std::floor(y);
    size_t idx00 = (y0 * t.width + x0) * 4;
    size_t idx10 = (y0 * t.width + x1) * 4;
    size_t idx01 = (y1 * t.width + x0) * 4;
    size_t idx11 = (y1 * t.width + x1) * 4;
    // the reference slices data[idx..idx+4] and would panic out of bounds; u,v are in [0,1] or NaN
    // after Texture::sample's repeat handling, so the indices stay in range.
    for (int i = 0; i < 4; ++i) {
        float v00 = (float)t.data[idx00 + i];
        float v10 = (float)t.data[idx10 + i];
        float v01 = (float)t.data[idx01 + i];
        float v11 = (float)t.data[idx11 + i];
        float v0 = v00 + dx * (v10 - v00);
        float v1 = v01 + dx * (v11 - v01);
        float vv = v0 + dy * (v1 - v0);
        out[i] = sat_u8(std::round(vv));
    }
}

// ---- src/texture.rs:203-232 ------------------------------------------------------------------
void texture_sample(const Texture &t, float u, float v, int sample_mode, int repeat_mode, uint8_t out[4]) {
    switch (repeat_mode) {
        case RXR_REPEAT_CLAMP_XY:
            u = rclamp(u, 0.0f, 1.0f);
            v = rclamp(v, 0.0f, 1.0f);
            break;
        case RXR_REPEAT_REPEAT_XY:
            u = u - std::floor(u);
            v = v - std::floor(v);
            break;
        case RXR_REPEAT_REPEAT_X:
            u = u - std::floor(u);
            v = rclamp(v, 0.0f, 1.0f);
            break;
        case RXR_REPEAT_REPEAT_Y:
            u = rclamp(u, 0.0f, 1.0f);
            v = v - std::floor(v);
            break;
    }
    if (sample_mode == RXR_SAMPLE_NEAREST)
        texture_sample_nearest(t, u, v, out);
    else
        texture_sample_linear(t, u, v, out);
}

// ---- src/map/light.rs:656-677 ----------------------------------------------------------------
static void apply_flicker(const CompiledLight &l, const float color[3], float intensity, float flicker, uint32_t hash,
                          float out[3]) {
    float flicker_factor;
    if (flicker > 0.0f) {
        uint32_t combined_hash =
            hash + (sat_u32(l.position[0]) + sat_u32(l.position[1]) + sat_u32(l.position[2])) * 100u;
        float flicker_value = rclamp((float)combined_hash / (float)UINT32_MAX, 0.0f, 1.0f);
        flicker_factor = 1.0f - flicker_value * flicker;
    } else {
        flicker_factor = 1.0f;
    }
    out[0] = color[0] * intensity * flicker_factor;
    out[1] = color[1] * intensity * flicker_factor;
    out[2] = color[2] * intensity * flicker_factor;
}
static float smoothstep(float edge0, float edge1, float x) {
    float t = rclamp((x - edge0) / (edge1 - edge0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static Vec3 lpos(const CompiledLight &l) { return {l.position[0], l.position[1], l.position[2]}; }

// ---- src/map/light.rs:535-552 ----------------------------------------------------------------
static bool calculate_point_light(const CompiledLight &l, Vec3 point, uint32_t hash, float out[3]) {
    float distance = rvek::magnitude(point - lpos(l));
    if (distance >= l.end_distance) return false;
    if (distance <= l.start_distance) {
        apply_flicker(l, l.color, l.intensity, l.flicker, hash, out);
        return true;
    }
    float attenuation = smoothstep(l.end_distance, l.start_distance, distance);
    float adjusted_intensity = l.intensity * attenuation;
    apply_flicker(l, l.color, adjusted_intensity, l.flicker, hash, out);
    return true;
}
// ---- :559-580 ---------------------------------------------------------------------------------
static bool calculate_spot_light(const CompiledLight &l, Vec3 point, uint32_t hash, float out[3]) {
    float distance = rvek::magnitude(point - lpos(l));
    if (distance >= l.end_distance) return false;
    float attenuation = (distance <= l.start_distance)
                            ? 1.0f
                            : 1.0f - ((distance - l.start_distance) / (l.end_distance - l.start_distance));
    Vec3 direction_to_point = rvek::normalized(point - lpos(l));
    Vec3 dir{l.direction[0], l.direction[1], l.direction[2]};
    float angle = std::acos(rvek::dot(dir, direction_to_point));
    if (angle > l.cone_angle) return false;
    float adjusted_intensity = l.intensity * attenuation;
    apply_flicker(l, l.color, adjusted_intensity, l.flicker, hash, out);
    return true;
}
// ---- :582-629 ---------------------------------------------------------------------------------
static bool calculate_area_light(const CompiledLight &l, Vec3 point, bool d2, float out[3]) {
    Vec3 to_point = point - lpos(l);
    float distance = rvek::magnitude(to_point);
    if (distance >= l.end_distance) return false;
    if (distance < 0.1f) {
        out[0] = l.color[0]; out[1] = l.color[1]; out[2] = l.color[2];
        return true;
    }
    float distance_attenuation =
        (distance <= l.start_distance) ? 1.0f : smoothstep(l.end_distance, l.start_distance, distance);
    float area = l.width * l.height;
    Vec3 direction = rvek::normalized(to_point);
    float attenuation;
    if (l.from_linedef) {
        attenuation = distance_attenuation * area * l.intensity;
    } else if (d2) {
        float distance_x = std::fabs(to_point.x / (l.width * 0.5f));
        float distance_y = std::fabs(to_point.y / (l.height * 0.5f));
        float attenuation_x = rmax(1.0f - distance_x, 0.0f);
        float attenuation_y = rmax(1.0f - distance_y, 0.0f);
        attenuation = attenuation_x * attenuation_y * distance_attenuation * l.intensity;
    } else {
        Vec3 n{l.normal[0], l.normal[1], l.normal[2]};
        float angle_attenuation = rmax(rvek::dot(n, direction), 0.0f);
        attenuation = angle_attenuation * distance_attenuation * area * l.intensity;
    }
    out[0] = l.color[0] * attenuation;
    out[1] = l.color[1] * attenuation;
    out[2] = l.color[2] * attenuation;
    return true;
}
// ---- :631-653 ---------------------------------------------------------------------------------
static bool calculate_daylight_light(const CompiledLight &l, Vec3 point, float out[3]) {
    Vec3 to_point = point - lpos(l);
    float distance = rvek::magnitude(to_point);
    if (distance >= l.end_distance) return false;
    Vec3 direction = rvek::normalized(to_point);
    Vec3 n{l.normal[0], l.normal[1], l.normal[2]};
    float angle_attenuation = rmax(rvek::dot(n, direction), 0.0f);
    float distance_attenuation =
        (distance <= l.start_distance) ? 1.0f : smoothstep(l.end_distance, l.start_distance, distance);
    float attenuation = angle_attenuation * distance_attenuation * l.intensity;
    out[0] = l.color[0] * attenuation;
    out[1] = l.color[1] * attenuation;
    out[2] = l.color[2] * attenuation;
    return true;
}

// ---- src/map/light.rs:491-502 ----------------------------------------------------------------
bool light_color_at(const CompiledLight &l, Vec3 point, uint32_t hash, bool d2, float out[3]) {
    if (!l.emitting) return false;
    switch (l.light_type) {
        case RXR_LIGHT_POINT: return calculate_point_light(l, point, hash, out);
        case RXR_LIGHT_AMBIENT:
        case RXR_LIGHT_AMBIENT_DAYLIGHT:
            apply_flicker(l, l.color, l.intensity, l.flicker, hash, out);  // :554-557
            return true;
        case RXR_LIGHT_SPOT: return calculate_spot_light(l, point, hash, out);
        case RXR_LIGHT_AREA: return calculate_area_light(l, point, d2, out);
        case RXR_LIGHT_DAYLIGHT: return calculate_daylight_light(l, point, out);
    }
    return false;
}

// ---- src/map/light.rs:504-533 ----------------------------------------------------------------
bool light_radiance_at(const CompiledLight &l, Vec3 point, bool has_n, Vec3 n, uint32_t hash, Vec3 &out) {
    float c[3];
    if (!light_color_at(l, point, hash, false, c)) return false;
    Vec3 incoming{c[0], c[1], c[2]};
    if (l.light_type == RXR_LIGHT_AMBIENT || l.light_type == RXR_LIGHT_AMBIENT_DAYLIGHT ||
        l.light_type == RXR_LIGHT_DAYLIGHT) {
        out = incoming;
        return true;
    }
    if (!has_n) {
        out = incoming;
        return true;
    }
    Vec3 dir_to_light = rvek::normalized(lpos(l) - point);
    float lambert = rmax(rvek::dot(n, dir_to_light), 0.0f);
    out = incoming * lambert;
    return true;
}

// ---- src/map/mini.rs:58-66, src/chunk.rs:154-161, src/map/bbox.rs:35-40 ----------------------
float mapmini_get_occlusion(const std::vector<Occluder> &occ, Vec2 at) {
    for (const Occluder &o : occ) {
        if (at.x >= o.min.x && at.x <= o.max.x && at.y >= o.min.y && at.y <= o.max.y) return o.occlusion;
    }
    return 1.0f;
}
// ---- src/map/mini.rs:68-86 -------------------------------------------------------------------
static bool segments_intersect(Vec2 a1, Vec2 a2, Vec2 b1, Vec2 b2) {
    float d = (a2.x - a1.x) * (b2.y - b1.y) - (a2.y - a1.y) * (b2.x - b1.x);
    if (d == 0.0f) return false;
    float u = ((b1.x - a1.x) * (b2.y - b1.y) - (b1.y - a1.y) * (b2.x - b1.x)) / d;
    float v = ((b1.x - a1.x) * (a2.y - a1.y) - (b1.y - a1.y) * (a2.x - a1.x)) / d;
    return (u >= 0.0f && u <= 1.0f) && (v >= 0.0f && v <= 1.0f);
}
// ---- src/map/mini.rs:88-95 -------------------------------------------------------------------
bool mapmini_is_visible(const MapMini &m, Vec2 from, Vec2 to) {
    for (const Linedef &l : m.linedefs) {
        if (segments_intersect(from, to, l.start, l.end)) return false;
    }
    return true;
}

// ---- src/batch/batch3d.rs:140-229 ------------------------------------------------------------
Batch3D batch3d_from_box(float x, float y, float z, float width, float height, float depth) {
    Batch3D b;
    auto V = [&](float a, float bb, float c) { b.vertices.push_back({a, bb, c, 1.0f}); };
    // Front face
    V(x, y, z); V(x + width, y, z); V(x + width, y + height, z); V(x, y + height, z);
    // Back face
    V(x, y, z + depth); V(x + width, y, z + depth); V(x + width, y + height, z + depth); V(x, y + height, z + depth);
    // Left face
    V(x, y, z); V(x, y + height, z); V(x, y + height, z + depth); V(x, y, z + depth);
    // Right face
    V(x + width, y, z); V(x + width, y + height, z); V(x + width, y + height, z + depth); V(x + width, y, z + depth);
    // Top face
    V(x, y + height, z); V(x + width, y + height, z); V(x + width, y + height, z + depth); V(x, y + height, z + depth);
    // Bottom face
    V(x, y, z); V(x + width, y, z); V(x + width, y, z + depth); V(x, y, z + depth);
    const size_t idx[12][3] = {{0, 1, 2},    {0, 2, 3},    {4, 6, 5},    {4, 7, 6},    {8, 9, 10},   {8, 10, 11},
                               {12, 14, 13}, {12, 15, 14}, {16, 17, 18}, {16, 18, 19}, {20, 23, 22}, {20, 22, 21}};
    for (auto &t : idx) b.indices.push_back({t[0], t[1], t[2]});
    for (int f = 0; f < 6; ++f) {
        b.uvs.push_back({0.0f, 1.0f});
        b.uvs.push_back({1.0f, 1.0f});
        b.uvs.push_back({1.0f, 0.0f});
        b.uvs.push_back({0.0f, 0.0f});
    }
    return b;
}

// ---- src/batch/batch3d.rs:238-253 ------------------------------------------------------------
void batch3d_add(Batch3D &b, const float *verts4, size_t nv, const uint32_t *idx3, size_t nt, const float *uvs2) {
    size_t base_index = b.vertices.size();
    for (size_t i = 0; i < nv; ++i) b.vertices.push_back({verts4[i * 4], verts4[i * 4 + 1], verts4[i * 4 + 2], verts4[i * 4 + 3]});
    for (size_t i = 0; i < nv; ++i) b.uvs.push_back({uvs2[i * 2], uvs2[i * 2 + 1]});
    for (size_t i = 0; i < nt; ++i)
        b.indices.push_back({idx3[i * 3] + base_index, idx3[i * 3 + 1] + base_index, idx3[i * 3 + 2] + base_index});
}

// ---- src/batch/batch3d.rs:771-809 (== with_computed_normals, :812-842) -----------------------
void batch3d_compute_vertex_normals(Batch3D &b) {
    b.normals.assign(b.vertices.size(), Vec3{0, 0, 0});
    std::vector<uint32_t> counts(b.vertices.size(), 0);
    for (auto &t : b.indices) {
        Vec3 p0{b.vertices[t[0]][0], b.vertices[t[0]][1], b.vertices[t[0]][2]};
        Vec3 p1{b.vertices[t[1]][0], b.vertices[t[1]][1], b.vertices[t[1]][2]};
        Vec3 p2{b.vertices[t[2]][0], b.vertices[t[2]][1], b.vertices[t[2]][2]};
        Vec3 normal = rvek::normalized(rvek::cross(p1 - p0, p2 - p0));
        b.normals[t[0]] += normal;
        b.normals[t[1]] += normal;
        b.normals[t[2]] += normal;
        counts[t[0]] += 1;
        counts[t[1]] += 1;
        counts[t[2]] += 1;
    }
    for (size_t i = 0; i < b.normals.size(); ++i) {
        if (counts[i] > 0) {
            b.normals[i] = b.normals[i] / (float)counts[i];
            b.normals[i] = rvek::normalized(b.normals[i]);
        }
    }
}

// ---- src/batch/batch3d.rs:742-746 ------------------------------------------------------------
static bool is_front_facing(const std::array<float, 4> &v0, const std::array<float, 4> &v1, const std::array<float, 4> &v2) {
    float orientation = (v1[0] - v0[0]) * (v2[1] - v0[1]) - (v1[1] - v0[1]) * (v2[0] - v0[0]);
    return orientation > 0.0f;
}

// ---- src/batch/batch3d.rs:482-740 ------------------------------------------------------------
bool batch3d_clip_and_project(Batch3D &b, const Mat4 &view_matrix, const Mat4 &projection_matrix, float viewport_width,
                              float viewport_height) {
    Mat4 mvp = (projection_matrix * view_matrix) * b.transform_3d;  // :490

    if (!b.vertices.empty()) {  // :493-552
        float min_x = INFINITY, min_y = INFINITY, min_z = INFINITY;
        float max_x = -INFINITY, max_y = -INFINITY, max_z = -INFINITY;
        for (auto &v : b.vertices) {
            min_x = rmin(min_x, v[0]); min_y = rmin(min_y, v[1]); min_z = rmin(min_z, v[2]);
            max_x = rmax(max_x, v[0]); max_y = rmax(max_y, v[1]); max_z = rmax(max_z, v[2]);
        }
        const float corners[8][4] = {{min_x, min_y, min_z, 1.0f}, {min_x, min_y, max_z, 1.0f}, {min_x, max_y, min_z, 1.0f},
                                     {min_x, max_y, max_z, 1.0f}, {max_x, min_y, min_z, 1.0f}, {max_x, min_y, max_z, 1.0f},
                                     {max_x, max_y, min_z, 1.0f}, {max_x, max_y, max_z, 1.0f}};
        bool outside_left = true, outside_right = true, outside_bottom = true, outside_top = true, outside_near = true,
             outside_far = true;
        for (auto &c : corners) {
            Vec4 v = mvp * Vec4{c[0], c[1], c[2], c[3]};
            float w = v.w;
            outside_left &= v.x < -w;
            outside_right &= v.x > w;
            outside_bottom &= v.y < -w;
            outside_top &= v.y > w;
            outside_near &= v.z < -w;
            outside_far &= v.z > w;
        }
        if (outside_left || outside_right || outside_bottom || outside_top || outside_near || outside_far) {
            b.projected_vertices.clear();
            b.clipped_indices.clear();
            b.clipped_uvs.clear();
            b.clipped_normals.clear();
            b.edges.clear();
            b.has_bounding_box = false;
            return true;
        }
    }

    Mat4 view_model = view_matrix * b.transform_3d;  // :555
    std::vector<std::array<float, 4>> view_space_vertices;
    view_space_vertices.reserve(b.vertices.size());
    for (auto &v : b.vertices) {
        Vec4 r = view_model * Vec4{v[0], v[1], v[2], v[3]};
        view_space_vertices.push_back({r.x, r.y, r.z, r.w});
    }

    const float near_plane = 0.1f;  // :563

    b.clipped_indices = b.indices;  // :566-574
    b.clipped_uvs = b.uvs;
    b.clipped_normals = b.normals;

    std::vector<std::array<float, 4>> new_vertices;
    std::vector<std::array<float, 2>> new_uvs;
    std::vector<Vec3> new_normals;

    std::vector<bool> edge_visibility(b.indices.size(), true);  // :582-583

    for (size_t triangle_idx = 0; triangle_idx < b.indices.size(); ++triangle_idx) {  // :586
        size_t i0 = b.indices[triangle_idx][0], i1 = b.indices[triangle_idx][1], i2 = b.indices[triangle_idx][2];
        auto v0 = view_space_vertices[i0];
        auto v1 = view_space_vertices[i1];
        auto v2 = view_space_vertices[i2];

        if (b.cull_mode != CullOff) {  // :592-600
            float orient = (v1[0] - v0[0]) * (v2[1] - v0[1]) - (v1[1] - v0[1]) * (v2[0] - v0[0]);
            bool is_front = orient > 0.0f;
            if (b.cull_mode == CullBack && is_front) continue;
            if (b.cull_mode == CullFront && !is_front) continue;
        }

        if (b.normals.size() <= i0 || b.normals.size() <= i1 || b.normals.size() <= i2) return false;  // :605-607 panics
        auto uv0 = b.uvs[i0], uv1 = b.uvs[i1], uv2 = b.uvs[i2];
        Vec3 n0 = b.normals[i0], n1 = b.normals[i1], n2 = b.normals[i2];

        bool is_v0_inside = v0[2] < -near_plane;
        bool is_v1_inside = v1[2] < -near_plane;
        bool is_v2_inside = v2[2] < -near_plane;

        if (is_v0_inside && is_v1_inside && is_v2_inside) continue;  // :613-616

        edge_visibility[triangle_idx] = false;  // :618

        if (!is_v0_inside && !is_v1_inside && !is_v2_inside) continue;  // :620-623

        struct VR { std::array<float, 4> v; std::array<float, 2> uv; Vec3 n; };
        VR vertices[3] = {{v0, uv0, n0}, {v1, uv1, n1}, {v2, uv2, n2}};
        std::vector<size_t> clipped_indices;
        std::vector<bool> new_edge_visibility;

        for (int i = 0; i < 3; ++i) {  // :630-669
            auto current = vertices[i].v;
            auto uv_current = vertices[i].uv;
            Vec3 n_current = vertices[i].n;
            auto next = vertices[(i + 1) % 3].v;
            auto uv_next = vertices[(i + 1) % 3].uv;
            Vec3 n_next = vertices[(i + 1) % 3].n;

            if (current[2] < -near_plane) {
                new_vertices.push_back(current);
                new_uvs.push_back(uv_current);
                new_normals.push_back(n_current);
                clipped_indices.push_back(b.vertices.size() + new_vertices.size() - 1);
                new_edge_visibility.push_back(true);
            }

            if ((current[2] < -near_plane) != (next[2] < -near_plane)) {
                float t = (-near_plane - current[2]) / (next[2] - current[2]);
                std::array<float, 4> intersection = {
                    current[0] + t * (next[0] - current[0]), current[1] + t * (next[1] - current[1]),
                    current[2] + t * (next[2] - current[2]), current[3] + t * (next[3] - current[3])};
                std::array<float, 2> interpolated_uv = {uv_current[0] + t * (uv_next[0] - uv_current[0]),
                                                        uv_current[1] + t * (uv_next[1] - uv_current[1])};
                Vec3 interpolated_normal = rvek::normalized(n_current * (1.0f - t) + n_next * t);
                new_vertices.push_back(intersection);
                new_uvs.push_back(interpolated_uv);
                new_normals.push_back(interpolated_normal);
                clipped_indices.push_back(b.vertices.size() + new_vertices.size() - 1);
                new_edge_visibility.push_back(true);
            }
        }

        for (size_t i = 1; i + 1 < clipped_indices.size(); ++i)  // :672-678
            b.clipped_indices.push_back({clipped_indices[0], clipped_indices[i], clipped_indices[i + 1]});

        edge_visibility.insert(edge_visibility.end(), new_edge_visibility.begin(), new_edge_visibility.end());  // :680
    }

    view_space_vertices.insert(view_space_vertices.end(), new_vertices.begin(), new_vertices.end());  // :684-686
    b.clipped_uvs.insert(b.clipped_uvs.end(), new_uvs.begin(), new_uvs.end());
    b.clipped_normals.insert(b.clipped_normals.end(), new_normals.begin(), new_normals.end());

    b.projected_vertices.clear();  // :689-700
    b.projected_vertices.reserve(view_space_vertices.size());
    for (auto &v : view_space_vertices) {
        Vec4 result = projection_matrix * Vec4{v[0], v[1], v[2], v[3]};
        float w = result.w;
        b.projected_vertices.push_back({((result.x / w) * 0.5f + 0.5f) * viewport_width,
                                        ((-result.y / w) * 0.5f + 0.5f) * viewport_height, result.z / w, w});
    }

    {  // :703, :749-768
        float min_x = INFINITY, max_x = -INFINITY, min_y = INFINITY, max_y = -INFINITY;
        for (auto &v : b.projected_vertices) {
            min_x = rmin(min_x, v[0]);
            max_x = rmax(max_x, v[0]);
            min_y = rmin(min_y, v[1]);
            max_y = rmax(max_y, v[1]);
        }
        b.has_bounding_box = true;
        b.bounding_box = Rect{min_x, min_y, max_x - min_x, max_y - min_y};
    }

    b.edges.clear();  // :706-739
    b.edges.reserve(b.clipped_indices.size());
    for (size_t triangle_idx = 0; triangle_idx < b.clipped_indices.size(); ++triangle_idx) {
        auto &t = b.clipped_indices[triangle_idx];
        auto v0 = b.projected_vertices[t[0]];
        auto v1 = b.projected_vertices[t[1]];
        auto v2 = b.projected_vertices[t[2]];
        bool visible;
        switch (b.cull_mode) {
            case CullOff:
                if (is_front_facing(v0, v1, v2)) std::swap(v1, v2);
                visible = true;
                break;
            case CullFront: visible = !is_front_facing(v0, v1, v2); break;
            default:  // CullBack
                if (is_front_facing(v0, v1, v2)) {
                    std::swap(v1, v2);
                    visible = true;
                } else {
                    visible = false;
                }
        }
        bool ev = triangle_idx < edge_visibility.size() ? (bool)edge_visibility[triangle_idx] : true;
        bool edge_visible = ev && visible;
        const float a[3][2] = {{v0[0], v0[1]}, {v1[0], v1[1]}, {v2[0], v2[1]}};
        const float c[3][2] = {{v1[0], v1[1]}, {v2[0], v2[1]}, {v0[0], v0[1]}};
        b.edges.push_back(edges_new(a, c, edge_visible));
    }
    return true;
}

// ---- src/batch/batch2d.rs:109-127 ------------------------------------------------------------
Batch2D batch2d_from_rectangle(float x, float y, float width, float height) {
    Batch2D b;
    b.vertices = {{x, y}, {x, y + height}, {x + width, y + height}, {x + width, y}};
    b.indices = {{0, 1, 2}, {0, 2, 3}};
    b.uvs = {{0.0f, 0.0f}, {0.0f, 1.0f}, {1.0f, 1.0f}, {1.0f, 0.0f}};
    return b;
}

// ---- src/batch/batch2d.rs:373-425 ------------------------------------------------------------
void batch2d_project(Batch2D &b, const Mat3 *matrix) {
    b.projected_vertices.clear();
    float min_x = INFINITY, max_x = -INFINITY, min_y = INFINITY, max_y = -INFINITY;
    for (auto &v : b.vertices) {
        std::array<float, 2> p;
        if (matrix) {
            Vec3 r = (*matrix) * Vec3{v[0], v[1], 1.0f};
            p = {r.x, r.y};
        } else {
            p = v;
        }
        min_x = rmin(min_x, p[0]);
        max_x = rmax(max_x, p[0]);
        min_y = rmin(min_y, p[1]);
        max_y = rmax(max_y, p[1]);
        b.projected_vertices.push_back(p);
    }
    b.has_bounding_box = true;
    b.bounding_box = Rect{min_x, min_y, max_x - min_x, max_y - min_y};
    b.edges.clear();
    for (auto &t : b.indices) {
        auto v0 = b.projected_vertices[t[0]];
        auto v1 = b.projected_vertices[t[1]];
        auto v2 = b.projected_vertices[t[2]];
        const float a[3][2] = {{v0[0], v0[1]}, {v1[0], v1[1]}, {v2[0], v2[1]}};
        const float c[3][2] = {{v1[0], v1[1]}, {v2[0], v2[1]}, {v0[0], v0[1]}};
        b.edges.push_back(edges_new(a, c, true));
    }
}

// ---- src/wavefront.rs:34-102 (v / vt / f only matter; vn is parsed and dropped) ---------------
Batch3D batch3d_from_obj(const char *text) {
    Batch3D b;
    std::vector<std::array<float, 2>> texture_coords;
    const char *p = text;
    while (*p) {
        const char *e = p;
        while (*e && *e != '\n') ++e;
        std::string line(p, e);
        p = *e ? e + 1 : e;
        size_t s = line.find_first_not_of(" \t\r");
        if (s == std::string::npos) continue;
        size_t t = line.find_last_not_of(" \t\r");
        line = line.substr(s, t - s + 1);
        if (line.empty() || line[0] == '#') continue;
        if (line.rfind("v ", 0) == 0) {
            float x = 0, y = 0, z = 0;
            sscanf(line.c_str() + 2, "%f %f %f", &x, &y, &z);
            b.vertices.push_back({x, y, z, 1.0f});
        } else if (line.rfind("vt ", 0) == 0) {
            float u = 0, v = 0;
            sscanf(line.c_str() + 3, "%f %f", &u, &v);
            texture_coords.push_back({u, v});
        } else if (line.rfind("f ", 0) == 0) {
            char a0[64], a1[64], a2[64];
            if (sscanf(line.c_str() + 2, "%63s %63s %63s", a0, a1, a2) == 3) {
                auto parse_face = [](const char *s) -> size_t { return (size_t)strtoull(s, nullptr, 10) - 1; };
                b.indices.push_back({parse_face(a0), parse_face(a1), parse_face(a2)});
            }
        }
    }
    if (texture_coords.empty()) {
        for (auto &v : b.vertices) b.uvs.push_back({v[0], v[1]});  // :92-95
    } else {
        b.uvs = texture_coords;
    }
    return b;
}

// ---- src/rasterizer.rs:92-152 ----------------------------------------------------------------
Rasterizer rasterizer_setup(const Mat3 *m2d, const Mat4 &view, const Mat4 &proj) {
    Rasterizer r;
    r.inverse_view_matrix = rvek::inverted(view);
    r.camera_pos = Vec3{r.inverse_view_matrix.m[12], r.inverse_view_matrix.m[13], r.inverse_view_matrix.m[14]};
    if (m2d) {
        r.has_m2d = true;
        r.projection_matrix_2d = *m2d;
        r.translationd2.x = m2d->at(0, 2);
        r.translationd2.y = m2d->at(1, 2);
        r.scaled2 = m2d->at(0, 0);
    }
    r.inverse_projection_matrix = rvek::inverted(proj);
    r.view_matrix = view;
    r.projection_matrix = proj;
    return r;
}

// ---- src/scene.rs:154-200 --------------------------------------------------------------------
bool scene_project(Scene &s, const Mat3 *m2d, const Mat4 &view, const Mat4 &proj, float w, float h) {
    bool ok = true;
    for (Chunk &c : s.chunks) {
        for (auto &b : c.batches2d) batch2d_project(b, m2d);
        for (auto &b : c.terrain_batch2d) batch2d_project(b, m2d);
        for (auto &b : c.batches3d_opacity) ok &= batch3d_clip_and_project(b, view, proj, w, h);
        for (auto &b : c.batches3d) ok &= batch3d_clip_and_project(b, view, proj, w, h);
        for (auto &b : c.terrain_batch3d) ok &= batch3d_clip_and_project(b, view, proj, w, h);
    }
    for (auto &b : s.d2_static) batch2d_project(b, m2d);
    for (auto &b : s.d2_dynamic) batch2d_project(b, m2d);
    for (auto &b : s.d3_static) ok &= batch3d_clip_and_project(b, view, proj, w, h);
    for (auto &b : s.d3_dynamic) ok &= batch3d_clip_and_project(b, view, proj, w, h);
    for (auto &b : s.d3_overlay) ok &= batch3d_clip_and_project(b, view, proj, w, h);
    return ok;
}

// ---- cameras ----------------------------------------------------------------------------------
// src/camera/d3orbit.rs:186-195 (eye_position), :50-56
void orbit_camera(Vec3 center, float distance, float azimuth, float elevation, float fov, float near, float far, float w,
                  float h, Mat4 &view, Mat4 &proj) {
    float x = distance * std::cos(azimuth) * std::cos(elevation);
    float y = distance * std::sin(elevation);
    float z = distance * std::sin(azimuth) * std::cos(elevation);
    Vec3 position = Vec3{x, y, z} + center;
    view = rvek::look_at_rh(position, center, Vec3{0, 1, 0});
    // f32::to_radians(): self * (PI / 180.0)
    proj = rvek::perspective_fov_rh_zo(fov * (3.14159265358979323846f / 180.0f), w, h, near, far);
}
// src/camera/d3firstp.rs:36-42
void firstp_camera(Vec3 position, Vec3 center, float fov, float near, float far, float w, float h, Mat4 &view,
                   Mat4 &proj) {
    view = rvek::look_at_rh(position, center, Vec3{0, 1, 0});
    proj = rvek::perspective_fov_rh_zo(fov * (3.14159265358979323846f / 180.0f), w, h, near, far);
}

// =================================================================================================
// The raster loops
// =================================================================================================

struct TileRect {  // src/rasterizer.rs:2013-2019
    size_t x, y, width, height;
};

// One rusteria Execution per tile (:310), shared by every fragment of the tile: state a shader leaves
// behind (emissive, globals, the components the raster loops do not overwrite) leaks into the following
// fragments exactly as in the reference.
using vm::Execution;

struct FrameCtx {
    const Rasterizer *r;
    const Scene *scene;
    const Assets *assets;
    std::vector<const CompiledLight *> lights;  // scene.lights.iter().chain(&scene.dynamic_lights)
    bool any_lights;
};

// src/rasterizer.rs:1731-1773 (2D and 3D variants are the same arithmetic on x,y)
static inline void barycentric_weights(const float *a, const float *b, const float *c, const float p[2], float out[3]) {
    float ac[2] = {c[0] - a[0], c[1] - a[1]};
    float ab[2] = {b[0] - a[0], b[1] - a[1]};
    float ap[2] = {p[0] - a[0], p[1] - a[1]};
    float pc[2] = {c[0] - p[0], c[1] - p[1]};
    float pb[2] = {b[0] - p[0], b[1] - p[1]};
    float area = ac[0] * ab[1] - ac[1] * ab[0];
    float alpha = (pc[0] * pb[1] - pc[1] * pb[0]) / area;
    float beta = (ac[0] * ap[1] - ac[1] * ap[0]) / area;
    float gamma = 1.0f - alpha - beta;
    out[0] = alpha;
    out[1] = beta;
    out[2] = gamma;
}

// src/rasterizer.rs:1707-1727
static inline Vec3 screen_to_world(const Rasterizer &r, float x, float y, float z_ndc) {
    float x_ndc = 2.0f * (x / r.width) - 1.0f;
    float y_ndc = 1.0f - 2.0f * (y / r.height);
    Vec4 ndc{x_ndc, y_ndc, z_ndc, 1.0f};
    Vec4 view_space = r.inverse_projection_matrix * ndc;
    view_space = view_space / view_space.w;
    Vec4 world_space = r.inverse_view_matrix * view_space;
    return Vec3{world_space.x, world_space.y, world_space.z};
}

// src/rasterizer.rs:1841-1869
static inline void screen_ray(const Rasterizer &r, float x, float y, Vec3 &origin, Vec3 &dir) {
    float ndc_x = 2.0f * (x / r.width) - 1.0f;
    float ndc_y = 1.0f - 2.0f * (y / r.height);
    Vec4 ndc_near{ndc_x, ndc_y, -1.0f, 1.0f}, ndc_far{ndc_x, ndc_y, 1.0f, 1.0f};
    Vec4 view_near = r.inverse_projection_matrix * ndc_near;
    Vec4 view_far = r.inverse_projection_matrix * ndc_far;
    view_near = view_near / view_near.w;
    view_far = view_far / view_far.w;
    Vec4 world_near = r.inverse_view_matrix * view_near;
    Vec4 world_far = r.inverse_view_matrix * view_far;
    origin = Vec3{world_near.x, world_near.y, world_near.z};
    Vec3 target{world_far.x, world_far.y, world_far.z};
    dir = rvek::normalized(target - origin);
}

// src/rasterizer.rs:1875-1951
static inline Vec3 shade_fast_brdf(Vec3 base_color, float roughness, float metallic, Vec3 emissive, Vec3 n, Vec3 v,
                                   Vec3 l, Vec3 light_radiance) {
    float n_dot_l = rmax(rvek::dot(n, l), 0.0f);
    if (n_dot_l <= 0.0f) return emissive;
    Vec3 f0 = rvek::lerp(Vec3{0.04f, 0.04f, 0.04f}, base_color, metallic);
    Vec3 kd = base_color * (1.0f - metallic);
    kd = kd * (1.0f - rmax(f0.x, rmax(f0.y, f0.z)));
    // roughness_to_shininess
    float a = rmax(roughness * roughness, 1e-4f);
    float shininess = rclamp(2.0f / a - 2.0f, 1.0f, 2048.0f);
    // blinn_phong_spec
    Vec3 h = rvek::normalized(l + v);
    float n_dot_h = rmax(rvek::dot(n, h), 0.0f);
    float spec_b = (n_dot_h <= 0.0f) ? 0.0f : std::exp2(shininess * std::log2(n_dot_h));  // pow32_fast
    float n_dot_v = rmax(rvek::dot(n, v), 0.0f);
    // schlick_fresnel
    float one_minus = 1.0f - rclamp(n_dot_v, 0.0f, 1.0f);
    float x = one_minus * one_minus * one_minus * one_minus * one_minus;
    Vec3 f = f0 + (Vec3{1.0f, 1.0f, 1.0f} - f0) * x;
    Vec3 diffuse = kd * n_dot_l;
    Vec3 specular = f * spec_b * n_dot_l;
    return (diffuse + specular) * light_radiance + emissive;
}

// texel switch shared by the three loops; `missing` is what the loop returns for unknown sources.
// 3D: src/rasterizer.rs:1101-1222 (unknown -> [0,0,0,255]); 2D: :672-758 (unknown -> [0,0,0,0]).
// Returns false where the reference would panic (3D tile_list[index] unchecked, :1103).
// `world3`: the fragment's world position in the two 3D loops (the terrain brush preview measures its distance from it,
// :1192-1213, :1601-1622), nullptr in the 2D loop (no brush preview there, :749-751)
static inline bool fetch_texel(const FrameCtx &fc, const Source &src, int repeat_mode, float u, float v, bool is_3d,
                               const Chunk *chunk, Vec2 terrain_pos, uint8_t texel[4], const Vec3 *world3 = nullptr) {
    const bool in_chunk = chunk != nullptr;
    const Rasterizer &r = *fc.r;
    auto zero = [&]() { texel[0] = texel[1] = texel[2] = texel[3] = 0; };
    switch (src.kind) {
        case RXR_SOURCE_STATIC_TILE: {
            if (src.index >= fc.assets->tile_list.size()) {
                if (is_3d) return false;
                zero();
                return true;
            }
            const Tile &textile = fc.assets->tile_list[src.index];
            if (textile.textures.empty()) return false;  // `% 0` panics
            size_t index = fc.scene->animation_frame % textile.textures.size();
            texture_sample(textile.textures[index], u, v, r.sample_mode, repeat_mode, texel);
            return true;
        }
        case RXR_SOURCE_DYNAMIC_TILE: {
            if (src.index >= fc.scene->dynamic_textures.size()) {
                if (is_3d) return false;
                zero();
                return true;
            }
            const Tile &textile = fc.scene->dynamic_textures[src.index];
            if (textile.textures.empty()) return false;
            size_t index = fc.scene->animation_frame % textile.textures.size();
            texture_sample(textile.textures[index], u, v, r.sample_mode, repeat_mode, texel);
            return true;
        }
        case RXR_SOURCE_PIXEL:
            memcpy(texel, src.pixel, 4);
            return true;
        case RXR_HOST_SOURCE_ENTITY_TILE:
        case RXR_HOST_SOURCE_ITEM_TILE: {
            // PixelSource::EntityTile(id, index) / ItemTile(id, index): src/rasterizer.rs:1140-1187 (3D), :705-748 (2D),
            // :1548-1595 (opacity pass) -- the same in all three: assets.entity_tiles.get(&id) -> .get_index(index) ->
            // textures[animation_frame % len].sample(..); either lookup failing gives [0, 0, 0, 0]
            const auto &tiles = src.kind == RXR_HOST_SOURCE_ENTITY_TILE ? fc.assets->entity_tiles : fc.assets->item_tiles;
            const auto it = tiles.find(src.index);
            if (it == tiles.end() || src.seq >= it->second.size()) {
                zero();
                return true;
            }
            const Tile &textile = it->second[src.seq];
            if (textile.textures.empty()) return false;  // `% 0` panics
            size_t index = fc.scene->animation_frame % textile.textures.size();
            texture_sample(textile.textures[index], u, v, r.sample_mode, repeat_mode, texel);
            return true;
        }
        case RXR_SOURCE_MISSING:
            zero();
            return true;
        case RXR_SOURCE_TERRAIN:
            if (in_chunk) {
                // 3D: chunk.sample_terrain_texture(world_2d, Vec2::one()), :1191; 2D: (world, Vec2::one()), :751
                if (!chunk_sample_terrain_texture(*chunk, terrain_pos, Vec2{1.0f, 1.0f}, texel)) return false;
                if (r.has_brush_preview && world3) {  // :1193-1212
                    float dist = rvek::magnitude(*world3 - r.brush_position);
                    if (dist < r.brush_radius) {
                        float normalized = dist / r.brush_radius;
                        float falloff = rclamp(r.brush_falloff, 0.001f, 1.0f);
                        float fade = rclamp((1.0f - normalized) / falloff, 0.0f, 1.0f);
                        float blend = 0.2f + 0.6f * fade;
                        for (int ch = 0; ch < 3; ++ch) texel[ch] = sat_u8(rmin((float)texel[ch] * (1.0f - blend) + 255.0f * blend, 255.0f));
                    }
                }
            } else if (is_3d) {
                texel[0] = 255; texel[1] = 0; texel[2] = 0; texel[3] = 255;  // :1218
            } else {
                zero();  // :753-755
            }
            return true;
        default:
            if (is_3d) {
                texel[0] = texel[1] = texel[2] = 0;
                texel[3] = 255;  // :1221
            } else {
                zero();  // :757
            }
            return true;
    }
}

// the triangle bounding-box clamp, src/rasterizer.rs:998-1017 (== :615-634, :1458-1477)
static inline void tri_bounds(const float *v0, const float *v1, const float *v2, const TileRect &tile, size_t &min_x,
                              size_t &max_x, size_t &min_y, size_t &max_y) {
    float min_xf = rmin(v0[0], rmin(v1[0], v2[0]));
    float max_xf = rmax(v0[0], rmax(v1[0], v2[0]));
    float min_yf = rmin(v0[1], rmin(v1[1], v2[1]));
    float max_yf = rmax(v0[1], rmax(v1[1], v2[1]));
    min_x = (size_t)sat_usize(rmax(std::floor(min_xf), (float)tile.x));
    max_x = (size_t)sat_usize(rmin(std::ceil(max_xf), (float)(tile.x + tile.width)));
    min_y = (size_t)sat_usize(rmax(std::floor(min_yf), (float)tile.y));
    max_y = (size_t)sat_usize(rmin(std::ceil(max_yf), (float)(tile.y + tile.height)));
}

// ---- src/chunk.rs:133-151 ----------------------------------------------------------------------
bool chunk_sample_terrain_texture(const Chunk &c, Vec2 world_pos, Vec2 scale, uint8_t out[4]) {
    float local_x = (world_pos.x / scale.x) - (float)c.origin[0];
    float local_y = (world_pos.y / scale.y) - (float)c.origin[1];
    if (c.has_terrain_texture) {
        const Texture &texture = c.terrain_texture;
        if (c.size == 0) return false;  // `texture.width as i32 / self.size` panics
        if ((int32_t)texture.width == INT32_MIN && c.size == -1) return false;
        int pixels_per_tile = (int)(int32_t)texture.width / c.size;
        float pixel_x = local_x * (float)pixels_per_tile;
        float pixel_y = local_y * (float)pixels_per_tile;
        // f32::clamp(0.0, w - 1.0) with w >= 1 never panics; `as u32` saturates, NaN -> 0
        uint32_t px = sat_u32(rclamp(std::floor(pixel_x), 0.0f, (float)texture.width - 1.0f));
        uint32_t py = sat_u32(rclamp(std::floor(pixel_y), 0.0f, (float)texture.height - 1.0f));
        // Texture::get_pixel, src/texture.rs:527-538
        size_t x = std::min<size_t>(px, texture.width - 1), y = std::min<size_t>(py, texture.height - 1);
        memcpy(out, &texture.data[(y * texture.width + x) * 4], 4);
        return true;
    }
    out[0] = out[1] = out[2] = out[3] = 0;
    return true;
}

// ---- src/rasterizer.rs:964-1420 --------------------------------------------------------------
static int d3_rasterize(const FrameCtx &fc, uint8_t *buffer, float *z_buffer, const int64_t *surface_id,
                        const TileRect &tile, const Batch3D &batch, const Chunk *chunk, Execution &execution) {
    const Rasterizer &r = *fc.r;
    if (!batch.has_bounding_box) return 0;
    const Rect &bbox = batch.bounding_box;
    if (!(bbox.x < (float)(tile.x + tile.width) && (bbox.x + bbox.width) > (float)tile.x &&
          bbox.y < (float)(tile.y + tile.height) && (bbox.y + bbox.height) > (float)tile.y))
        return 0;

    for (size_t triangle_index = 0; triangle_index < batch.edges.size(); ++triangle_index) {
        const Edges &edges = batch.edges[triangle_index];
        if (!edges.visible) continue;

        size_t i0 = batch.clipped_indices[triangle_index][0];
        size_t i1 = batch.clipped_indices[triangle_index][1];
        size_t i2 = batch.clipped_indices[triangle_index][2];
        const float *v0 = batch.projected_vertices[i0].data();
        const float *v1 = batch.projected_vertices[i1].data();
        const float *v2 = batch.projected_vertices[i2].data();
        const float *uv0 = batch.clipped_uvs[i0].data();
        const float *uv1 = batch.clipped_uvs[i1].data();
        const float *uv2 = batch.clipped_uvs[i2].data();

        size_t min_x, max_x, min_y, max_y;
        tri_bounds(v0, v1, v2, tile, min_x, max_x, min_y, max_y);

        for (size_t ty = min_y; ty < max_y; ++ty) {
            for (size_t tx = min_x; tx < max_x; ++tx) {
                float p[2] = {(float)tx + 0.5f, (float)ty + 0.5f};
                if (!edges_evaluate(edges, p)) continue;

                size_t idx = (ty - tile.y) * tile.width + (tx - tile.x);
                // surface_id[idx].is_some() && surface_id[idx] == batch.profile_id   (:1044-1048)
                if (surface_id[idx] >= 0 && batch.has_profile_id && surface_id[idx] == (int64_t)batch.profile_id) continue;

                float w3[3];
                barycentric_weights(v0, v1, v2, p, w3);
                float alpha = w3[0], beta = w3[1], gamma = w3[2];

                float one_over_z = 1.0f / v0[2] * alpha + 1.0f / v1[2] * beta + 1.0f / v2[2] * gamma;
                float z = 1.0f / one_over_z;

                size_t zidx = idx;
                if (!(z < z_buffer[zidx])) continue;

                float interpolated_u = (uv0[0] / v0[3]) * alpha + (uv1[0] / v1[3]) * beta + (uv2[0] / v2[3]) * gamma;
                float interpolated_v = (uv0[1] / v0[3]) * alpha + (uv1[1] / v1[3]) * beta + (uv2[1] / v2[3]) * gamma;
                float interpolated_reciprocal_w = (1.0f / v0[3]) * alpha + (1.0f / v1[3]) * beta + (1.0f / v2[3]) * gamma;
                interpolated_u /= interpolated_reciprocal_w;
                interpolated_v /= interpolated_reciprocal_w;

                Vec3 world = screen_to_world(r, p[0], p[1], z);
                Vec2 world_2d{world.x, world.z};

                Vec3 normal;
                if (!batch.normals.empty()) {  // :1083-1099
                    Vec3 n0 = batch.clipped_normals[i0], n1 = batch.clipped_normals[i1], n2 = batch.clipped_normals[i2];
                    normal = rvek::normalized(n0 * alpha + n1 * beta + n2 * gamma);
                    Vec3 view_dir = rvek::normalized(r.camera_pos - world);
                    if (rvek::dot(normal, view_dir) < 0.0f) normal = -normal;
                } else {
                    normal = Vec3{0, 0, 0};
                }

                uint8_t texel[4];
                if (!fetch_texel(fc, batch.source, batch.repeat_mode, interpolated_u, interpolated_v, true, chunk, world_2d, texel, &world))
                    return RXR_ERR_INVALID;

                float color[4];
                pixel_to_vec4(texel, color);

                // :1226-1317.  No-shader defaults first (:1305-1317); a chunk's baked shader texture (chunk.shader_textures, :1226-1267) and a
                // batch's program (:1268-1304) replace them below
                color[0] = srgb_to_linear_fast(color[0]);
                color[1] = srgb_to_linear_fast(color[1]);
                color[2] = srgb_to_linear_fast(color[2]);
                execution.color = Vec3{color[0], color[1], color[2]};
                execution.opacity.x = (float)texel[3] / 255.0f;
                execution.normal = normal;
                execution.roughness.x = 0.5f;
                execution.metallic.x = 0.0f;
                const Texture *baked = nullptr;  // chunk.shader_textures.get(shader_index), :1226-1237
                if (batch.shader >= 0 && chunk && (size_t)batch.shader < chunk->shader_textures.size() &&
                    chunk->shader_texture_present[(size_t)batch.shader])
                    baked = &chunk->shader_textures[(size_t)batch.shader];
                if (baked) {  // :1239-1267: the baked shader texture replaces the texel; the program does not run
                    texture_sample(*baked, interpolated_u, interpolated_v, r.sample_mode, batch.repeat_mode, texel);
                    pixel_to_vec4(texel, color);
                    color[0] = srgb_to_linear_fast(color[0]);
                    color[1] = srgb_to_linear_fast(color[1]);
                    color[2] = srgb_to_linear_fast(color[2]);
                    execution.color = Vec3{color[0], color[1], color[2]};
                    execution.opacity.x = color[3];
                    execution.roughness.x = 0.5f;
                    execution.metallic.x = 0.0f;
                    execution.normal = normal;
                } else if (batch.shader >= 0) {  // :1283-1304
                    const std::vector<vm::Program> &progs = chunk ? chunk->shaders : fc.scene->shaders;
                    if ((size_t)batch.shader < progs.size()) {
                        const vm::Program &program = progs[(size_t)batch.shader];
                        if (program.shade_index >= 0) {
                            execution.uv.x = interpolated_u / 4.0f;
                            execution.uv.y = interpolated_v / 4.0f;
                            execution.hitpoint = world;
                            execution.time = Vec3{r.time, r.time, r.time};
                            execution.reset(program.globals);
                            try {
                                execution.shade((size_t)program.shade_index, program, fc.assets->vm_env);
                            } catch (const vm::Fault &) {
                                return RXR_ERR_INVALID;
                            }
                        }
                    }
                }

                Vec3 mat_base = execution.color;  // :1319-1323
                normal = rvek::normalized(execution.normal);
                float mat_roughness = rclamp(execution.roughness.x, 0.0f, 1.0f);
                float mat_metallic = rclamp(execution.metallic.x, 0.0f, 1.0f);
                Vec3 mat_emissive = execution.emissive;

                Vec3 lit{0, 0, 0};

                float occlusion = chunk ? mapmini_get_occlusion(chunk->occluded_sectors, world_2d)
                                        : mapmini_get_occlusion(r.mapmini.occluded_sectors, world_2d);

                if (occlusion > 0.0f) {  // :1334-1365
                    if (r.has_ambient) {
                        float hemi = 0.5f * (normal.y + 1.0f);
                        Vec3 kd = mat_base * (1.0f - mat_metallic) * (1.0f - 0.04f);
                        Vec3 sky{r.ambient_color.x, r.ambient_color.y, r.ambient_color.z};
                        lit += sky * kd * hemi;
                    }
                    if (r.has_sun) {
                        if (r.day_factor > 0.0f) {
                            Vec3 ldir = rvek::normalized(-r.sun_dir);
                            float df = rmax(r.day_factor, 0.0f);
                            Vec3 sun_radiance{df, df, df};
                            lit += shade_fast_brdf(mat_base, mat_roughness, mat_metallic, Vec3{0, 0, 0}, normal,
                                                   rvek::normalized(r.camera_pos - world), ldir, sun_radiance);
                        }
                    }
                    lit.x *= occlusion;
                    lit.y *= occlusion;
                    lit.z *= occlusion;
                }

                float hemi = 0.5f * (normal.y + 1.0f);  // :1368-1370
                Vec3 kd = mat_base * (1.0f - mat_metallic) * (1.0f - 0.04f);
                lit += batch.ambient_color * kd * hemi;

                for (const CompiledLight *light : fc.lights) {  // :1373-1391
                    Vec3 radiance;
                    if (!light_radiance_at(*light, world, true, normal, r.hash_anim, radiance)) continue;
                    Vec3 ldir = rvek::normalized(lpos(*light) - world);
                    lit += shade_fast_brdf(mat_base, mat_roughness, mat_metallic, Vec3{0, 0, 0}, normal,
                                           rvek::normalized(r.camera_pos - world), ldir, radiance);
                }

                lit += mat_emissive;  // :1394

                color[0] = linear_to_srgb_fast(lit.x);
                color[1] = linear_to_srgb_fast(lit.y);
                color[2] = linear_to_srgb_fast(lit.z);
                color[3] = execution.opacity.x;
                vec4_to_pixel(color, texel);

                if (texel[3] == 255) {  // :1408-1412
                    memcpy(buffer + zidx * 4, texel, 4);
                    z_buffer[zidx] = z;
                }
            }
        }
    }
    return 0;
}

// ---- src/rasterizer.rs:1425-1690 -------------------------------------------------------------
static int d3_rasterize_opacity(const FrameCtx &fc, uint8_t *buffer, float *z_buffer, int64_t *surface_id,
                                const TileRect &tile, const Batch3D &batch, const Chunk *chunk, Execution &execution) {
    if (!batch.has_bounding_box) return 0;
    const Rect &bbox = batch.bounding_box;
    if (!(bbox.x < (float)(tile.x + tile.width) && (bbox.x + bbox.width) > (float)tile.x &&
          bbox.y < (float)(tile.y + tile.height) && (bbox.y + bbox.height) > (float)tile.y))
        return 0;

    for (size_t triangle_index = 0; triangle_index < batch.edges.size(); ++triangle_index) {
        const Edges &edges = batch.edges[triangle_index];
        if (!edges.visible) continue;
        size_t i0 = batch.clipped_indices[triangle_index][0];
        size_t i1 = batch.clipped_indices[triangle_index][1];
        size_t i2 = batch.clipped_indices[triangle_index][2];
        const float *v0 = batch.projected_vertices[i0].data();
        const float *v1 = batch.projected_vertices[i1].data();
        const float *v2 = batch.projected_vertices[i2].data();
        const float *uv0 = batch.clipped_uvs[i0].data();
        const float *uv1 = batch.clipped_uvs[i1].data();
        const float *uv2 = batch.clipped_uvs[i2].data();

        size_t min_x, max_x, min_y, max_y;
        tri_bounds(v0, v1, v2, tile, min_x, max_x, min_y, max_y);

        for (size_t ty = min_y; ty < max_y; ++ty) {
            for (size_t tx = min_x; tx < max_x; ++tx) {
                float p[2] = {(float)tx + 0.5f, (float)ty + 0.5f};
                if (!edges_evaluate(edges, p)) continue;
                float w3[3];
                barycentric_weights(v0, v1, v2, p, w3);
                float alpha = w3[0], beta = w3[1], gamma = w3[2];
                float one_over_z = 1.0f / v0[2] * alpha + 1.0f / v1[2] * beta + 1.0f / v2[2] * gamma;
                float z = 1.0f / one_over_z;
                size_t zidx = (ty - tile.y) * tile.width + (tx - tile.x);
                if (!(z < z_buffer[zidx])) continue;

                float interpolated_u = (uv0[0] / v0[3]) * alpha + (uv1[0] / v1[3]) * beta + (uv2[0] / v2[3]) * gamma;
                float interpolated_v = (uv0[1] / v0[3]) * alpha + (uv1[1] / v1[3]) * beta + (uv2[1] / v2[3]) * gamma;
                float interpolated_reciprocal_w = (1.0f / v0[3]) * alpha + (1.0f / v1[3]) * beta + (1.0f / v2[3]) * gamma;
                interpolated_u /= interpolated_reciprocal_w;
                interpolated_v /= interpolated_reciprocal_w;

                Vec3 world = screen_to_world(*fc.r, p[0], p[1], z);  // :1515
                Vec2 world_2d{world.x, world.z};

                uint8_t texel[4];
                if (!fetch_texel(fc, batch.source, batch.repeat_mode, interpolated_u, interpolated_v, true, chunk, world_2d, texel, &world))
                    return RXR_ERR_INVALID;

                float color[4];
                pixel_to_vec4(texel, color);
                color[0] = srgb_to_linear_fast(color[0]);
                color[1] = srgb_to_linear_fast(color[1]);
                color[2] = srgb_to_linear_fast(color[2]);
                execution.color = Vec3{color[0], color[1], color[2]};
                execution.opacity.x = (float)texel[3] / 255.0f;

                if (batch.shader >= 0) {  // :1642-1667
                    const std::vector<vm::Program> &progs = chunk ? chunk->shaders : fc.scene->shaders;
                    if ((size_t)batch.shader < progs.size()) {
                        const vm::Program &program = progs[(size_t)batch.shader];
                        if (program.shade_index >= 0) {
                            execution.normal = Vec3{0, 0, 0};
                            execution.uv.x = interpolated_u / 4.0f;
                            execution.uv.y = interpolated_v / 4.0f;
                            execution.hitpoint = world;
                            execution.time = Vec3{fc.r->time, fc.r->time, fc.r->time};
                            execution.roughness.x = 0.5f;
                            execution.metallic.x = 0.0f;
                            execution.reset(program.globals);
                            try {
                                execution.shade((size_t)program.shade_index, program, fc.assets->vm_env);
                            } catch (const vm::Fault &) {
                                return RXR_ERR_INVALID;
                            }
                        }
                    }
                }

                color[0] = linear_to_srgb_fast(execution.color.x);  // :1670-1674
                color[1] = linear_to_srgb_fast(execution.color.y);
                color[2] = linear_to_srgb_fast(execution.color.z);
                color[3] = execution.opacity.x;
                vec4_to_pixel(color, texel);

                memcpy(buffer + zidx * 4, texel, 4);  // :1678-1682
                z_buffer[zidx] = z;
                surface_id[zidx] = batch.has_profile_id ? (int64_t)batch.profile_id : -1;
            }
        }
    }
    return 0;
}

// ---- src/rasterizer.rs:1777-1821 -------------------------------------------------------------
static void rasterize_line_bresenham(const float p0[2], const float p1[2], uint8_t *buffer, const TileRect &tile,
                                     const uint8_t color[4]) {
    int64_t x0 = sat_isize(p0[0]), y0 = sat_isize(p0[1]);
    int64_t x1 = sat_isize(p1[0]), y1 = sat_isize(p1[1]);
    int64_t dx = std::llabs(x1 - x0), dy = std::llabs(y1 - y0);
    int64_t sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
    int64_t err = dx - dy;
    int64_t x = x0, y = y0;
    while (x != x1 || y != y1) {
        uint64_t tx = (uint64_t)(x - (int64_t)tile.x);
        uint64_t ty = (uint64_t)(y - (int64_t)tile.y);
        if (tx < tile.width && ty < tile.height) {
            size_t idx = (ty * tile.width + tx) * 4;
            memcpy(buffer + idx, color, 4);
        }
        int64_t e2 = err * 2;
        if (e2 > -dy) {
            err -= dy;
            x += sx;
        }
        if (e2 < dx) {
            err += dx;
            y += sy;
        }
    }
}

// ---- src/rasterizer.rs:584-959 ---------------------------------------------------------------
static int d2_rasterize(const FrameCtx &fc, uint8_t *buffer, const TileRect &tile, const Batch2D &batch,
                        const Chunk *chunk, Execution &execution) {
    const Rasterizer &r = *fc.r;
    if (!batch.has_bounding_box) return 0;
    const Rect &bbox = batch.bounding_box;
    const float pad = 0.5f;
    if (!(bbox.x < (float)(tile.x + tile.width) + pad && (bbox.x + bbox.width) > (float)tile.x - pad &&
          bbox.y < (float)(tile.y + tile.height) + pad && (bbox.y + bbox.height) > (float)tile.y - pad))
        return 0;

    const uint8_t WHITE[4] = {255, 255, 255, 255};
    const uint8_t *line_color = (batch.source.kind == RXR_SOURCE_PIXEL) ? batch.source.pixel : WHITE;

    switch (batch.mode) {
        case RXR_MODE_TRIANGLES: {
            for (size_t triangle_index = 0; triangle_index < batch.edges.size(); ++triangle_index) {
                const Edges &edges = batch.edges[triangle_index];
                size_t i0 = batch.indices[triangle_index][0];
                size_t i1 = batch.indices[triangle_index][1];
                size_t i2 = batch.indices[triangle_index][2];
                const float *v0 = batch.projected_vertices[i0].data();
                const float *v1 = batch.projected_vertices[i1].data();
                const float *v2 = batch.projected_vertices[i2].data();
                const float *uv0 = batch.uvs[i0].data();
                const float *uv1 = batch.uvs[i1].data();
                const float *uv2 = batch.uvs[i2].data();

                size_t min_x, max_x, min_y, max_y;
                tri_bounds(v0, v1, v2, tile, min_x, max_x, min_y, max_y);

                for (size_t ty = min_y; ty < max_y; ++ty) {
                    for (size_t tx = min_x; tx < max_x; ++tx) {
                        float p[2] = {(float)tx + 0.5f, (float)ty + 0.5f};
                        // :641-652 -- unreachable because tx,ty are clamped to the tile, kept for fidelity
                        if (p[0] >= (float)(tile.x + tile.width)) p[0] -= (float)tile.width;
                        else if (p[0] < (float)tile.x) p[0] += (float)tile.width;
                        if (p[1] >= (float)(tile.y + tile.height)) p[1] -= (float)tile.height;
                        else if (p[1] < (float)tile.y) p[1] += (float)tile.height;

                        if (!(edges.visible && edges_evaluate(edges, p))) continue;

                        float w[3];
                        barycentric_weights(v0, v1, v2, p, w);
                        float u = uv0[0] * w[0] + uv1[0] * w[1] + uv2[0] * w[2];
                        float v = uv0[1] * w[0] + uv1[1] * w[1] + uv2[1] * w[2];

                        // :664-670
                        Vec2 grid_space_pos = (Vec2{(float)tx, (float)ty} - Vec2{r.width, r.height} / 2.0f) -
                                              Vec2{r.translationd2.x - r.width / 2.0f, r.translationd2.y - r.height / 2.0f};
                        Vec2 world = grid_space_pos / r.scaled2;

                        uint8_t texel[4];
                        if (!fetch_texel(fc, batch.source, batch.repeat_mode, u, v, false, chunk, world, texel))
                            return RXR_ERR_INVALID;

                        if (batch.shader >= 0) {  // :760-797
                            const std::vector<vm::Program> &progs = chunk ? chunk->shaders : fc.scene->shaders;
                            if ((size_t)batch.shader < progs.size()) {
                                const vm::Program &program = progs[(size_t)batch.shader];
                                if (program.shade_index >= 0) {
                                    float color[4];
                                    pixel_to_vec4(texel, color);
                                    execution.uv.x = u / 4.0f;
                                    execution.uv.y = v / 4.0f;
                                    execution.color = Vec3{color[0], color[1], color[2]};
                                    execution.hitpoint.x = world.x;
                                    execution.hitpoint.y = world.y;
                                    execution.time = Vec3{r.time, r.time, r.time};
                                    execution.roughness.x = 0.5f;
                                    execution.metallic.x = 0.0f;
                                    execution.reset(program.globals);
                                    try {
                                        execution.shade((size_t)program.shade_index, program, fc.assets->vm_env);
                                    } catch (const vm::Fault &) {
                                        return RXR_ERR_INVALID;
                                    }
                                    color[0] = execution.color.x;
                                    color[1] = execution.color.y;
                                    color[2] = execution.color.z;
                                    color[3] = 1.0f;
                                    vec4_to_pixel(color, texel);
                                }
                            }
                        }

                        // :799-803 (operator precedence: `a && b || c`)
                        if ((batch.receives_light && fc.any_lights) || r.has_ambient) {
                            float accumulated_light[3] = {0.0f, 0.0f, 0.0f};
                            if (r.has_ambient) {
                                float occlusion = chunk ? mapmini_get_occlusion(chunk->occluded_sectors, world)
                                                        : mapmini_get_occlusion(r.mapmini.occluded_sectors, world);
                                accumulated_light[0] += r.ambient_color.x * occlusion;
                                accumulated_light[1] += r.ambient_color.y * occlusion;
                                accumulated_light[2] += r.ambient_color.z * occlusion;
                            }
                            for (const CompiledLight *light : fc.lights) {
                                float light_color[3];
                                if (!light_color_at(*light, Vec3{world.x, 0.0f, world.y}, r.hash_anim, true, light_color)) continue;
                                bool light_is_visible = true;
                                if (light->light_type == RXR_LIGHT_AMBIENT_DAYLIGHT) {
                                    float occlusion = chunk ? mapmini_get_occlusion(chunk->occluded_sectors, world)
                                                            : mapmini_get_occlusion(r.mapmini.occluded_sectors, world);
                                    light_color[0] *= occlusion;
                                    light_color[1] *= occlusion;
                                    light_color[2] *= occlusion;
                                }
                                if (light->light_type != RXR_LIGHT_AMBIENT && light->light_type != RXR_LIGHT_AMBIENT_DAYLIGHT &&
                                    !mapmini_is_visible(r.mapmini, world, Vec2{light->position[0], light->position[2]}))
                                    light_is_visible = false;
                                if (light_is_visible) {
                                    accumulated_light[0] += light_color[0];
                                    accumulated_light[1] += light_color[1];
                                    accumulated_light[2] += light_color[2];
                                }
                            }
                            for (int i = 0; i < 3; ++i) accumulated_light[i] = rclamp(accumulated_light[i], 0.0f, 1.0f);
                            for (int i = 0; i < 3; ++i)
                                texel[i] = sat_u8(rclamp(((float)texel[i] / 255.0f) * accumulated_light[i] * 255.0f, 0.0f, 255.0f));
                        }

                        size_t idx = ((ty - tile.y) * tile.width + (tx - tile.x)) * 4;  // :876-895
                        if (texel[3] == 255) {
                            memcpy(buffer + idx, texel, 4);
                        } else {
                            float src_alpha = (float)texel[3] / 255.0f;
                            float dst_alpha = 1.0f - src_alpha;
                            for (int i = 0; i < 3; ++i)
                                buffer[idx + i] = sat_u8(((float)texel[i] * src_alpha) + ((float)buffer[idx + i] * dst_alpha));
                            if (!r.preserve_transparency)
                                buffer[idx + 3] = 255;
                            else
                                buffer[idx + 3] = buffer[idx + 3] > texel[3] ? buffer[idx + 3] : texel[3];
                        }
                    }
                }
            }
            break;
        }
        case RXR_MODE_LINES:
            for (auto &t : batch.indices) {
                const float *p0 = batch.projected_vertices[t[0]].data();
                const float *p1 = batch.projected_vertices[t[1]].data();
                rasterize_line_bresenham(p0, p1, buffer, tile, line_color);
            }
            break;
        case RXR_MODE_LINE_STRIP:
            // `0..(len - 1)` underflows (panics) on an empty batch in the reference; treated as empty here
            for (size_t i = 0; i + 1 < batch.projected_vertices.size(); ++i)
                rasterize_line_bresenham(batch.projected_vertices[i].data(), batch.projected_vertices[i + 1].data(), buffer,
                                         tile, line_color);
            break;
        case RXR_MODE_LINE_LOOP:
            for (size_t i = 0; i < batch.projected_vertices.size(); ++i)
                rasterize_line_bresenham(batch.projected_vertices[i].data(),
                                         batch.projected_vertices[(i + 1) % batch.projected_vertices.size()].data(), buffer,
                                         tile, line_color);
            break;
    }
    return 0;
}

// ---- one tile of src/rasterizer.rs:275-556 ---------------------------------------------------
static int raster_tile(const FrameCtx &fc, const TileRect &tile, std::vector<uint8_t> &buffer) {
    const Rasterizer &r = *fc.r;
    const Scene &scene = *fc.scene;
    size_t n = tile.width * tile.height;
    buffer.assign(n * 4, 0);  // :277-282
    if (r.has_background_color)
        for (size_t i = 0; i < n; ++i) memcpy(&buffer[i * 4], r.background_color, 4);
    std::vector<uint8_t> buffer_opacity(n * 4, 0);  // :285
    std::vector<float> z_buffer(n, 1.0f);           // :287-288
    std::vector<float> z_buffer_opacity(n, 1.0f);
    std::vector<int64_t> surface_id(n, -1);         // Vec<Option<u32>>, -1 == None  (:290)

    if (!r.ignore_background_shader && scene.background == RXR_BG_VGRADIENT) {  // :292-308
        float screen_x = (float)fc.r->width, screen_y = (float)fc.r->height;
        for (size_t ty = 0; ty < tile.height; ++ty) {
            for (size_t tx = 0; tx < tile.width; ++tx) {
                float uvx = (float)(tile.x + tx) / screen_x;
                float uvy = (float)(tile.y + ty) / screen_y;
                (void)uvx;
                uint8_t intensity = sat_u8(rclamp(uvy * 128.0f, 0.0f, 128.0f));  // src/shader/vgradient.rs:11-15
                size_t idx = (ty * tile.width + tx) * 4;
                buffer[idx] = intensity;
                buffer[idx + 1] = intensity;
                buffer[idx + 2] = intensity;
                buffer[idx + 3] = 255;
            }
        }
    }

    if (!r.ignore_background_shader && scene.background == RXR_BG_GRID) {  // :292-308 with GridShader (shader/grid.rs:36-108)
        const float screen_x = (float)fc.r->width, screen_y = (float)fc.r->height;
        const float grid_size = scene.background_grid[0], sub_grid_div = scene.background_grid[1];
        const float off_x = scene.background_grid[2], off_y = scene.background_grid[3];
        auto closest_mul = [](float delta, float value) { return delta * roundf(value / delta); };
        auto mul_dist = [&](float delta, float value) { return fabsf(value - closest_mul(delta, value)); };
        const float bg_color[4] = {0.05f, 0.05f, 0.05f, 1.0f}, line_color[4] = {0.15f, 0.15f, 0.15f, 1.0f}, sub_line_color[4] = {0.11f, 0.11f, 0.11f, 1.0f};
        for (size_t ty = 0; ty < tile.height; ++ty) {
            for (size_t tx = 0; tx < tile.width; ++tx) {
                const float uvx = (float)(tile.x + tx) / screen_x, uvy = (float)(tile.y + ty) / screen_y;
                const float position_x = uvx * screen_x, position_y = uvy * screen_y;
                const float origin_x = screen_x / 2.0f + off_x, origin_y = screen_y / 2.0f + off_y;
                const float th = 1.0f, sth = 1.0f;
                // align_pixel(origin, 1): 1 is odd
                const float aligned_x = roundf(origin_x - 0.5f) + 0.5f, aligned_y = roundf(origin_y - 0.5f) + 0.5f;
                const float rel_x = position_x - aligned_x, rel_y = position_y - aligned_y;
                const float dist_x = mul_dist(grid_size, rel_x), dist_y = mul_dist(grid_size, rel_y);
                const float *c = bg_color;
                if (fminf(dist_x, dist_y) <= th * 0.5f) {
                    c = line_color;
                } else {
                    const float dtf_x = fabsf(rel_x - grid_size * floorf(rel_x / grid_size)), dtf_y = fabsf(rel_y - grid_size * floorf(rel_y / grid_size));
                    const float sub_size = grid_size / roundf(sub_grid_div);
                    float sub_dist_x = mul_dist(sub_size, dtf_x), sub_dist_y = mul_dist(sub_size, dtf_y);
                    const float rc_x = roundf(dist_x / sub_size), rc_y = roundf(dist_y / sub_size);
                    const float extra = grid_size - sub_size * sub_grid_div;
                    if (rc_x == sub_grid_div) sub_dist_x = sub_dist_x + extra;
                    if (rc_y == sub_grid_div) sub_dist_y = sub_dist_y + extra;
                    if (fminf(sub_dist_x, sub_dist_y) <= sth * 0.5f) c = sub_line_color;
                }
                const size_t idx = (ty * tile.width + tx) * 4;
                for (int k = 0; k < 4; ++k) buffer[idx + k] = f32_to_u8_saturated(c[k]);  // vec4_to_pixel, lib.rs:64-79
            }
        }
    }

    Execution execution;  // :310
    int rc = 0;

    if (r.d3_active) {  // :312-499
        for (const Chunk &chunk : scene.chunks) {
            for (const Batch3D &b : chunk.batches3d_opacity)
                if ((rc = d3_rasterize_opacity(fc, buffer_opacity.data(), z_buffer_opacity.data(), surface_id.data(), tile, b, &chunk, execution))) return rc;
            for (const Batch3D &b : chunk.batches3d)
                if ((rc = d3_rasterize(fc, buffer.data(), z_buffer.data(), surface_id.data(), tile, b, &chunk, execution))) return rc;
            for (const Batch3D &b : chunk.terrain_batch3d)  // :343-356
                if ((rc = d3_rasterize(fc, buffer.data(), z_buffer.data(), surface_id.data(), tile, b, &chunk, execution))) return rc;
        }
        for (const Batch3D &b : scene.d3_static)
            if ((rc = d3_rasterize(fc, buffer.data(), z_buffer.data(), surface_id.data(), tile, b, nullptr, execution))) return rc;
        for (const Batch3D &b : scene.d3_dynamic)
            if ((rc = d3_rasterize(fc, buffer.data(), z_buffer.data(), surface_id.data(), tile, b, nullptr, execution))) return rc;
        for (const Batch3D &b : scene.d3_overlay)
            if ((rc = d3_rasterize(fc, buffer.data(), z_buffer.data(), surface_id.data(), tile, b, nullptr, execution))) return rc;

        for (size_t ty = 0; ty < tile.height; ++ty) {  // :409-497
            for (size_t tx = 0; tx < tile.width; ++tx) {
                size_t idx = (ty * tile.width + tx) * 4;
                size_t z_idx = ty * tile.width + tx;
                if (z_buffer[z_idx] == 1.0f) {  // :420-461 (render-graph miss nodes: none)
                    float color[4] = {0.0f, 0.0f, 0.0f, 1.0f};
                    if (r.has_brush_preview) {  // :435-458; the ray starts at the pixel's corner, not its centre (:422-423)
                        Vec3 origin, dir;
                        screen_ray(r, (float)(tile.x + tx), (float)(tile.y + ty), origin, dir);
                        if (std::fabs(dir.y) > 1e-5f) {
                            float t = -origin.y / dir.y;
                            if (t > 0.0f) {
                                Vec3 world = origin + dir * t;
                                float dist = rvek::magnitude(world - r.brush_position);
                                if (dist < r.brush_radius) {
                                    float normalized = dist / r.brush_radius;
                                    float falloff = rclamp(r.brush_falloff, 0.001f, 1.0f);
                                    float fade = rclamp((1.0f - normalized) / falloff, 0.0f, 1.0f);
                                    float blend = 0.2f + 0.6f * fade;
                                    for (int ch = 0; ch < 3; ++ch) color[ch] = rmin(color[ch] * (1.0f - blend) + blend, 1.0f);
                                }
                            }
                        }
                    }
                    vec4_to_pixel(color, &buffer[idx]);
                }
                if (z_buffer_opacity[z_idx] < 1.0f && z_buffer[z_idx] > z_buffer_opacity[z_idx]) {  // :464-495
                    float src_r = (float)buffer_opacity[idx];
                    float src_g = (float)buffer_opacity[idx + 1];
                    float src_b = (float)buffer_opacity[idx + 2];
                    float src_a = (float)buffer_opacity[idx + 3] / 255.0f;
                    float dst_r = (float)buffer[idx];
                    float dst_g = (float)buffer[idx + 1];
                    float dst_b = (float)buffer[idx + 2];
                    float dst_a = (float)buffer[idx + 3] / 255.0f;
                    float inv_a = 1.0f - src_a;
                    float out_r = src_r * src_a + dst_r * inv_a;
                    float out_g = src_g * src_a + dst_g * inv_a;
                    float out_b = src_b * src_a + dst_b * inv_a;
                    float out_a = !r.preserve_transparency ? 1.0f : rclamp(src_a + dst_a * inv_a, 0.0f, 1.0f);
                    buffer[idx] = sat_u8(rclamp(out_r, 0.0f, 255.0f));
                    buffer[idx + 1] = sat_u8(rclamp(out_g, 0.0f, 255.0f));
                    buffer[idx + 2] = sat_u8(rclamp(out_b, 0.0f, 255.0f));
                    buffer[idx + 3] = sat_u8(rclamp(out_a * 255.0f, 0.0f, 255.0f));
                }
            }
        }
    }

    if (r.d2_active) {  // :501-553
        for (const Chunk &chunk : scene.chunks) {
            for (const Batch2D &b : chunk.batches2d)
                if ((rc = d2_rasterize(fc, buffer.data(), tile, b, &chunk, execution))) return rc;
            for (const Batch2D &b : chunk.terrain_batch2d)  // :515-525
                if ((rc = d2_rasterize(fc, buffer.data(), tile, b, &chunk, execution))) return rc;
        }
        for (const Batch2D &b : scene.d2_static)
            if ((rc = d2_rasterize(fc, buffer.data(), tile, b, nullptr, execution))) return rc;
        for (const Batch2D &b : scene.d2_dynamic)
            if ((rc = d2_rasterize(fc, buffer.data(), tile, b, nullptr, execution))) return rc;
    }
    return 0;
}

// ---- src/rasterizer.rs:185-580 ---------------------------------------------------------------
int rasterize(Rasterizer &r, Scene &scene, uint8_t *pixels, size_t width, size_t height, size_t tile_size,
              const Assets &assets, int n_threads) {
    if (tile_size == 0) return RXR_ERR_INVALID;  // step_by(0) panics
    r.width = (float)width;
    r.height = (float)height;
    r.hash_anim = hash_u32((uint32_t)scene.animation_frame);  // :208

    if (!scene_project(scene, r.has_m2d ? &r.projection_matrix_2d : nullptr, r.view_matrix, r.projection_matrix, r.width,
                       r.height))  // :210-216
        return RXR_ERR_INVALID;

    for (const Chunk &c : scene.chunks)  // :219-223 (never cleared in the reference either)
        for (const CompiledLight &l : c.lights) scene.dynamic_lights.push_back(l);

    std::vector<TileRect> tiles;  // :256-268
    for (size_t y = 0; y < height; y += tile_size)
        for (size_t x = 0; x < width; x += tile_size)
            tiles.push_back(TileRect{x, y, std::min(tile_size, width - x), std::min(tile_size, height - y)});

    FrameCtx fc;
    fc.r = &r;
    fc.scene = &scene;
    fc.assets = &assets;
    for (auto &l : scene.lights) fc.lights.push_back(&l);
    for (auto &l : scene.dynamic_lights) fc.lights.push_back(&l);
    fc.any_lights = !scene.lights.empty() || !scene.dynamic_lights.empty();

    std::vector<std::vector<uint8_t>> tile_buffers(tiles.size());  // :273-557
    std::atomic<size_t> next{0};
    std::atomic<int> status{0};
    auto worker = [&]() {
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= tiles.size()) break;
            int rc = raster_tile(fc, tiles[i], tile_buffers[i]);
            if (rc) status.store(rc);
        }
    };
    if (n_threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
    }
    if (status.load()) return status.load();

    for (size_t i = 0; i < tiles.size(); ++i) {  // :559-579
        const TileRect &tile = tiles[i];
        const std::vector<uint8_t> &tile_buffer = tile_buffers[i];
        size_t tile_row_bytes = tile.width * 4;
        size_t framebuffer_row_bytes = width * 4;
        size_t src_offset = 0;
        size_t dst_offset = (tile.y * width + tile.x) * 4;
        for (size_t row = 0; row < tile.height; ++row) {
            memcpy(pixels + dst_offset, tile_buffer.data() + src_offset, tile_row_bytes);
            src_offset += tile_row_bytes;
            dst_offset += framebuffer_row_bytes;
        }
    }
    return 0;
}

}  // namespace orc
