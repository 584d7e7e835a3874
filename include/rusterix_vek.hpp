// rusterix_vek.hpp -- host-side restatement of the vek 0.17.2 operations the rasterizer path uses.
//
// vek is a third-party crate (reference Cargo.lock:3826-3827, Cargo.toml:33) and is NOT vendored in
// /root/reference, so its arithmetic cannot be read from source here.  What is restated below is the
// published behaviour of vek's column-major Mat4/Mat3/Vec types as recalled in SURVEY.md section 8c:
//   - Mat * Vec: first column times v.x, then the other columns accumulated with f32::mul_add (a true
//     fused multiply-add).  RXR_VEK_FUSED_MATVEC=0 switches to unfused mul+add (parity is unpinned
//     here; the oracle and the device code flip together).
//   - dot = left-to-right sum of products; magnitude = sqrt(dot); normalized = v / magnitude
//     (three divisions); lerp(a,b,t) = mul_add(clamp01(t), b-a, a).
// Call sites in the reference: src/rasterizer.rs:97,116 (inverted), :1719,1723 (Mat4*Vec4),
// src/batch/batch3d.rs:490,527,555,558,692 (Mat4*Mat4, Mat4*Vec4), src/batch/batch2d.rs:385
// (Mat3*Vec3), src/camera/d3orbit.rs:50-56, src/camera/d3firstp.rs:36-42 (look_at_rh,
// perspective_fov_rh_zo).
//
// Compile every translation unit that includes this with -ffp-contract=off: Rust never contracts.
#pragma once
#include <cmath>
#include <cstdint>

#ifndef RXR_VEK_FUSED_MATVEC
#define RXR_VEK_FUSED_MATVEC 1
#endif

namespace rvek {

struct Vec2 { float x = 0, y = 0; };
struct Vec3 { float x = 0, y = 0, z = 0; };
struct Vec4 { float x = 0, y = 0, z = 0, w = 0; };

inline Vec2 operator+(Vec2 a, Vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline Vec2 operator-(Vec2 a, Vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline Vec2 operator*(Vec2 a, float s) { return {a.x * s, a.y * s}; }
inline Vec2 operator/(Vec2 a, float s) { return {a.x / s, a.y / s}; }

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline Vec3 &operator+=(Vec3 &a, Vec3 b) { a = a + b; return a; }

inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
inline float magnitude(Vec3 a) { return std::sqrt(dot(a, a)); }
inline float magnitude(Vec2 a) { return std::sqrt(dot(a, a)); }
inline Vec3 normalized(Vec3 a) { return a / magnitude(a); }
inline Vec2 normalized(Vec2 a) { return a / magnitude(a); }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Rust f32::clamp: NaN stays NaN (Appendix B of SURVEY.md)
inline float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

inline float lerp(float a, float b, float t) { return std::fmaf(rclamp(t, 0.0f, 1.0f), b - a, a); }
inline Vec3 lerp(Vec3 a, Vec3 b, float t) { return {lerp(a.x, b.x, t), lerp(a.y, b.y, t), lerp(a.z, b.z, t)}; }

// Column-major 4x4: m[c*4 + r] == vek cols[c][r]
struct Mat4 {
    float m[16];
    float &at(int r, int c) { return m[c * 4 + r]; }
    float at(int r, int c) const { return m[c * 4 + r]; }
    static Mat4 identity() {
        Mat4 o{};
        for (int i = 0; i < 16; ++i) o.m[i] = 0.0f;
        o.m[0] = o.m[5] = o.m[10] = o.m[15] = 1.0f;
        return o;
    }
    // vek Mat4::new takes row-major arguments
    static Mat4 from_rows(float m00, float m01, float m02, float m03, float m10, float m11, float m12, float m13,
                          float m20, float m21, float m22, float m23, float m30, float m31, float m32, float m33) {
        Mat4 o{};
        o.at(0, 0) = m00; o.at(0, 1) = m01; o.at(0, 2) = m02; o.at(0, 3) = m03;
        o.at(1, 0) = m10; o.at(1, 1) = m11; o.at(1, 2) = m12; o.at(1, 3) = m13;
        o.at(2, 0) = m20; o.at(2, 1) = m21; o.at(2, 2) = m22; o.at(2, 3) = m23;
        o.at(3, 0) = m30; o.at(3, 1) = m31; o.at(3, 2) = m32; o.at(3, 3) = m33;
        return o;
    }
    static Mat4 scaling_3d(Vec3 s) {
        Mat4 o = identity();
        o.at(0, 0) = s.x; o.at(1, 1) = s.y; o.at(2, 2) = s.z;
        return o;
    }
    static Mat4 translation_3d(Vec3 t) {
        Mat4 o = identity();
        o.at(0, 3) = t.x; o.at(1, 3) = t.y; o.at(2, 3) = t.z;
        return o;
    }
};

inline float madd(float a, float b, float c) {
#if RXR_VEK_FUSED_MATVEC
    return std::fmaf(a, b, c);
#else
    return a * b + c;
#endif
}

inline Vec4 operator*(const Mat4 &a, Vec4 v) {
    float o[4];
    for (int r = 0; r < 4; ++r) {
        float acc = a.m[0 * 4 + r] * v.x;
        acc = madd(a.m[1 * 4 + r], v.y, acc);
        acc = madd(a.m[2 * 4 + r], v.z, acc);
        acc = madd(a.m[3 * 4 + r], v.w, acc);
        o[r] = acc;
    }
    return {o[0], o[1], o[2], o[3]};
}

inline Mat4 operator*(const Mat4 &a, const Mat4 &b) {
    Mat4 o{};
    for (int c = 0; c < 4; ++c) {
        Vec4 col = a * Vec4{b.m[c * 4 + 0], b.m[c * 4 + 1], b.m[c * 4 + 2], b.m[c * 4 + 3]};
        o.m[c * 4 + 0] = col.x; o.m[c * 4 + 1] = col.y; o.m[c * 4 + 2] = col.z; o.m[c * 4 + 3] = col.w;
    }
    return o;
}

inline Vec4 operator/(Vec4 a, float s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }

// Column-major 3x3: m[c*3 + r]
struct Mat3 {
    float m[9];
    float at(int r, int c) const { return m[c * 3 + r]; }
    float &at(int r, int c) { return m[c * 3 + r]; }
    static Mat3 identity() {
        Mat3 o{};
        for (int i = 0; i < 9; ++i) o.m[i] = 0.0f;
        o.m[0] = o.m[4] = o.m[8] = 1.0f;
        return o;
    }
};

inline Vec3 operator*(const Mat3 &a, Vec3 v) {
    float o[3];
    for (int r = 0; r < 3; ++r) {
        float acc = a.m[0 * 3 + r] * v.x;
        acc = madd(a.m[1 * 3 + r], v.y, acc);
        acc = madd(a.m[2 * 3 + r], v.z, acc);
        o[r] = acc;
    }
    return {o[0], o[1], o[2]};
}

// General 4x4 inverse by cofactors (the classic 16-cofactor expansion; vek's `inverted`).
// Host-only: the real shim calls vek itself, so only self-consistency matters here.
inline Mat4 inverted(const Mat4 &a) {
    const float *m = a.m;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    det = 1.0f / det;
    Mat4 o{};
    for (int i = 0; i < 16; ++i) o.m[i] = inv[i] * det;
    return o;
}

// Mat4::look_at_rh(eye, target, up): right-handed view matrix (OpenGL convention)
inline Mat4 look_at_rh(Vec3 eye, Vec3 target, Vec3 up) {
    Vec3 f = normalized(target - eye);
    Vec3 s = normalized(cross(f, up));
    Vec3 u = cross(s, f);
    return Mat4::from_rows(s.x, s.y, s.z, -dot(s, eye),
                           u.x, u.y, u.z, -dot(u, eye),
                           -f.x, -f.y, -f.z, dot(f, eye),
                           0.0f, 0.0f, 0.0f, 1.0f);
}

// Mat4::perspective_fov_rh_zo(fov_y_radians, width, height, near, far): right-handed, depth 0..1
inline Mat4 perspective_fov_rh_zo(float fov_y_radians, float width, float height, float near, float far) {
    float rad = fov_y_radians;
    float h = std::cos(rad / 2.0f) / std::sin(rad / 2.0f);
    float w = h * height / width;
    float m22 = far / (near - far);
    float m23 = -(far * near) / (far - near);
    return Mat4::from_rows(w, 0, 0, 0,
                           0, h, 0, 0,
                           0, 0, m22, m23,
                           0, 0, -1.0f, 0);
}

}  // namespace rvek
