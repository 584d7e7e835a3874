// Writes the tables tests/golden/vek_probe/expected.txt is made of: for the same 64 seeded inputs as src/main.rs, the results
// of every candidate rounding of the six vek operations, plus (variant "header") what include/rusterix_vek.hpp computes in
// the mode this repository is built in.  Build: g++ -O2 -std=c++17 -ffp-contract=off -I../../../include gen_expected.cpp
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "rusterix_vek.hpp"

static uint32_t S = 0x52585231u;
static float lcg() {
    S = S * 1664525u + 1013904223u;
    return (float)(S >> 8) / 16777216.0f * 8.0f - 4.0f;
}
static void hex(const char *op, const char *variant, int i, const float *v, int n) {
    printf("%d %s %s", i, op, variant);
    for (int k = 0; k < n; ++k) {
        uint32_t b;
        memcpy(&b, &v[k], 4);
        printf(" %08x", b);
    }
    printf("\n");
}
template <bool FUSED>
static void matvec(const float *m, const float *v, float *o) {  // column-major m[c*4+r]
    for (int r = 0; r < 4; ++r) {
        float acc = m[r] * v[0];
        for (int c = 1; c < 4; ++c) acc = FUSED ? std::fmaf(m[c * 4 + r], v[c], acc) : m[c * 4 + r] * v[c] + acc;
        o[r] = acc;
    }
}
int main() {
    for (int i = 0; i < 64; ++i) {
        rvek::Mat4 m, n;
        for (float &x : m.m) x = lcg();
        for (float &x : n.m) x = lcg();
        float v[4] = {lcg(), lcg(), lcg(), lcg()};
        float a[3] = {lcg(), lcg(), lcg()}, b[3] = {lcg(), lcg(), lcg()};
        const float t = lcg() / 8.0f + 0.5f;
        float o[16];
        // Mat4 * Vec4: accumulate the columns with f32::mul_add ("fused") or with mul + add ("unfused")
        matvec<true>(m.m, v, o); hex("matvec", "fused", i, o, 4);
        matvec<false>(m.m, v, o); hex("matvec", "unfused", i, o, 4);
        { rvek::Vec4 h = m * rvek::Vec4{v[0], v[1], v[2], v[3]}; float hh[4] = {h.x, h.y, h.z, h.w}; hex("matvec", "header", i, hh, 4); }
        // Mat4 * Mat4: column c of the product = M * (column c of N)
        for (int c = 0; c < 4; ++c) matvec<true>(m.m, n.m + 4 * c, o + 4 * c);
        hex("matmat", "fused", i, o, 16);
        for (int c = 0; c < 4; ++c) matvec<false>(m.m, n.m + 4 * c, o + 4 * c);
        hex("matmat", "unfused", i, o, 16);
        { rvek::Mat4 h = m * n; hex("matmat", "header", i, h.m, 16); }
        // Vec3::normalized: v / magnitude (three divisions) or v * (1 / magnitude)
        const float mag = std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
        { float d[3] = {a[0] / mag, a[1] / mag, a[2] / mag}; hex("normalized", "div", i, d, 3); }
        { const float r = 1.0f / mag; float d[3] = {a[0] * r, a[1] * r, a[2] * r}; hex("normalized", "rcp", i, d, 3); }
        { rvek::Vec3 h = rvek::normalized(rvek::Vec3{a[0], a[1], a[2]}); float hh[3] = {h.x, h.y, h.z}; hex("normalized", "header", i, hh, 3); }
        // Vec3::lerp(from, to, factor), factor clamped to [0, 1]: mul_add(t, to - from, from) | from + (to - from) * t | from * (1 - t) + to * t
        const float tc = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        { float d[3]; for (int k = 0; k < 3; ++k) d[k] = std::fmaf(tc, b[k] - a[k], a[k]); hex("lerp", "mul_add", i, d, 3); }
        { float d[3]; for (int k = 0; k < 3; ++k) d[k] = a[k] + (b[k] - a[k]) * tc; hex("lerp", "unfused", i, d, 3); }
        { float d[3]; for (int k = 0; k < 3; ++k) d[k] = a[k] * (1.0f - tc) + b[k] * tc; hex("lerp", "precise", i, d, 3); }
        { rvek::Vec3 h = rvek::lerp(rvek::Vec3{a[0], a[1], a[2]}, rvek::Vec3{b[0], b[1], b[2]}, t); float hh[3] = {h.x, h.y, h.z}; hex("lerp", "header", i, hh, 3); }
        // Vec3::dot: products summed left to right | right to left | accumulated with mul_add from either end
        { float d = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; hex("dot", "sum", i, &d, 1); }
        { float d = a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); hex("dot", "sum_right", i, &d, 1); }
        { float d = std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])); hex("dot", "mul_add", i, &d, 1); }
        { float d = std::fmaf(a[0], b[0], std::fmaf(a[1], b[1], a[2] * b[2])); hex("dot", "mul_add_right", i, &d, 1); }
        { float d = rvek::dot(rvek::Vec3{a[0], a[1], a[2]}, rvek::Vec3{b[0], b[1], b[2]}); hex("dot", "header", i, &d, 1); }
        // Vec3::magnitude = sqrt(dot(v, v)) with the same candidates for the dot product
        { float d = std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); hex("magnitude", "sum", i, &d, 1); }
        { float d = std::sqrt(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2])); hex("magnitude", "sum_right", i, &d, 1); }
        { float d = std::sqrt(std::fmaf(a[2], a[2], std::fmaf(a[1], a[1], a[0] * a[0]))); hex("magnitude", "mul_add", i, &d, 1); }
        { float d = rvek::magnitude(rvek::Vec3{a[0], a[1], a[2]}); hex("magnitude", "header", i, &d, 1); }
    }
    return 0;
}
