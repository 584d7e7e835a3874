// rxr_api.hip -- implementation of the C ABI declared in include/rxr.h (host side, compiled by hipcc).
//
// One rxr_ctx == one HIP device == one process (multi-GPU hosts run one process per GPU and gather
// the rendered row bands with RCCL, see rusterix_amd/distributed.py).
//
// Frame hand-over: rxr_upload_frame validates the projected frame, packs every per-frame array into
// ONE pinned staging blob and issues ONE host->device copy; the kernels of rxr_kernels.hip then run
// entirely out of HBM.  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rusterix_vek.hpp"  // host-side Mat4 products for the device-projection path
#include "rxr_ctx.h"
#include "rxr_parallel.h"

// sparse frames (rxr_ctx::content_row0 / 1, row spans): the empty tiles a clamp must save before its fill launches pay (RXR_CONTENT_MIN_TILES
// overrides it -- the tests use small frames)
static size_t content_min_tiles() {
    const char *e = getenv("RXR_CONTENT_MIN_TILES");
    return e ? (size_t)atol(e) : 8192u;
}
thread_local LaunchTimes *rxr_launch_times = nullptr;  // rxr_launch.h: the profiling slot of the render this thread is queueing
extern "C" void rxr_launch_proj_static(const ProjectParams *P, hipStream_t s);
extern "C" void rxr_launch_project(const ProjectParams *P, hipStream_t s);
extern "C" void rxr_launch_proj_edges(const ProjectParams *P, hipStream_t s);
extern "C" void rxr_launch_setup(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_scan(const ScanArgs *A, hipStream_t s);
extern "C" void rxr_launch_bin2d_count(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_bin2d_fill(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_fill(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_blockscan(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_blockscan2d(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_raster(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_fill_words(uint32_t *dst, uint64_t n_words, uint32_t value, hipStream_t s);
extern "C" void rxr_launch_fill_outside_spans(const RasterParams *P, hipStream_t s);
extern "C" void rxr_launch_spans_from_meshes(const RasterParams *P, uint32_t n_tile_rows, const uint32_t *d2_box, uint2 *host_copy, hipStream_t s);
extern "C" uint32_t rxr_span_meshes_max(void);
extern "C" void rxr_launch_raster_grid(const RasterParams *P, uint32_t grid_x, hipStream_t s);
extern "C" int rxr_raster_takes_spans(const RasterParams *P);
extern "C" void rxr_launch_selftest_math(uint64_t seed, uint32_t blocks, uint32_t iters, unsigned long long *mismatch, hipStream_t s);

namespace {

thread_local std::string g_create_error;

}  // namespace

int rxr_fail(rxr_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}

// Nothing this context has queued may still be running when the host rewrites the pinned staging blob, the frame blob is
// overwritten by the next host->device copy, or a scratch buffer is reallocated.  Renders issued through
// rxr_render_rows_to / rxr_render_stripes_to run on the CALLER's stream (ctx->last_stream): that one is waited for as well.
int rxr_quiesce(rxr_ctx *ctx) {
    if (ctx->last_stream && ctx->last_stream != ctx->stream) HIPCHK(ctx, hipStreamSynchronize(ctx->last_stream));
    if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RXR_OK;
}

int rxr_ensure(rxr_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return RXR_OK;
    if (b.p) {
        int rc = rxr_quiesce(ctx);
        if (rc != RXR_OK) return rc;
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&b.p, cap);
    if (e != hipSuccess) return rxr_fail(ctx, RXR_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
    b.cap = cap;
    return RXR_OK;
}

// Execution fields whose lanes the raster loops do not (all) assign before each call (rasterizer.rs:773-785, :1259-1298,
// :1637-1662): a read sees what an EARLIER fragment's program left there unless this invocation wrote the field first
enum : uint32_t { PF_UV = 1, PF_ROUGHNESS = 2, PF_METALLIC = 4, PF_OPACITY = 8, PF_BUMP = 16, PF_NORMAL = 32, PF_HITPOINT = 64, PF_EMISSIVE = 128 };
// DevProgram.flags
enum : uint32_t {
    PG_WRITES_OPACITY = 1,     // contains SetOpacity: an opaque-pass batch running it needs the program in the visibility loop
    PG_WRITES_EMISSIVE = 2,    // contains SetEmissive (anywhere, callees included)
    PG_ASSIGNS_EMISSIVE = 4,   // `shade` executes a SetEmissive on EVERY path to its end (definite assignment, see PurityCheck)
};

namespace {

int fail(rxr_ctx *ctx, int code, const std::string &msg) { return rxr_fail(ctx, code, msg); }

int ensure(rxr_ctx *ctx, DevBuf &b, size_t bytes) { return rxr_ensure(ctx, b, bytes); }

int ensure_stage(rxr_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->h_stage_cap) return RXR_OK;
    if (ctx->h_stage) {
        int rc = rxr_quiesce(ctx);
        if (rc != RXR_OK) return rc;
        HIPCHK(ctx, hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr;
        ctx->h_stage_cap = 0;
    }
    size_t cap = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(&ctx->h_stage, cap, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, RXR_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    ctx->h_stage_cap = cap;
    return RXR_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

uint32_t pack_px(const uint8_t p[4]) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Rust `x as isize` narrowed to i32 for the Bresenham end points (rasterizer.rs:1785-1788);
// coordinates beyond +-2^30 are rejected at upload (the walk would not terminate in a frame's time), and so is NaN: `NaN as isize` is 0,
// a point OUTSIDE the batch's bounding box (f32::min / max drop NaN, batch2d.rs:377-403), and the reference skips a batch for every
// tile its box does not meet (:594-600) -- which pixels of such a segment it draws depends on its tile size (found by
// tests/test_gpu_special_2d.py); the caller's CPU path draws it
bool to_isize32(float x, int32_t &out) {
    if (!(x == x)) {
        out = 0;
        return false;
    }
    if (x <= -1073741824.0f || x >= 1073741824.0f) return false;
    out = (int32_t)x;
    return true;
}

// `x as u32` (Rust: saturating, NaN -> 0), as the device's sat_u32
static uint32_t rust_as_u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// The LightFast record of one light (rxr_device.h): the fragment-independent factors of the relaxed point-light term, by the
// reference's own operations (CompiledLight::apply_flicker, light.rs:506-527; smoothstep(end, start, d), light.rs:545).
static void light_fast_record(const rxr_light &l, uint32_t hash_anim, LightFast &out) {
    memset(&out, 0, sizeof(out));
    memcpy(out.pos, l.position, 12);
    out.end_distance = l.end_distance;
    out.cull_kind = (l.light_type == RXR_LIGHT_POINT || l.light_type == RXR_LIGHT_SPOT || l.light_type == RXR_LIGHT_AREA || l.light_type == RXR_LIGHT_DAYLIGHT) ? 1u : 0u;
    const float ssd = l.start_distance - l.end_distance;
    const float a = std::fabs(ssd);
    // (the window of rxr_exact_math.h: 2^-40 .. 2^40; the fused form also wants start < end, as every real light has it)
    if (l.light_type != RXR_LIGHT_POINT || !l.emitting || !(a >= 0x1p-40f && a <= 0x1p40f) || !(l.start_distance < l.end_distance)) return;
    float ff = 1.0f;
    if (l.flicker > 0.0f) {
        const uint32_t combined = hash_anim + (rust_as_u32(l.position[0]) + rust_as_u32(l.position[1]) + rust_as_u32(l.position[2])) * 100u;
        float fv = (float)combined / 4294967296.0f;
        fv = fv < 0.0f ? 0.0f : (fv > 1.0f ? 1.0f : fv);
        ff = 1.0f - fv * l.flicker;
    }
    out.ss_r = 1.0f / ssd;
    out.c0 = -l.end_distance * out.ss_r;
    for (int k = 0; k < 3; ++k) out.cfi[k] = l.color[k] * l.intensity * ff;
}

struct Layout {
    size_t off_b3, off_base, off_pv, off_uv, off_nrm, off_idx, off_edges, off_tinfo, off_clip3d, off_lights, off_lights_fast, off_occ, off_ld, off_chunks, off_tdesc,
        off_ltex, off_b2, off_p2, off_bg, total;
};

}  // namespace

extern "C" {

int rxr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *rxr_last_error(const rxr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rxr_create(rxr_ctx **out, int device_id) {
    if (!out) return fail(nullptr, RXR_ERR_INVALID, "rxr_create: out is NULL");
    *out = nullptr;
    int n = rxr_device_count();
    if (n <= 0) return fail(nullptr, RXR_ERR_NO_DEVICE, "rxr_create: no HIP device visible (there is no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, RXR_ERR_NO_DEVICE, "rxr_create: device id out of range");
    rxr_ctx *ctx = new rxr_ctx();
    ctx->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev2);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_upload);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_render, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    for (hipEvent_t &ev : ctx->ev_band)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_counters, HS_WORDS * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_row_spans, 2u * RXR_MAX_TILE_ROWS * sizeof(uint2), hipHostMallocDefault);  // (second half: the table as the device completed it, rxr_render_download)
    if (e != hipSuccess) {
        std::string msg = std::string("rxr_create: ") + hipGetErrorString(e);
        rxr_destroy(ctx);
        return fail(nullptr, RXR_ERR_HIP, msg);
    }
    memset(ctx->h_counters, 0, HS_WORDS * sizeof(uint32_t));
    if (const char *sm = getenv("RXR_SMALL_MODE")) {
        if (sm[0] >= '0' && sm[0] <= '2') ctx->small_mode = (uint32_t)(sm[0] - '0');
    }
    if (const char *lf = getenv("RXR_LIST_CAPACITY_FLOOR")) {  // tests: a small floor makes ordinary scenes overflow their bin lists
        const long v = atol(lf);
        if (v > 0) ctx->list_floor = (size_t)v;
    }
    if (const char *kl = getenv("RXR_MIN_KERNEL_LEVEL")) {  // A-B runs: render with k_raster_chunk (1) / k_raster_vm (2) regardless
        if (kl[0] >= '0' && kl[0] <= '2') ctx->min_kernel_level = (uint32_t)(kl[0] - '0');
    }
    if (hipHostGetDevicePointer((void **)&ctx->d_host_status, ctx->h_counters, 0) != hipSuccess) ctx->d_host_status = ctx->h_counters;
    *out = ctx;
    return RXR_OK;
}

int rxr_set_light_math(rxr_ctx *ctx, int mode) {
    if (!ctx) return fail(nullptr, RXR_ERR_INVALID, "rxr_set_light_math: ctx is NULL");
    if (mode != RXR_LIGHT_MATH_EXACT && mode != RXR_LIGHT_MATH_RELAXED) return fail(ctx, RXR_ERR_INVALID, "rxr_set_light_math: mode must be RXR_LIGHT_MATH_EXACT or RXR_LIGHT_MATH_RELAXED");
    if (ctx->group) {
        for (int i = 0; i < rxr_member_count(ctx); ++i) rxr_member(ctx, i)->relaxed_lights = mode == RXR_LIGHT_MATH_RELAXED;
    }
    ctx->relaxed_lights = mode == RXR_LIGHT_MATH_RELAXED;
    return RXR_OK;
}

void rxr_destroy(rxr_ctx *ctx) {
    if (ctx && !ctx->group && ctx->fstream.table) {
        (void)hipSetDevice(ctx->device);
        (void)hipHostFree(ctx->fstream.table);
        ctx->fstream.table = nullptr;
    }
    if (!ctx) return;
    if (ctx->group) {
        rxr_group_destroy(ctx);
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)rxr_quiesce(ctx);
    DevBuf *bufs[] = {&ctx->d_stripes, &ctx->d_obj, &ctx->d_proj_out, &ctx->d_proj_misc, &ctx->d_tex, &ctx->d_texels, &ctx->d_frame, &ctx->d_tri_setup, &ctx->d_tri_shade, &ctx->d_tri_box, &ctx->d_bin_count, &ctx->d_bins, &ctx->d_bin2d_count, &ctx->d_bins2d,
                      &ctx->d_list2d, &ctx->d_large2d,
                      &ctx->d_list, &ctx->d_large, &ctx->d_counters, &ctx->d_fb,
                      &ctx->d_vm_code, &ctx->d_programs, &ctx->d_patterns, &ctx->d_pattern_data, &ctx->d_palette};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    rxr_jit_drop(ctx);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    if (ctx->h_row_spans) (void)hipHostFree(ctx->h_row_spans);
    if (ctx->d_row_spans.p) (void)hipFree(ctx->d_row_spans.p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->ev_render) (void)hipEventDestroy(ctx->ev_render);
    for (hipEvent_t ev : ctx->ev_band)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (ProfSlot &p : ctx->prof)
        for (hipEvent_t e : p.ev)
            if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rxr_set_textures(rxr_ctx *ctx, const rxr_tile *static_tiles, uint32_t n_static, const rxr_tile *dynamic_tiles,
                     uint32_t n_dynamic) {
    if (!ctx) return RXR_ERR_INVALID;
    if ((n_static && !static_tiles) || (n_dynamic && !dynamic_tiles)) return fail(ctx, RXR_ERR_INVALID, "rxr_set_textures: NULL tile array");
    if (ctx->group) return rxr_group_set_textures(ctx, static_tiles, n_static, dynamic_tiles, n_dynamic);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        int qrc = rxr_quiesce(ctx);  // a render (possibly on the caller's stream) may still read the old texels
        if (qrc != RXR_OK) return qrc;
    }
    ctx->h_tex.clear();
    ctx->tiles_static.clear();
    ctx->tiles_dynamic.clear();
    size_t texels = 0;
    auto scan = [&](const rxr_tile *tiles, uint32_t n, std::vector<TileRange> &out) -> int {
        for (uint32_t i = 0; i < n; ++i) {
            TileRange r{(uint32_t)ctx->h_tex.size(), tiles[i].n_textures};
            if (tiles[i].n_textures && !tiles[i].textures) return RXR_ERR_INVALID;
            for (uint32_t k = 0; k < tiles[i].n_textures; ++k) {
                const rxr_texture &t = tiles[i].textures[k];
                if (!t.rgba || t.width == 0 || t.height == 0 || t.width > 32768 || t.height > 32768) return RXR_ERR_INVALID;
                DevTexDesc d{};
                d.offset = (uint32_t)texels;
                d.w = t.width;
                d.h = t.height;
                bool opaque = true;
                const uint8_t *p = t.rgba;
                size_t n_px = (size_t)t.width * t.height;
                for (size_t q = 0; q < n_px; ++q)
                    if (p[q * 4 + 3] != 255) {
                        opaque = false;
                        break;
                    }
                d.all_opaque = opaque ? 1u : 0u;
                ctx->h_tex.push_back(d);
                texels += align_up(n_px, 4);
                if (texels >= (1ull << 32)) return RXR_ERR_INVALID;
            }
            out.push_back(r);
        }
        return RXR_OK;
    };
    if (scan(static_tiles, n_static, ctx->tiles_static) != RXR_OK || scan(dynamic_tiles, n_dynamic, ctx->tiles_dynamic) != RXR_OK)
        return fail(ctx, RXR_ERR_INVALID, "rxr_set_textures: bad texture (NULL data, zero or oversized extent)");

    size_t desc_bytes = ctx->h_tex.size() * sizeof(DevTexDesc);
    size_t bytes = texels * 4;
    int rc;
    if ((rc = ensure(ctx, ctx->d_tex, desc_bytes ? desc_bytes : 16)) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_texels, bytes ? bytes : 16)) != RXR_OK) return rc;
    if ((rc = ensure_stage(ctx, bytes + desc_bytes + 64)) != RXR_OK) return rc;
    // stage texels then descriptors
    uint8_t *st = (uint8_t *)ctx->h_stage;
    size_t ti = 0;
    auto stage = [&](const rxr_tile *tiles, uint32_t n) {
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t k = 0; k < tiles[i].n_textures; ++k) {
                const rxr_texture &t = tiles[i].textures[k];
                memcpy(st + (size_t)ctx->h_tex[ti].offset * 4, t.rgba, (size_t)t.width * t.height * 4);
                ++ti;
            }
    };
    stage(static_tiles, n_static);
    stage(dynamic_tiles, n_dynamic);
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(ctx->d_texels.p, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (desc_bytes) {
        memcpy(st + bytes, ctx->h_tex.data(), desc_bytes);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_tex.p, st + bytes, desc_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RXR_OK;
}

int rxr_set_meshes(rxr_ctx *ctx, const rxr_mesh3d *meshes, uint32_t n_meshes) {
    if (!ctx) return RXR_ERR_INVALID;
    if (n_meshes && !meshes) return fail(ctx, RXR_ERR_INVALID, "rxr_set_meshes: NULL mesh array");
    if (ctx->group) return rxr_group_set_meshes(ctx, meshes, n_meshes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        int qrc = rxr_quiesce(ctx);
        if (qrc != RXR_OK) return qrc;
    }
    ctx->meshes.clear();
    ctx->has_frame = false;
    size_t vin = 0, tin = 0, vout = 0, tout = 0;
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const rxr_mesh3d &m = meshes[i];
        if (m.n_vertices && (!m.vertices || !m.uvs)) return fail(ctx, RXR_ERR_INVALID, "mesh: NULL vertex arrays");
        if (m.n_triangles && !m.indices) return fail(ctx, RXR_ERR_INVALID, "mesh: NULL indices");
        if (m.n_triangles && !m.normals)
            return fail(ctx, RXR_ERR_INVALID, "mesh without normals (clip_and_project panics at batch3d.rs:605)");
        if (m.cull_mode > RXR_CULL_BACK) return fail(ctx, RXR_ERR_INVALID, "mesh: bad cull mode");
        for (size_t t = 0; t < (size_t)m.n_triangles * 3u; ++t)
            if (m.indices[t] >= m.n_vertices) return fail(ctx, RXR_ERR_INVALID, "mesh: vertex index out of range");
        HostMesh h{};
        h.dev.vin_base = (uint32_t)vin;
        h.dev.tin_base = (uint32_t)tin;
        h.dev.n_verts = m.n_vertices;
        h.dev.n_tris = m.n_triangles;
        h.dev.vout_base = (uint32_t)vout;
        h.dev.tout_base = (uint32_t)tout;
        h.dev.cull_mode = m.cull_mode;
        memcpy(h.transform, m.transform_3d, 64);
        // object-space AABB with f32::min / f32::max semantics (batch3d.rs:494-507)
        for (int k = 0; k < 3; ++k) {
            h.aabb_lo[k] = INFINITY;
            h.aabb_hi[k] = -INFINITY;
        }
        for (uint32_t v = 0; v < m.n_vertices; ++v)
            for (int k = 0; k < 3; ++k) {
                h.aabb_lo[k] = std::fmin(h.aabb_lo[k], m.vertices[4 * (size_t)v + k]);
                h.aabb_hi[k] = std::fmax(h.aabb_hi[k], m.vertices[4 * (size_t)v + k]);
            }
        h.has_vertices = m.n_vertices > 0;
        h.repeat_mode = m.repeat_mode;
        h.source = m.source;
        memcpy(h.ambient, m.ambient_color, 12);
        h.shader = m.shader;
        h.has_profile_id = m.has_profile_id;
        h.profile_id = m.profile_id;
        h.list = m.list;
        h.chunk = m.chunk;
        ctx->meshes.push_back(h);
        vin += m.n_vertices;
        tin += m.n_triangles;
        vout += (size_t)m.n_vertices + 4 * (size_t)m.n_triangles;
        tout += 3 * (size_t)m.n_triangles;
    }
    if (vout >= (1ull << 31) || tout >= (1ull << 31)) return fail(ctx, RXR_ERR_INVALID, "meshes too large (>= 2^31 output slots)");
    ctx->mesh_verts_out = vout;
    ctx->mesh_tris_out = tout;

    // ---- object-space pools + static prefix arrays: one staging blob, one copy ----
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + (bytes ? bytes : 16), 256);
        return at;
    };
    const size_t off_v = take(vin * 16), off_i = take(tin * 12), off_uv = take(vin * 8), off_n = take(vin * 12);
    const size_t off_pv = take((n_meshes + 1) * 4), off_pt = take((n_meshes + 1) * 4), off_po = take((n_meshes + 1) * 4);
    const size_t off_dm = take((size_t)n_meshes * sizeof(DevMesh));
    const size_t obj_total = o;
    int rc;
    if ((rc = ensure_stage(ctx, obj_total)) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_obj, obj_total)) != RXR_OK) return rc;
    uint8_t *st = (uint8_t *)ctx->h_stage;
    uint32_t *pv = (uint32_t *)(st + off_pv), *pt = (uint32_t *)(st + off_pt), *po = (uint32_t *)(st + off_po);
    DevMesh *dm = (DevMesh *)(st + off_dm);
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const rxr_mesh3d &m = meshes[i];
        const HostMesh &h = ctx->meshes[i];
        pv[i] = h.dev.vin_base;
        pt[i] = h.dev.tin_base;
        po[i] = h.dev.tout_base;
        dm[i] = h.dev;
        if (m.n_vertices) {
            memcpy(st + off_v + (size_t)h.dev.vin_base * 16, m.vertices, (size_t)m.n_vertices * 16);
            memcpy(st + off_uv + (size_t)h.dev.vin_base * 8, m.uvs, (size_t)m.n_vertices * 8);
            if (m.normals) memcpy(st + off_n + (size_t)h.dev.vin_base * 12, m.normals, (size_t)m.n_vertices * 12);
            else memset(st + off_n + (size_t)h.dev.vin_base * 12, 0, (size_t)m.n_vertices * 12);
        }
        if (m.n_triangles) memcpy(st + off_i + (size_t)h.dev.tin_base * 12, m.indices, (size_t)m.n_triangles * 12);
    }
    pv[n_meshes] = (uint32_t)vin;
    pt[n_meshes] = (uint32_t)tin;
    po[n_meshes] = (uint32_t)tout;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_obj.p, st, obj_total, hipMemcpyHostToDevice, ctx->stream));

    // ---- output pools (the arrays k_setup3d reads) + scratch ----
    o = 0;
    const size_t q_vs = take(vout * 16), q_pv = take(vout * 16), q_uv = take(vout * 8), q_nrm = take(vout * 12);
    const size_t q_idx = take(tout * 12), q_edges = take(tout * sizeof(rxr_edges));
    const size_t out_total = o;
    if ((rc = ensure(ctx, ctx->d_proj_out, out_total)) != RXR_OK) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_proj_out.p, 0, out_total, ctx->stream));  // indices 0 / edges invisible until written
    o = 0;
    const size_t n_chunks = (tin + 1 + RXR_PROJ_SCAN_CHUNK - 1) / RXR_PROJ_SCAN_CHUNK + 1;
    const size_t m_evis = take(tin + 1), m_app = take((tin + 1) * 8), m_ct = take(n_chunks * 8), m_cb = take(n_chunks * 8);
    const size_t m_ticket = take(16), m_bbox = take((size_t)n_meshes * sizeof(DevBBox));
    const size_t m_dm = take((size_t)n_meshes * sizeof(DevMesh));
    const size_t m_live = take((size_t)n_meshes * sizeof(uint32_t));
    if ((rc = ensure(ctx, ctx->d_proj_misc, o)) != RXR_OK) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_proj_misc.p, 0, o, ctx->stream));
    ctx->pp_off_meshes = m_dm;

    ProjectParams &PP = ctx->PP;
    memset(&PP, 0, sizeof(PP));
    PP.n_meshes = n_meshes;
    PP.n_verts_in = (uint32_t)vin;
    PP.n_tris_in = (uint32_t)tin;
    PP.n_tris_out = (uint32_t)tout;
    uint8_t *d = (uint8_t *)ctx->d_obj.p, *q = (uint8_t *)ctx->d_proj_out.p, *mm = (uint8_t *)ctx->d_proj_misc.p;
    PP.meshes = (const DevMesh *)(d + off_dm);  // static copy; replaced by the per-frame array at render time
    PP.vin_prefix = (const uint32_t *)(d + off_pv);
    PP.tin_prefix = (const uint32_t *)(d + off_pt);
    PP.tout_prefix = (const uint32_t *)(d + off_po);
    PP.obj_verts = (const float4 *)(d + off_v);
    PP.obj_idx = (const uint32_t *)(d + off_i);
    PP.obj_uvs = (const float2 *)(d + off_uv);
    PP.obj_normals = (const float *)(d + off_n);
    PP.view_verts = (float4 *)(q + q_vs);
    PP.pv = (float4 *)(q + q_pv);
    PP.uv = (float2 *)(q + q_uv);
    PP.nrm = (float *)(q + q_nrm);
    PP.idx = (uint32_t *)(q + q_idx);
    PP.edges = (rxr_edges *)(q + q_edges);
    PP.edge_vis = (uint8_t *)(mm + m_evis);
    PP.append = (AppendCount *)(mm + m_app);
    PP.chunk_tot = (AppendCount *)(mm + m_ct);
    PP.chunk_base = (AppendCount *)(mm + m_cb);
    PP.ticket = (uint32_t *)(mm + m_ticket);
    PP.bbox = (DevBBox *)(mm + m_bbox);
    PP.mesh_live = (uint32_t *)(mm + m_live);
    rxr_launch_proj_static(&PP, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RXR_OK;
}

// ---- the 2D half of the device-side projection (row N1): Batch2D::project's inputs, registered once ---------------------------
extern "C" void rxr_launch_project2d(const Project2DParams *P, hipStream_t s);

int rxr_set_meshes2d(rxr_ctx *ctx, const rxr_mesh2d *meshes, uint32_t n_meshes) {
    if (!ctx) return RXR_ERR_INVALID;
    if (n_meshes && !meshes) return fail(ctx, RXR_ERR_INVALID, "rxr_set_meshes2d: NULL mesh array");
    if (ctx->group) return rxr_group_set_meshes2d(ctx, meshes, n_meshes);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        int qrc = rxr_quiesce(ctx);
        if (qrc != RXR_OK) return qrc;
    }
    // A call that fails half way must leave an EMPTY registration, never a partial list beside the previous call's device data
    // (round-3 advisor finding: a later frame with use_meshes bit 1 would have built batch headers from the partial list while the
    // projection kernels still named the old mesh indices): everything that describes the registration is reset before the first check.
    ctx->meshes2d.clear();
    ctx->meshes2d_prims = 0;
    ctx->meshes2d_tris = 0;
    ctx->PP2 = Project2DParams{};
    ctx->has_frame = false;
    struct EmptyOnFailure {
        rxr_ctx *c;
        bool ok = false;
        ~EmptyOnFailure() {
            if (!ok) {
                c->meshes2d.clear();
                c->meshes2d_prims = c->meshes2d_tris = 0;
                c->PP2 = Project2DParams{};
            }
        }
    } registration{ctx};
    size_t vin = 0, prims = 0, tris = 0;
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const rxr_mesh2d &m = meshes[i];
        if (m.n_vertices && (!m.vertices || !m.uvs)) return fail(ctx, RXR_ERR_INVALID, "mesh2d: NULL vertex arrays");
        if (m.n_triangles && !m.indices) return fail(ctx, RXR_ERR_INVALID, "mesh2d: NULL indices");
        if (m.mode > RXR_MODE_LINE_LOOP) return fail(ctx, RXR_ERR_INVALID, "mesh2d: bad mode");
        if (m.mode == RXR_MODE_TRIANGLES || m.mode == RXR_MODE_LINES)
            for (size_t t = 0; t < (size_t)m.n_triangles * 3u; ++t) {
                if (m.mode == RXR_MODE_LINES && (t % 3u) == 2u) continue;  // only .0/.1 are read, :902
                if (m.indices[t] >= m.n_vertices) return fail(ctx, RXR_ERR_INVALID, "mesh2d: vertex index out of range");
            }
        rxr_ctx::HostMesh2D h{};
        h.vin_base = (uint32_t)vin;
        h.n_verts = m.n_vertices;
        h.n_tris = m.n_triangles;
        switch (m.mode) {
            case RXR_MODE_TRIANGLES: h.n_prims = m.n_triangles; tris += m.n_triangles; break;
            case RXR_MODE_LINES: h.n_prims = m.n_triangles; break;
            case RXR_MODE_LINE_STRIP: h.n_prims = m.n_vertices ? m.n_vertices - 1 : 0; break;
            default: h.n_prims = m.n_vertices; break;
        }
        h.prim_base = (uint32_t)prims;
        h.mode = m.mode;
        h.repeat_mode = m.repeat_mode;
        h.receives_light = m.receives_light;
        h.source = m.source;
        h.shader = m.shader;
        h.chunk = m.chunk;
        ctx->meshes2d.push_back(h);
        vin += m.n_vertices;
        prims += h.n_prims;
    }
    if (vin >= (1ull << 31) || prims >= (1ull << 30)) return fail(ctx, RXR_ERR_INVALID, "2D meshes too large");
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + (bytes ? bytes : 16), 256);
        return at;
    };
    const size_t off_v = take(vin * 8), off_uv = take(vin * 8), off_src = take(prims * sizeof(Prim2DSrc)), off_pv = take((n_meshes + 1) * 4),
                 off_dm = take((size_t)n_meshes * sizeof(DevMesh2D));
    const size_t total = o;
    int rc;
    if ((rc = ensure_stage(ctx, total)) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_obj2d, total)) != RXR_OK) return rc;
    uint8_t *st = (uint8_t *)ctx->h_stage;
    uint32_t *pv = (uint32_t *)(st + off_pv);
    DevMesh2D *dm = (DevMesh2D *)(st + off_dm);
    Prim2DSrc *src = (Prim2DSrc *)(st + off_src);
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const rxr_mesh2d &m = meshes[i];
        const rxr_ctx::HostMesh2D &h = ctx->meshes2d[i];
        pv[i] = h.vin_base;
        const uint8_t white[4] = {255, 255, 255, 255};
        dm[i] = DevMesh2D{h.vin_base, h.n_verts, h.mode, pack_px(m.source.kind == RXR_SOURCE_PIXEL ? m.source.pixel : white)};  // :911-915
        if (m.n_vertices) {
            memcpy(st + off_v + (size_t)h.vin_base * 8, m.vertices, (size_t)m.n_vertices * 8);
            memcpy(st + off_uv + (size_t)h.vin_base * 8, m.uvs, (size_t)m.n_vertices * 8);
        }
        Prim2DSrc *q = src + h.prim_base;
        if (m.mode == RXR_MODE_TRIANGLES)
            for (uint32_t t = 0; t < m.n_triangles; ++t) q[t] = Prim2DSrc{i, m.indices[3 * (size_t)t], m.indices[3 * (size_t)t + 1], m.indices[3 * (size_t)t + 2]};
        else if (m.mode == RXR_MODE_LINES)
            for (uint32_t t = 0; t < m.n_triangles; ++t) q[t] = Prim2DSrc{i, m.indices[3 * (size_t)t], m.indices[3 * (size_t)t + 1], 0u};
        else if (m.mode == RXR_MODE_LINE_STRIP)
            for (uint32_t k = 0; k + 1 < m.n_vertices; ++k) q[k] = Prim2DSrc{i, k, k + 1u, 0u};
        else
            for (uint32_t k = 0; k < m.n_vertices; ++k) q[k] = Prim2DSrc{i, k, (k + 1u) % m.n_vertices, 0u};
    }
    pv[n_meshes] = (uint32_t)vin;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_obj2d.p, st, total, hipMemcpyHostToDevice, ctx->stream));
    o = 0;
    const size_t m_bbox = take((size_t)n_meshes * sizeof(DevBBox)), m_box = take(16);
    if ((rc = ensure(ctx, ctx->d_proj2d_misc, o)) != RXR_OK) return rc;
    Project2DParams &PP = ctx->PP2;
    memset(&PP, 0, sizeof(PP));
    PP.n_meshes = n_meshes;
    PP.n_verts = (uint32_t)vin;
    PP.n_prims = (uint32_t)prims;
    uint8_t *d = (uint8_t *)ctx->d_obj2d.p, *mm = (uint8_t *)ctx->d_proj2d_misc.p;
    PP.meshes = (const DevMesh2D *)(d + off_dm);
    PP.vin_prefix = (const uint32_t *)(d + off_pv);
    PP.obj_verts = (const float2 *)(d + off_v);
    PP.obj_uvs = (const float2 *)(d + off_uv);
    PP.src = (const Prim2DSrc *)(d + off_src);
    PP.bbox = (DevBBox *)(mm + m_bbox);
    PP.d2_box = (uint32_t *)(mm + m_box);
    PP.bad_line = ctx->d_host_status + HS_BAD_LINE2D;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->meshes2d_prims = prims;
    ctx->meshes2d_tris = tris;
    registration.ok = true;
    return RXR_OK;
}

int rxr_set_projection2d(rxr_ctx *ctx, const float *mat3) {
    if (!ctx) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_group_set_projection2d(ctx, mat3);
    ctx->has_matrix2d = mat3 != nullptr;
    if (mat3) memcpy(ctx->matrix2d, mat3, sizeof(ctx->matrix2d));
    return RXR_OK;
}

int rxr_read_projected_mesh(rxr_ctx *ctx, uint32_t index, uint32_t counts[2], float *projected_vertices, float *clipped_uvs,
                            float *clipped_normals, uint32_t *clipped_indices, rxr_edges *edges, float bounding_box[5],
                            uint32_t capacity_vertices, uint32_t capacity_triangles) {
    if (!ctx || !counts) return RXR_ERR_INVALID;
    if (ctx->group)
        return rxr_read_projected_mesh(rxr_member(ctx, 0), index, counts, projected_vertices, clipped_uvs, clipped_normals, clipped_indices, edges,
                                       bounding_box, capacity_vertices, capacity_triangles);
    if (index >= ctx->meshes.size()) return fail(ctx, RXR_ERR_INVALID, "rxr_read_projected_mesh: no such mesh");
    int rc = rxr_synchronize(ctx);
    if (rc != RXR_OK) return rc;
    const DevMesh &M = ctx->meshes[index].dev;
    const ProjectParams &PP = ctx->PP;
    // appended counts of this mesh = prefix(end) - prefix(start)
    auto prefix_at = [&](uint32_t i, AppendCount &out) -> int {
        AppendCount a = 0, b = 0;
        HIPCHK(ctx, hipMemcpy(&a, PP.append + i, 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(&b, PP.chunk_base + i / RXR_PROJ_SCAN_CHUNK, 8, hipMemcpyDeviceToHost));
        out = a + b;
        return RXR_OK;
    };
    AppendCount p0 = 0, p1 = 0;
    DevMesh frame_mesh{};
    HIPCHK(ctx, hipMemcpy(&frame_mesh, (uint8_t *)ctx->d_proj_misc.p + ctx->pp_off_meshes + (size_t)index * sizeof(DevMesh), sizeof(DevMesh), hipMemcpyDeviceToHost));
    uint32_t nv = 0, nt = 0;
    if (!frame_mesh.rejected) {
        if ((rc = prefix_at(M.tin_base, p0)) != RXR_OK || (rc = prefix_at(M.tin_base + M.n_tris, p1)) != RXR_OK) return rc;
        AppendCount tot = p1 - p0;
        nv = M.n_verts + (uint32_t)(tot & 0xFFFFFFFFull);
        nt = M.n_tris + (uint32_t)(tot >> 32);
    }
    counts[0] = nv;
    counts[1] = nt;
    if (nv > capacity_vertices || nt > capacity_triangles) return fail(ctx, RXR_ERR_INVALID, "rxr_read_projected_mesh: capacity too small");
    if (projected_vertices && nv) HIPCHK(ctx, hipMemcpy(projected_vertices, PP.pv + M.vout_base, (size_t)nv * 16, hipMemcpyDeviceToHost));
    if (clipped_uvs && nv) HIPCHK(ctx, hipMemcpy(clipped_uvs, PP.uv + M.vout_base, (size_t)nv * 8, hipMemcpyDeviceToHost));
    if (clipped_normals && nv) HIPCHK(ctx, hipMemcpy(clipped_normals, PP.nrm + 3 * (size_t)M.vout_base, (size_t)nv * 12, hipMemcpyDeviceToHost));
    if (clipped_indices && nt) HIPCHK(ctx, hipMemcpy(clipped_indices, PP.idx + 3 * (size_t)M.tout_base, (size_t)nt * 12, hipMemcpyDeviceToHost));
    if (edges && nt) {
        if (PP.edges_in_setup) {  // (the frame's set-up built its records itself: the pool is filled for this call)
            HIPCHK(ctx, hipSetDevice(ctx->device));
            rxr_launch_proj_edges(&PP, ctx->stream);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        HIPCHK(ctx, hipMemcpy(edges, PP.edges + M.tout_base, (size_t)nt * sizeof(rxr_edges), hipMemcpyDeviceToHost));
    }
    if (bounding_box) {
        DevBBox bb{};
        HIPCHK(ctx, hipMemcpy(&bb, PP.bbox + index, sizeof(bb), hipMemcpyDeviceToHost));
        auto dec = [](uint32_t e) {
            uint32_t u = (e & 0x80000000u) ? (e ^ 0x80000000u) : ~e;
            float f;
            memcpy(&f, &u, 4);
            return f;
        };
        bounding_box[0] = frame_mesh.rejected ? 0.0f : 1.0f;
        bounding_box[1] = dec(bb.min_x);
        bounding_box[2] = dec(bb.min_y);
        bounding_box[3] = dec(bb.max_x) - dec(bb.min_x);
        bounding_box[4] = dec(bb.max_y) - dec(bb.min_y);
    }
    return RXR_OK;
}

// resolves a PixelSource to (tex index | constant texel); see include/rxr.h RXR_SOURCE_*
static int resolve_source(rxr_ctx *ctx, const rxr_source &src, bool is_3d, int chunk, uint64_t animation_frame, int32_t &tex,
                          uint32_t &pixel) {
    tex = -1;
    pixel = 0;
    auto from_tiles = [&](const std::vector<TileRange> &tiles) -> int {
        if (src.index >= tiles.size()) {
            if (is_3d) return RXR_ERR_INVALID;  // tile_list[index] panics, rasterizer.rs:1103
            pixel = 0;                          // 2D uses .get(): [0,0,0,0], :685-687
            return RXR_OK;
        }
        const TileRange &r = tiles[src.index];
        if (r.n == 0) return RXR_ERR_INVALID;  // `% 0` panics
        tex = (int32_t)(r.first + (uint32_t)(animation_frame % r.n));
        return RXR_OK;
    };
    switch (src.kind) {
        case RXR_SOURCE_STATIC_TILE: return from_tiles(ctx->tiles_static);
        case RXR_SOURCE_DYNAMIC_TILE: return from_tiles(ctx->tiles_dynamic);
        case RXR_SOURCE_PIXEL: pixel = pack_px(src.pixel); return RXR_OK;
        case RXR_SOURCE_MISSING: pixel = 0; return RXR_OK;
        case RXR_HOST_SOURCE_ENTITY_TILE:
        case RXR_HOST_SOURCE_ITEM_TILE: return RXR_ERR_INVALID;  // host-side variants: the host resolves them (include/rxr.h)
        case RXR_SOURCE_TERRAIN:
            if (chunk >= 0) pixel = 0;                       // chunk without a terrain texture, chunk.rs:150
            else pixel = is_3d ? 0xFF0000FFu : 0u;           // [255,0,0,255] (:1218) | [0,0,0,0] (:753)
            return RXR_OK;
        default: pixel = is_3d ? 0xFF000000u : 0u; return RXR_OK;  // [0,0,0,255] (:1221) | [0,0,0,0] (:757)
    }
}

// rxr_stream_begin_pinned: the device pulls a group of batches out of the host's own (page-locked) arrays.  One workgroup per 16 KB
// piece; an entry's pieces are consecutive (first_piece), the workgroup finds its entry by bisection in the (pinned) table.  Sources
// and destinations are 4-byte aligned (destinations are v0 * 12 / t0 * 12 bytes into a pool): dword copies, 256 bytes per wave
// instruction, sixteen independent loads per lane in flight over PCIe.
#define RXR_GATHER_PIECE 16384u
extern "C" __global__ void __launch_bounds__(256) k_gather_host(const FrameStream::GatherEntry *entries, uint32_t n_entries, uint8_t *blob) {
    const uint32_t piece = blockIdx.x;
    uint32_t lo = 0, hi = n_entries;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (entries[mid].first_piece <= piece) lo = mid;
        else hi = mid;
    }
    const FrameStream::GatherEntry E = entries[lo];
    const uint32_t at = (piece - E.first_piece) * RXR_GATHER_PIECE;  // byte offset of this piece inside the entry
    const uint32_t n_bytes = min(E.bytes - at, RXR_GATHER_PIECE);
    const uint8_t *src = (const uint8_t *)E.src + at;
    uint8_t *dst = blob + E.dst_off + at;
    if ((((uintptr_t)src | (uintptr_t)dst | n_bytes) & 15u) == 0u) {
        // (workgroup-uniform) both ends 16-byte aligned -- the projected vertices always are: 1 KB per wave instruction
        const uint32_t n16 = n_bytes >> 4;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = (uint32_t)k * 256u + threadIdx.x;
            v[k] = i < n16 ? __builtin_nontemporal_load((const u32x4 *)src + i) : (u32x4)(0u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = (uint32_t)k * 256u + threadIdx.x;
            if (i < n16) ((u32x4 *)dst)[i] = v[k];
        }
        return;
    }
    const uint32_t n_dw = n_bytes >> 2;
    uint32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t i = (uint32_t)k * 256u + threadIdx.x;
        v[k] = i < n_dw ? __builtin_nontemporal_load((const uint32_t *)src + i) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t i = (uint32_t)k * 256u + threadIdx.x;
        if (i < n_dw) ((uint32_t *)dst)[i] = v[k];
    }
}

void *rxr_alloc_pinned(size_t bytes) {
    void *p = nullptr;
    if (rxr_device_count() <= 0 || hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void rxr_free_pinned(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

// the frame blob up to the end of the projected arrays: headers, triangle bases, then the five pools (n_v vertices, n_t triangles)
static size_t layout_prefix(size_t n_b3, size_t n_v, size_t n_t, Layout &L) {
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + (bytes ? bytes : 16), 256);
        return at;
    };
    L.off_b3 = take(n_b3 * sizeof(DevBatch));
    L.off_base = take((n_b3 + 1) * sizeof(uint32_t));
    L.off_pv = take(n_v * 16);
    L.off_uv = take(n_v * 8);
    L.off_nrm = take(n_v * 12);
    L.off_idx = take(n_t * 12);
    L.off_edges = take(n_t * sizeof(rxr_edges));
    return o;
}

static int stream_begin(rxr_ctx *ctx, uint32_t n_batches3d, const uint32_t *vertex_capacity, const uint32_t *triangle_capacity, bool pinned);
int rxr_stream_begin(rxr_ctx *ctx, uint32_t n_batches3d, const uint32_t *vertex_capacity, const uint32_t *triangle_capacity) {
    return stream_begin(ctx, n_batches3d, vertex_capacity, triangle_capacity, false);
}
int rxr_stream_begin_pinned(rxr_ctx *ctx, uint32_t n_batches3d, const uint32_t *vertex_capacity, const uint32_t *triangle_capacity) {
    return stream_begin(ctx, n_batches3d, vertex_capacity, triangle_capacity, true);
}
static int stream_begin(rxr_ctx *ctx, uint32_t n_batches3d, const uint32_t *vertex_capacity, const uint32_t *triangle_capacity, bool pinned) {
    if (!ctx) return RXR_ERR_INVALID;
    if (ctx->group) return fail(ctx, RXR_ERR_UNSUPPORTED, "rxr_stream_begin on a multi-device context: hand the frame over with rxr_upload_frame");
    FrameStream &S = ctx->fstream;
    S.active = false;
    if (n_batches3d == 0 || !vertex_capacity || !triangle_capacity) return fail(ctx, RXR_ERR_INVALID, "rxr_stream_begin: no batches / NULL capacities");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t cv = 0, ct = 0;
    for (uint32_t i = 0; i < n_batches3d; ++i) {
        cv += vertex_capacity[i];
        ct += triangle_capacity[i];
    }
    if (cv >= (1ull << 31) || ct >= (1ull << 31)) return fail(ctx, RXR_ERR_INVALID, "rxr_stream_begin: frame too large (>= 2^31 vertices or triangles)");
    Layout L{};
    const size_t prefix = layout_prefix(n_batches3d, cv, ct, L);
    // what follows the arrays in the blob (lights, 2D primitives, chunk textures ...) is only known at rxr_upload_frame: room for what
    // the last frame needed plus a margin; a frame that needs more falls back to the plain hand-over there
    const size_t tail_room = std::max<size_t>(16u << 20, 2 * ctx->last_blob_tail);
    int rc;
    // the previous frame's renders read the blob, its transfers the staging memory: nothing of that may still run
    if ((rc = rxr_quiesce(ctx)) != RXR_OK) return rc;
    if ((rc = ensure_stage(ctx, prefix + tail_room)) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_frame, prefix + tail_room)) != RXR_OK) return rc;
    S.n = n_batches3d;
    S.rec.assign(n_batches3d, FrameStream::Rec{});
    S.cap_v.assign(vertex_capacity, vertex_capacity + n_batches3d);
    S.cap_t.assign(triangle_capacity, triangle_capacity + n_batches3d);
    S.done.reset(new std::atomic<uint8_t>[n_batches3d]);
    for (uint32_t i = 0; i < n_batches3d; ++i) S.done[i].store(0, std::memory_order_relaxed);
    // few, large transfers (every hipMemcpyAsync costs its caller ~20 us): eight groups of consecutive batches
    static const uint32_t n_groups_wanted = []() {
        const char *e = getenv("RXR_STREAM_GROUPS");   // (tuning: tools/e2e_probe.py)
        const int v = e ? atoi(e) : 0;
        return (uint32_t)(v >= 1 && v <= 256 ? v : 8);
    }();
    S.group_size = std::max<uint32_t>(1u, (n_batches3d + n_groups_wanted - 1u) / n_groups_wanted);
    S.n_groups = (n_batches3d + S.group_size - 1u) / S.group_size;
    S.group_left.reset(new std::atomic<uint32_t>[S.n_groups]);
    for (uint32_t g = 0; g < S.n_groups; ++g) S.group_left[g].store(std::min(S.group_size, n_batches3d - g * S.group_size), std::memory_order_relaxed);
    S.next = 0;
    S.retired.store(0);
    S.copy_next.store(0);
    S.vcur = S.tcur = 0;
    S.total_cap_v = cv;
    S.total_cap_t = ct;
    S.off_pv = L.off_pv; S.off_uv = L.off_uv; S.off_nrm = L.off_nrm; S.off_idx = L.off_idx; S.off_edges = L.off_edges;
    S.off_after = prefix;
    S.blob_capacity = std::min(ctx->h_stage_cap, ctx->d_frame.cap);
    S.pinned = pinned;
    if (pinned && S.table_cap < (size_t)n_batches3d * 5u) {
        if (S.table) (void)hipHostFree(S.table);
        S.table = nullptr;
        S.table_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void **)&S.table, ((size_t)n_batches3d * 5u + 64u) * sizeof(FrameStream::GatherEntry), hipHostMallocDefault));
        S.table_cap = (size_t)n_batches3d * 5u + 64u;
    }
    S.handed.store(0);
    S.failed.store(0);
    S.edgeless.store(-1);
    S.err.clear();
    ctx->has_frame = false;
    S.active = true;
    if (getenv("RXR_E2E_TIMING")) fprintf(stderr, "rxr_e2e_timing stream_begin (%s)\n", pinned ? "pinned" : "copy");
    return RXR_OK;
}

// copy mode: claims retired batches one by one, copies their arrays to their dense places in the pinned staging blob and ships a
// group's five ranges when its last batch has landed.  Any thread; `limit` bounds the batches taken by this call.
static bool rxr_stream_copy_some(rxr_ctx *ctx, uint32_t limit) {
    FrameStream &S = ctx->fstream;
    uint8_t *const st = (uint8_t *)ctx->h_stage;
    for (uint32_t taken = 0; taken < limit; ++taken) {
        uint32_t j = S.copy_next.load(std::memory_order_relaxed);
        for (;;) {
            if (j >= S.retired.load(std::memory_order_acquire)) return true;
            if (S.copy_next.compare_exchange_weak(j, j + 1u, std::memory_order_relaxed)) break;
        }
        const FrameStream::Rec &Q = S.rec[j];
        if (Q.nv) {
            memcpy(st + S.off_pv + Q.v0 * 16, Q.pv, (size_t)Q.nv * 16);
            memcpy(st + S.off_uv + Q.v0 * 8, Q.uv, (size_t)Q.nv * 8);
            if (Q.nrm) memcpy(st + S.off_nrm + Q.v0 * 12, Q.nrm, (size_t)Q.nv * 12);
        }
        if (Q.nt) {
            memcpy(st + S.off_idx + Q.t0 * 12, Q.idx, (size_t)Q.nt * 12);
            memcpy(st + S.off_edges + Q.t0 * S.tri5(), Q.edges, (size_t)Q.nt * S.tri5());
        }
        const uint32_t g = j / S.group_size;
        if (S.group_left[g].fetch_sub(1u, std::memory_order_acq_rel) == 1u) {
            // the group's last batch has landed in pinned memory: its five ranges go to the device
            const FrameStream::Rec &A = S.rec[g * S.group_size], &Z = S.rec[std::min(S.n, (g + 1u) * S.group_size) - 1u];
            const size_t v0 = A.v0, nv = Z.v0 + Z.nv - A.v0, t0 = A.t0, nt = Z.t0 + Z.nt - A.t0;
            const struct { size_t off, bytes; } r[5] = {{S.off_pv + v0 * 16, nv * 16}, {S.off_uv + v0 * 8, nv * 8}, {S.off_nrm + v0 * 12, nv * 12},
                                                        {S.off_idx + t0 * 12, nt * 12}, {S.off_edges + t0 * S.tri5(), nt * S.tri5()}};
            std::lock_guard<std::mutex> lk(S.ship_mu);
            hipError_t e = hipSetDevice(ctx->device);
            for (const auto &x : r)
                if (x.bytes && e == hipSuccess) e = hipMemcpyAsync((uint8_t *)ctx->d_frame.p + x.off, st + x.off, x.bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> lk2(S.mu);
                if (S.err.empty()) S.err = "rxr_stream_batch3d: host->device copy failed";
                S.failed.store(1);
                return false;
            }
        }
    }
    return true;
}

// thread-safe; never touches ctx->err (the failure is kept in the stream and reported by rxr_upload_frame's fall-back)
int rxr_stream_batch3d(rxr_ctx *ctx, uint32_t index, const rxr_batch3d *b) {
    if (!ctx || !b || ctx->group) return RXR_ERR_INVALID;
    FrameStream &S = ctx->fstream;
    if (!S.active || index >= S.n) return RXR_ERR_INVALID;
    static const bool timing = getenv("RXR_E2E_TIMING") != nullptr;
    struct Tick {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        bool on;
        uint32_t index;
        ~Tick() {
            if (!on) return;
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            static std::atomic<uint64_t> total{0}, worst{0}, calls{0};
            total += (uint64_t)us;
            uint64_t w = worst.load();
            while ((uint64_t)us > w && !worst.compare_exchange_weak(w, (uint64_t)us)) {}
            if (++calls % 289 == 0) fprintf(stderr, "rxr_e2e_timing stream_batch3d: %llu calls, %.1f us each on average, worst %llu us\n", (unsigned long long)calls.load(), (double)total.load() / (double)calls.load(), (unsigned long long)worst.load());
        }
    } tick;
    tick.on = timing;
    tick.index = index;
    auto give_up = [&](const char *why) {
        std::lock_guard<std::mutex> lk(S.mu);
        if (S.err.empty()) S.err = why;
        S.failed.store(1);
        return RXR_ERR_INVALID;
    };
    if (S.failed.load()) return RXR_ERR_INVALID;
    if (S.done[index].load(std::memory_order_acquire)) return give_up("rxr_stream_batch3d: batch handed over twice");
    if (b->n_vertices > S.cap_v[index] || b->n_triangles > S.cap_t[index]) return give_up("rxr_stream_batch3d: batch exceeds the capacity announced to rxr_stream_begin");
    if ((b->n_triangles && (!b->clipped_indices || (!b->edges && !b->edge_visible))) || (b->n_vertices && (!b->projected_vertices || !b->clipped_uvs)))
        return give_up("rxr_stream_batch3d: NULL arrays");
    if (b->n_triangles) {  // with Edges records or without (ABI 5): one form per frame, fixed by the first batch that has triangles
        const int form = b->edges ? 0 : 1;
        int seen = -1;
        if (!S.edgeless.compare_exchange_strong(seen, form) && seen != form) return give_up("rxr_stream_batch3d: batches with and without Edges records in one frame");
        if (form == 1 && b->cull_mode > RXR_CULL_BACK) return give_up("rxr_stream_batch3d: bad cull mode");
    }
    const void *const fifth = b->edges ? (const void *)b->edges : (const void *)b->edge_visible;
    const size_t fifth_bytes = (size_t)b->n_triangles * (b->edges ? sizeof(rxr_edges) : sizeof(uint32_t));
    {
        uint32_t worst = 0;  // every index (the kernels trust them)
        for (size_t t = 0; t < (size_t)b->n_triangles * 3u; ++t) worst = std::max(worst, b->clipped_indices[t]);
        if (b->n_triangles && worst >= b->n_vertices) return give_up("batch3d: vertex index out of range");
    }
    const void *dev_ptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (S.pinned) {
        // the promise is VERIFIED, array by array, first and last byte: a pointer the device cannot read would be a GPU page fault
        // (which can take the whole node down), so it costs the caller a frame handed over the plain way instead.  The kernel reads
        // through the DEVICE address the runtime reports for the array (round-3 advisor finding: for memory locked with hipHostRegister
        // / rxr_pin_host_buffer the device's address of a page need not be the host's; hipHostMalloc / rxr_alloc_pinned memory has one
        // address for both), and only when first and last byte lie `bytes - 1` apart there too: one registration, mapped in one piece.
        const struct { const void *p; size_t bytes; } arr[5] = {{b->projected_vertices, (size_t)b->n_vertices * 16}, {b->clipped_uvs, (size_t)b->n_vertices * 8},
                                                                {b->clipped_normals, b->clipped_normals ? (size_t)b->n_vertices * 12 : 0},
                                                                {b->clipped_indices, (size_t)b->n_triangles * 12}, {fifth, fifth_bytes}};
        for (int i = 0; i < 5; ++i) {
            const auto &x = arr[i];
            if (!x.bytes) continue;
            hipPointerAttribute_t a0{}, a1{};
            const hipError_t e0 = hipPointerGetAttributes(&a0, x.p), e1 = hipPointerGetAttributes(&a1, (const uint8_t *)x.p + x.bytes - 1);
            if (e0 != hipSuccess || e1 != hipSuccess || a0.type != hipMemoryTypeHost || a1.type != hipMemoryTypeHost) {
                (void)hipGetLastError();
                return give_up("rxr_stream_batch3d: an array is not in page-locked, device-readable memory (the promise of rxr_stream_begin_pinned)");
            }
            if (!a0.devicePointer || !a1.devicePointer || (const uint8_t *)a1.devicePointer - (const uint8_t *)a0.devicePointer != (ptrdiff_t)(x.bytes - 1))
                return give_up("rxr_stream_batch3d: an array is page-locked but has no device address in one piece (registered in parts, or not mapped)");
            dev_ptr[i] = a0.devicePointer;
        }
    }
    FrameStream::Rec &R = S.rec[index];
    R.pv = b->projected_vertices; R.uv = b->clipped_uvs; R.nrm = b->clipped_normals; R.idx = b->clipped_indices; R.edges = fifth;
    R.nv = b->n_vertices; R.nt = b->n_triangles;
    for (int i = 0; i < 5; ++i) R.dev[i] = dev_ptr[i];
    S.done[index].store(1, std::memory_order_release);
    S.handed.fetch_add(1);
    // retire in index order: a batch's place in the pools is the sum of its predecessors' sizes
    uint32_t first, last;
    {
        std::lock_guard<std::mutex> lk(S.mu);
        first = S.next;
        while (S.next < S.n && S.done[S.next].load(std::memory_order_acquire)) {
            FrameStream::Rec &Q = S.rec[S.next];
            Q.v0 = S.vcur;
            Q.t0 = S.tcur;
            S.vcur += Q.nv;
            S.tcur += Q.nt;
            ++S.next;
        }
        last = S.next;
        S.retired.store(S.next, std::memory_order_release);
    }
    if (!S.pinned) {
        // copy mode: a retired batch is copied by whichever caller gets to it (the thread that retires a long run must not be the
        // one that copies all of it: the others would project on while it falls behind)
        (void)first;
        (void)last;
        return rxr_stream_copy_some(ctx, 0xFFFFFFFFu) ? RXR_OK : RXR_ERR_INVALID;
    }
    for (uint32_t j = first; j < last; ++j) {
        {
            // no host copy: when the group's last batch has retired, one kernel pulls the group out of the caller's arrays
            const uint32_t g = j / S.group_size;
            if (S.group_left[g].fetch_sub(1u, std::memory_order_acq_rel) != 1u) continue;
            const uint32_t b0 = g * S.group_size, b1 = std::min(S.n, (g + 1u) * S.group_size);
            FrameStream::GatherEntry *T = S.table + (size_t)b0 * 5u;
            uint32_t n_e = 0, piece = 0;
            for (uint32_t k = b0; k < b1; ++k) {
                const FrameStream::Rec &B = S.rec[k];
                // (sources: the arrays' DEVICE addresses, verified when the batch was handed over)
                const struct { const void *src; size_t off, bytes; } r[5] = {{B.dev[0], S.off_pv + B.v0 * 16, (size_t)B.nv * 16}, {B.dev[1], S.off_uv + B.v0 * 8, (size_t)B.nv * 8},
                                                                             {B.dev[2], S.off_nrm + B.v0 * 12, B.nrm ? (size_t)B.nv * 12 : 0}, {B.dev[3], S.off_idx + B.t0 * 12, (size_t)B.nt * 12},
                                                                             {B.dev[4], S.off_edges + B.t0 * S.tri5(), (size_t)B.nt * S.tri5()}};
                for (const auto &x : r) {
                    if (!x.bytes) continue;
                    T[n_e++] = FrameStream::GatherEntry{x.src, (uint64_t)x.off, (uint32_t)x.bytes, piece};
                    piece += (uint32_t)((x.bytes + RXR_GATHER_PIECE - 1u) / RXR_GATHER_PIECE);
                }
            }
            if (n_e) {
                std::lock_guard<std::mutex> lk(S.ship_mu);
                hipError_t e = hipSetDevice(ctx->device);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(k_gather_host, dim3(piece), dim3(256), 0, ctx->stream, (const FrameStream::GatherEntry *)T, n_e, (uint8_t *)ctx->d_frame.p);
                    e = hipGetLastError();
                }
                if (e != hipSuccess) return give_up("rxr_stream_batch3d: the gather launch failed");
            }
        }
    }
    return RXR_OK;
}

int rxr_upload_frame(rxr_ctx *ctx, const rxr_frame *f) {
    if (!ctx) return RXR_ERR_INVALID;
    if (!f) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: frame is NULL");
    if (ctx->group) return rxr_group_upload_frame(ctx, f);
    if (f->abi_version != RXR_ABI_VERSION) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: abi_version mismatch");
    if (f->width == 0 || f->height == 0 || f->width > 32768 || f->height > 32768)
        return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: width/height must be in 1..32768");
    if (f->tile_size == 0) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: tile_size 0 (step_by(0) panics in the reference)");
    if ((f->n_batches3d && !f->batches3d) || (f->n_batches2d && !f->batches2d) || (f->n_lights && !f->lights) ||
        (f->n_occluders && !f->occluders) || (f->n_linedefs && !f->linedefs) || (f->n_chunks && !f->chunks))
        return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: NULL array with non-zero count");
    if (f->background_kind > RXR_BG_GRID) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: unknown background_kind");
    if (f->background_kind == RXR_BG_HOST_PIXELS && !f->background_pixels)
        return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: RXR_BG_HOST_PIXELS without background_pixels");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->has_frame = false;
    // Were this frame's 3D batches handed over while they were projected (rxr_stream_begin / rxr_stream_batch3d)?  Then their arrays
    // are in the blob already, or on their way.  Anything that does not match -- a batch missing, other arrays than the ones streamed,
    // a failure on the way -- and the frame takes the plain path below from scratch.
    FrameStream &S = ctx->fstream;
    bool streamed = S.active && !S.failed.load() && !(f->use_meshes & 1u) && f->n_batches3d == S.n && S.handed.load() == S.n;
    if (S.active && streamed) {
        std::lock_guard<std::mutex> lk(S.mu);
        streamed = S.next == S.n;
        for (uint32_t i = 0; i < S.n && streamed; ++i) {
            const rxr_batch3d &b = f->batches3d[i];
            const FrameStream::Rec &R = S.rec[i];
            streamed = b.projected_vertices == R.pv && b.clipped_uvs == R.uv && b.clipped_normals == R.nrm && b.clipped_indices == R.idx &&
                       (b.edges ? (const void *)b.edges : (const void *)b.edge_visible) == R.edges && (!b.n_triangles || (b.edges ? 0 : 1) == S.edgeless.load()) &&
                       b.n_vertices == R.nv && b.n_triangles == R.nt;
        }
    }
    if (S.active && !streamed) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (transfers of the abandoned stream still read the staging memory)
    S.active = false;
    static const bool e2e_timing = getenv("RXR_E2E_TIMING") != nullptr;  // diagnostics (tools/e2e_probe.py)
    const auto ut0 = std::chrono::steady_clock::now();
    auto ut_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ut0).count(); };
    double ut_validated = 0, ut_quiesced = 0, ut_headers = 0, ut_arrays = 0;

    // ---- pass 1: validate + size ---------------------------------------------------------------
    if (f->n_shader_programs > ctx->programs.size())
        return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: n_shader_programs exceeds the programs set with rxr_set_shaders");
    // batch.shader -> program: chunk.shaders.get(index) for chunk batches, scene.shaders.get(index) otherwise
    // (src/rasterizer.rs:763-766, :1285-1288, :1645-1648)
    auto program_of = [&](int32_t shader, int32_t chunk) -> uint32_t {
        if (shader < 0) return 0u;
        if (chunk >= 0) {
            const rxr_chunk &ck = f->chunks[chunk];
            return (uint32_t)shader < ck.n_programs ? ck.program_base + (uint32_t)shader + 1u : 0u;
        }
        return (uint32_t)shader < f->n_shader_programs ? (uint32_t)shader + 1u : 0u;
    };
    bool uses_programs = false, uses_chunk_tex = false;
    bool vis_programs = false;  // an opaque-pass batch whose program may write `opacity`: the visibility loop has to run it (DB_FULL_ALPHA)
    bool any_3d_visible = false, any_3d_program = false;
    int edgeless3d = -1;  // ABI 5: -1 no triangles yet, 0 the batches carry Edges records, 1 they carry `edge_visible` words instead
    // the pixel rows in which the reference can draw anything of this frame: per kept batch the rows of the reference's own tiles that pass
    // its batch box test (rxr_ref_tile_span: rasterizer.rs:978-983, :594-600) -- outside them a 3D frame is the miss colour
    uint32_t content_y0 = f->height, content_y1 = 0;
    // ... and per tile row of the frame the tile columns, from the same boxes (a frame of very many boxes gives up: span_budget)
    const uint32_t n_tile_rows = (f->height + RXR_TILE_H - 1u) / RXR_TILE_H, n_tile_cols = (f->width + RXR_TILE_W - 1u) / RXR_TILE_W;
    std::vector<uint32_t> span_lo(n_tile_rows, n_tile_cols), span_hi(n_tile_rows, 0u);
    long span_budget = 400000;  // row updates
    auto add_span = [&](uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1) {  // pixels [x0, x1) x [y0, y1)
        if (x0 >= x1 || y0 >= y1 || span_budget < 0) return;
        const uint32_t c0 = x0 / RXR_TILE_W, c1 = std::min((x1 + RXR_TILE_W - 1u) / RXR_TILE_W, n_tile_cols), r0 = y0 / RXR_TILE_H, r1 = std::min((y1 + RXR_TILE_H - 1u) / RXR_TILE_H, n_tile_rows);
        span_budget -= (long)(r1 - r0);
        for (uint32_t r = r0; r < r1 && span_budget >= 0; ++r) {
            span_lo[r] = std::min(span_lo[r], c0);
            span_hi[r] = std::max(span_hi[r], c1);
        }
    };
    // emissive (rasterizer.rs:1323, :1394): see the check behind the 3D batches below
    bool emissive_writer_3d = false;        // a visible 3D batch (either pass) runs a program that contains SetEmissive
    bool opaque_without_emissive = false;   // a visible opaque-pass 3D batch whose fragments do NOT assign emissive themselves
    uint32_t reads_2d = 0;  // PF_* read-before-written by the programs of visible 2D batches
    int32_t first_opacity_chunk = -1;  // opacity batches in two or more chunks: surface_id needs the exact prefix order (level 1)
    bool seen_profiled_opaque = false; // ... and so does an opacity batch submitted AFTER an opaque batch that carries a profile id:
                                       // that opaque batch must not see it (rasterizer.rs:314-357; found by tools/fuzz_sweep.py)
    size_t n_v3 = 0, n_t3 = 0;
    bool has_opacity = false;
    const bool use_meshes = (f->use_meshes & 1u) != 0;     // the 3D batches are the meshes of rxr_set_meshes
    const bool use_meshes2d = (f->use_meshes & 2u) != 0;   // the 2D batches are the meshes of rxr_set_meshes2d
    if (f->use_meshes > 3u) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: use_meshes has unknown bits");
    if (use_meshes && f->n_batches3d) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: use_meshes with batches3d");
    if (use_meshes2d && f->n_batches2d) return fail(ctx, RXR_ERR_INVALID, "rxr_upload_frame: use_meshes (2D) with batches2d");
    if (use_meshes2d)
        for (const rxr_ctx::HostMesh2D &h : ctx->meshes2d)
            if (h.chunk >= (int32_t)f->n_chunks) return fail(ctx, RXR_ERR_INVALID, "mesh2d: chunk index out of range");
    const uint32_t n_b3 = use_meshes ? (uint32_t)ctx->meshes.size() : f->n_batches3d;
    if (use_meshes)
        for (const HostMesh &h : ctx->meshes) {
            if (h.chunk >= (int32_t)f->n_chunks) return fail(ctx, RXR_ERR_INVALID, "mesh: chunk index out of range");
            if (h.list == RXR_LIST_CHUNK_OPACITY) has_opacity = true;
        }
    for (uint32_t i = 0; i < f->n_batches3d; ++i) {
        const rxr_batch3d &b = f->batches3d[i];
        if (b.n_triangles && (!b.clipped_indices || (!b.edges && !b.edge_visible))) return fail(ctx, RXR_ERR_INVALID, "batch3d: NULL indices/edges");
        if (b.n_triangles) {  // ABI 5: with Edges records or without, one form per frame
            const int form = b.edges ? 0 : 1;
            if (edgeless3d < 0) edgeless3d = form;
            else if (edgeless3d != form) return fail(ctx, RXR_ERR_INVALID, "batch3d: batches with and without Edges records in one frame");
            if (form == 1 && b.cull_mode > RXR_CULL_BACK) return fail(ctx, RXR_ERR_INVALID, "batch3d: bad cull mode");
        }
        if (b.n_vertices && (!b.projected_vertices || !b.clipped_uvs)) return fail(ctx, RXR_ERR_INVALID, "batch3d: NULL vertex arrays");
        if (b.chunk >= (int32_t)f->n_chunks) return fail(ctx, RXR_ERR_INVALID, "batch3d: chunk index out of range");
        if (b.list == RXR_LIST_CHUNK_OPACITY) has_opacity = true;
        n_v3 += b.n_vertices;
        n_t3 += b.n_triangles;
    }
    if (!streamed) {  // (a streamed batch was checked when it was handed over)
        // every index of every batch (the kernels trust them): one job per batch on the host worker pool (rxr_parallel.h)
        std::atomic<bool> bad{false};
        rxr_parallel::run(f->n_batches3d, n_t3, [&](size_t i) {
            const rxr_batch3d &b = f->batches3d[i];
            uint32_t worst = 0;
            for (size_t t = 0; t < (size_t)b.n_triangles * 3u; ++t) worst = std::max(worst, b.clipped_indices[t]);
            if (b.n_triangles && worst >= b.n_vertices) bad.store(true, std::memory_order_relaxed);
        });
        if (bad.load()) return fail(ctx, RXR_ERR_INVALID, "batch3d: vertex index out of range");
    }
    if (n_v3 >= (1ull << 31) || n_t3 >= (1ull << 31)) return fail(ctx, RXR_ERR_INVALID, "frame too large (>= 2^31 vertices or triangles)");
    size_t n_t2 = 0, n_l2 = 0, n_items = 0;
    for (uint32_t i = 0; i < f->n_batches2d; ++i) {
        const rxr_batch2d &b = f->batches2d[i];
        if (b.n_triangles && !b.indices) return fail(ctx, RXR_ERR_INVALID, "batch2d: NULL indices");  // (edges may be NULL since ABI 5: built below)
        if (b.n_vertices && (!b.projected_vertices || !b.uvs)) return fail(ctx, RXR_ERR_INVALID, "batch2d: NULL vertex arrays");
        if (b.chunk >= (int32_t)f->n_chunks) return fail(ctx, RXR_ERR_INVALID, "batch2d: chunk index out of range");
        if (b.mode > RXR_MODE_LINE_LOOP) return fail(ctx, RXR_ERR_INVALID, "batch2d: bad mode");
        if (b.mode == RXR_MODE_TRIANGLES || b.mode == RXR_MODE_LINES)
            for (size_t t = 0; t < (size_t)b.n_triangles * 3u; ++t) {
                if (b.mode == RXR_MODE_LINES && (t % 3u) == 2u) continue;  // only .0/.1 are read, :902
                if (b.indices[t] >= b.n_vertices) return fail(ctx, RXR_ERR_INVALID, "batch2d: vertex index out of range");
            }
        switch (b.mode) {
            case RXR_MODE_TRIANGLES: n_t2 += b.n_triangles; break;
            case RXR_MODE_LINES: n_l2 += b.n_triangles; break;
            case RXR_MODE_LINE_STRIP: n_l2 += b.n_vertices ? b.n_vertices - 1 : 0; break;
            default: n_l2 += b.n_vertices; break;
        }
        n_items += 1;
    }
    if (use_meshes2d) {
        n_t2 = ctx->meshes2d_tris;
        n_l2 = ctx->meshes2d_prims - ctx->meshes2d_tris;
    }
    const uint32_t n_b2 = use_meshes2d ? (uint32_t)ctx->meshes2d.size() : f->n_batches2d;
    size_t n_occ_total = f->n_occluders;
    // this frame's chunk textures (terrain, baked shader textures): DevTexDesc indices after the resident ones
    struct LocalTex {
        const rxr_texture *t;
        bool opaque;
    };
    std::vector<LocalTex> local_tex;
    std::vector<int32_t> chunk_terrain(f->n_chunks, -1);
    std::vector<std::vector<int32_t>> chunk_baked(f->n_chunks);
    size_t local_texels = 0;
    const uint32_t n_res_tex = (uint32_t)ctx->h_tex.size();
    for (uint32_t c = 0; c < f->n_chunks; ++c) {
        const rxr_chunk &ck = f->chunks[c];
        if (ck.n_occluders && !ck.occluders) return fail(ctx, RXR_ERR_INVALID, "chunk: NULL occluders");
        n_occ_total += ck.n_occluders;
        if ((uint64_t)ck.program_base + ck.n_programs > ctx->programs.size())
            return fail(ctx, RXR_ERR_INVALID, "chunk: program range exceeds the programs set with rxr_set_shaders");
        if (ck.n_shader_textures && !ck.shader_textures) return fail(ctx, RXR_ERR_INVALID, "chunk: NULL shader_textures");
        auto add_local = [&](const rxr_texture &t) -> int32_t {
            if (t.width == 0 || t.height == 0 || t.width > 32768 || t.height > 32768) return -2;
            bool opaque = true;
            const size_t n = (size_t)t.width * t.height;
            for (size_t k = 0; k < n && opaque; ++k) opaque = t.rgba[4 * k + 3] == 255;
            local_tex.push_back(LocalTex{&t, opaque});
            local_texels += n;
            return (int32_t)(n_res_tex + local_tex.size() - 1);
        };
        if (ck.terrain_texture && ck.terrain_texture->rgba) {
            if (ck.size == 0) return fail(ctx, RXR_ERR_INVALID, "chunk: size 0 with a terrain texture (the reference divides by it, chunk.rs:138)");
            if ((chunk_terrain[c] = add_local(*ck.terrain_texture)) == -2) return fail(ctx, RXR_ERR_INVALID, "chunk: bad terrain texture size");
        }
        chunk_baked[c].assign(ck.n_shader_textures, -1);
        for (uint32_t k = 0; k < ck.n_shader_textures; ++k)
            if (ck.shader_textures[k].rgba && (chunk_baked[c][k] = add_local(ck.shader_textures[k])) == -2)
                return fail(ctx, RXR_ERR_INVALID, "chunk: bad shader texture size");
    }
    if (local_texels >= (1ull << 31)) return fail(ctx, RXR_ERR_INVALID, "chunk textures too large");
    auto tex_opaque = [&](int32_t tex) { return (uint32_t)tex < n_res_tex ? ctx->h_tex[tex].all_opaque != 0 : local_tex[(uint32_t)tex - n_res_tex].opaque; };

    Layout L{};
    // (a streamed frame's pools were laid out for the capacities announced before projection: the arrays sit densely at their front)
    size_t o = streamed ? layout_prefix(n_b3, S.total_cap_v, S.total_cap_t, L) : layout_prefix(n_b3, n_v3, n_t3, L);
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + (bytes ? bytes : 16), 256);
        return at;
    };
    const bool with_tri_info = !use_meshes && n_t3 > 0 && n_t3 <= RXR_TRI_INFO_MAX;
    L.off_tinfo = take(with_tri_info ? n_t3 * sizeof(uint2) : 0);
    // host-projected 3D batches whose bounding-box arithmetic the reference's per-tile test cannot be trusted with (rxr_device.h,
    // rxr_ref_tile_span): per batch the pixels the reference draws it in
    bool any_risky3d = false;
    if (!(use_meshes & 1u))
        for (uint32_t i = 0; i < n_b3 && !any_risky3d; ++i) {
            const rxr_batch3d &b = f->batches3d[i];
            any_risky3d = b.has_bounding_box && b.n_triangles > 0 && rxr_box_is_risky(b.bounding_box[0], b.bounding_box[1], b.bounding_box[2], b.bounding_box[3]);
        }
    L.off_clip3d = take(any_risky3d ? n_b3 * sizeof(uint4) : 0);
    L.off_lights = take(f->n_lights * sizeof(rxr_light));
    L.off_lights_fast = take(f->n_lights * sizeof(LightFast));
    L.off_occ = take(n_occ_total * sizeof(rxr_occluder));
    L.off_ld = take(f->n_linedefs * sizeof(rxr_linedef));
    L.off_chunks = take(f->n_chunks * sizeof(ChunkRange));
    L.off_tdesc = take((n_res_tex + local_tex.size()) * sizeof(DevTexDesc));
    L.off_ltex = take(local_texels * 4);
    L.off_b2 = take(n_b2 * sizeof(DevBatch));
    L.off_p2 = take((n_t2 + n_l2) * sizeof(Prim2D));
    (void)n_items;
    L.off_bg = take(f->background_kind == RXR_BG_HOST_PIXELS ? (size_t)f->width * f->height * 4 : 0);
    L.total = o;

    int rc;
    if (streamed && L.total > S.blob_capacity) {
        // what follows the arrays does not fit the room rxr_stream_begin left: a reallocation would lose what has been shipped.  The plain
        // path from scratch (rare: the frame's lights / 2D primitives / chunk textures more than doubled against the previous frame).
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->last_blob_tail = L.total - L.off_tinfo;
        S.failed.store(1);
        return rxr_upload_frame(ctx, f);
    }
    ctx->last_blob_tail = L.total - L.off_tinfo;
    if ((rc = ensure_stage(ctx, L.total)) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_frame, L.total)) != RXR_OK) return rc;
    // the staging blob may still be in flight from the previous upload, and the previous frame's renders -- possibly on
    // the caller's stream -- still read the frame blob and the scratch buffers
    ut_validated = ut_ms();
    // (a streamed frame: rxr_stream_begin has waited for the previous frame; what runs now are this frame's own transfers)
    if (!streamed && (rc = rxr_quiesce(ctx)) != RXR_OK) return rc;
    ut_quiesced = ut_ms();
    uint8_t *st = (uint8_t *)ctx->h_stage;

    // texel source, program, baked texture and the flags that follow from them, for one 3D batch header
    // (src/rasterizer.rs:1101-1304).  `keep` comes in as the box test and goes out false when nothing of the
    // batch can ever be written.
    auto classify3d = [&](DevBatch &d, const rxr_source &source, int32_t chunk, int32_t shader, uint32_t list, bool &keep,
                          const char *what) -> int {
        const bool opacity_list = list == RXR_LIST_CHUNK_OPACITY;
        d.program_plus1 = program_of(shader, chunk);
        d.baked_plus1 = 0;
        // chunk.shader_textures.get(shader_index) comes first, and only in the opaque pass (:1226-1267)
        if (!opacity_list && shader >= 0 && chunk >= 0 && (size_t)shader < chunk_baked[chunk].size() && chunk_baked[chunk][shader] >= 0) {
            d.baked_plus1 = (uint32_t)chunk_baked[chunk][shader] + 1u;
            d.program_plus1 = 0;
        }
        bool prog_runs = d.program_plus1 && ctx->programs[d.program_plus1 - 1].shade_entry != 0xFFFFFFFFu;
        const bool prog_opacity = prog_runs && (ctx->programs[d.program_plus1 - 1].flags & PG_WRITES_OPACITY);
        if (keep && prog_runs && (ctx->programs[d.program_plus1 - 1].flags & PG_WRITES_EMISSIVE)) emissive_writer_3d = true;
        if (keep && !opacity_list && !(prog_runs && (ctx->programs[d.program_plus1 - 1].flags & PG_ASSIGNS_EMISSIVE))) opaque_without_emissive = true;
        if (!prog_runs) d.program_plus1 = 0;
        if (prog_runs) d.flags |= DB_HAS_PROGRAM;
        if (!keep) return RXR_OK;
        bool alpha_varies = false, alpha_never_255 = false;
        if (source.kind == RXR_SOURCE_TERRAIN && chunk >= 0) {  // chunk.sample_terrain_texture, :1189-1191
            if (chunk_terrain[chunk] >= 0) {
                d.tex = chunk_terrain[chunk];
                d.flags |= DB_TERRAIN;
                alpha_varies = !tex_opaque(d.tex);
            } else {
                d.tex = -1;
                d.pixel = 0;  // no terrain texture: [0,0,0,0], chunk.rs:150
                alpha_never_255 = true;
            }
        } else {
            int rc2 = resolve_source(ctx, source, true, chunk, f->animation_frame, d.tex, d.pixel);
            if (rc2 != RXR_OK) return fail(ctx, rc2, std::string(what) + ": texture tile index out of range or tile without textures (the reference panics)");
            if (d.tex >= 0) alpha_varies = !tex_opaque(d.tex);
            else alpha_never_255 = (d.pixel >> 24) != 255u;
        }
        if (d.baked_plus1) {  // the baked texel replaces colour AND alpha (:1253-1262)
            alpha_varies = !tex_opaque((int32_t)d.baked_plus1 - 1);
            alpha_never_255 = false;
        }
        if (opacity_list) return RXR_OK;  // the opacity pass writes unconditionally (:1678-1682)
        if (prog_opacity) {
            d.flags |= DB_FULL_ALPHA;
            vis_programs = true;
        } else if (alpha_never_255) {
            keep = false;  // encoded alpha != 255: never written (:1408)
        } else if (alpha_varies) {
            // a plain texture's alpha needs the uv only; terrain texels need the world position, baked textures replace the source
            d.flags |= ((d.flags & DB_TERRAIN) || d.baked_plus1) ? DB_FULL_ALPHA : DB_ALPHA_TEST;
        }
        return RXR_OK;
    };

    // Opacity groups (rxr_kernels.hip front_insert): the surface_id staircase of a pixel keeps one entry per GROUP of opacity
    // batches -- a maximal run, in submission order, with no opaque batch that carries a profile id in between.  Only such
    // batches ever read surface_id (:1044-1048), and their triangles' indices lie outside every group's index range, so of a
    // group's prefix minima only the last can be what they see.  (Counting a batch that turns out to be skipped as a separator
    // only makes groups smaller, which is always safe.)
    uint32_t opacity_group = 0;
    auto opacity_group_of = [&](DevBatch &d, bool opacity_list, bool has_profile) -> int {
        if (opacity_list) {
            if (opacity_group >= 0xFFFFu) return fail(ctx, RXR_ERR_UNSUPPORTED, "more than 65534 groups of opacity batches");
            d.flags |= opacity_group << DB_GROUP_SHIFT;
        } else if (has_profile) {
            ++opacity_group;
        }
        return RXR_OK;
    };

    // ---- pass 2: flatten -----------------------------------------------------------------------
    const float W = (float)f->width, H = (float)f->height;
    DevBatch *b3 = (DevBatch *)(st + L.off_b3);
    uint32_t *base = (uint32_t *)(st + L.off_base);
    size_t vcur = 0, tcur = 0;
    for (uint32_t i = 0; i < f->n_batches3d; ++i) {
        const rxr_batch3d &b = f->batches3d[i];
        DevBatch d{};
        d.vert_base = (uint32_t)vcur;
        d.tri_base = (uint32_t)tcur;
        d.n_tris = b.n_triangles;
        d.n_verts = b.n_vertices;
        d.flags = 0;
        if (b.clipped_normals) d.flags |= DB_HAS_NORMALS;
        if (b.has_profile_id) d.flags |= DB_HAS_PROFILE;
        if (b.list == RXR_LIST_CHUNK_OPACITY) d.flags |= DB_OPACITY_LIST;
        d.profile_id = b.profile_id;
        d.repeat_mode = b.repeat_mode;
        d.chunk = b.chunk;
        d.mode = edgeless3d == 1 ? b.cull_mode : 0u;  // (3D batches without Edges records: the cull mode make_setup builds them under)
        if ((rc = opacity_group_of(d, b.list == RXR_LIST_CHUNK_OPACITY, b.has_profile_id != 0)) != RXR_OK) return rc;
        memcpy(d.ambient, b.ambient_color, 12);
        // batch-level box reject, rasterizer.rs:978-983, evaluated against the whole screen
        bool keep = b.has_bounding_box && b.n_triangles > 0;
        if (keep) {
            const float *bb = b.bounding_box;
            keep = bb[0] < W && (bb[0] + bb[2]) > 0.0f && bb[1] < H && (bb[1] + bb[3]) > 0.0f;
        }
        if ((rc = classify3d(d, b.source, b.chunk, b.shader, b.list, keep, "batch3d")) != RXR_OK) return rc;
        if (!keep) d.flags |= DB_SKIP;
        else {
            any_3d_visible = true;
            {
                uint32_t y0 = 0, y1 = 0;
                rxr_ref_tile_span(b.bounding_box[1], b.bounding_box[3], f->height, f->tile_size, 0.0f, y0, y1);
                if (y0 < y1) {
                    content_y0 = std::min(content_y0, y0);
                    content_y1 = std::max(content_y1, y1);
                    uint32_t x0 = 0, x1 = 0;
                    rxr_ref_tile_span(b.bounding_box[0], b.bounding_box[2], f->width, f->tile_size, 0.0f, x0, x1);
                    add_span(x0, x1, y0, y1);
                }
            }
            if (d.program_plus1) uses_programs = any_3d_program = true;
            if ((d.flags & (DB_TERRAIN | DB_FULL_ALPHA)) || d.baked_plus1) uses_chunk_tex = true;
            if (b.list == RXR_LIST_CHUNK_OPACITY) {
                if (first_opacity_chunk < 0) first_opacity_chunk = b.chunk;
                else if (first_opacity_chunk != b.chunk) uses_chunk_tex = true;
                if (seen_profiled_opaque) uses_chunk_tex = true;
            } else if (d.flags & DB_HAS_PROFILE) {
                seen_profiled_opaque = true;
            }
        }
        b3[i] = d;
        base[i] = (uint32_t)tcur;
        vcur += b.n_vertices;
        tcur += b.n_triangles;
    }
    base[f->n_batches3d] = (uint32_t)tcur;
    if (with_tri_info) {  // (batch, vert_base) per triangle: k_setup3d of a small frame looks its batch up instead of searching for it
        uint2 *ti = (uint2 *)(st + L.off_tinfo);
        for (uint32_t i = 0; i < f->n_batches3d; ++i)
            for (uint32_t t = 0; t < f->batches3d[i].n_triangles; ++t) ti[b3[i].tri_base + t] = make_uint2(i, b3[i].vert_base);
    }
    if (any_risky3d) {
        uint4 *clip = (uint4 *)(st + L.off_clip3d);
        for (uint32_t i = 0; i < n_b3; ++i) {
            const rxr_batch3d &b = f->batches3d[i];
            uint4 c = make_uint4(0u, f->width, 0u, f->height);
            if (b.has_bounding_box && rxr_box_is_risky(b.bounding_box[0], b.bounding_box[1], b.bounding_box[2], b.bounding_box[3])) {
                rxr_ref_tile_span(b.bounding_box[0], b.bounding_box[2], f->width, f->tile_size, 0.0f, c.x, c.y);
                rxr_ref_tile_span(b.bounding_box[1], b.bounding_box[3], f->height, f->tile_size, 0.0f, c.z, c.w);
            }
            clip[i] = c;
        }
    }
    // the arrays themselves: independent per batch (offsets are in the headers just written), through the host worker pool --
    // 124 MB for the 1 M-triangle grid, which one thread copies in about as long as the GPU takes for forty frames
    ut_headers = ut_ms();
    // Large frames (the 1 M-triangle grid: 124 MB): the copy into pinned memory and the host->device copy are pipelined.  The
    // batches are cut into groups of consecutive batches of ~8 MB; the workers copy batch after batch, and the calling thread ships
    // the five array ranges of a group (vertices, uvs, normals, indices, edges: contiguous per group in each pool) as soon as the
    // group's last batch has landed -- the PCIe transfer of group g runs under the copies of the groups behind it.  Measured
    // (profiles/r03/c5_e2e_breakdown.jsonl): the whole call 14.1 -> 10.6 ms together with the wider worker pool (the hand-over alone 2.6 ms
    // either way on 64 threads: what the pipeline hides is the transfer, 124 MB at 50 GB/s).  RXR_UPLOAD_PIPELINE=0 keeps one copy at the end.
    struct ShipGroup {
        size_t v0, v1, t0, t1;
        std::atomic<uint32_t> left{0};
    };
    const size_t big_bytes = n_v3 * 36 + n_t3 * 52;
    static const bool pipeline_off = getenv("RXR_UPLOAD_PIPELINE") && getenv("RXR_UPLOAD_PIPELINE")[0] == '0';
    std::vector<uint32_t> group_of;
    std::unique_ptr<ShipGroup[]> groups;
    size_t n_groups_ship = 0;
    if (!streamed && !use_meshes && !pipeline_off && big_bytes >= (32u << 20) && f->n_batches3d >= 8) {
        size_t target = std::max<size_t>(8u << 20, big_bytes / 8);  // (few, large transfers: every hipMemcpyAsync costs the calling thread ~20 us)
        if (const char *gs = getenv("RXR_UPLOAD_GROUP_MB")) target = std::max<size_t>(1u << 20, (size_t)atol(gs) << 20);
        group_of.resize(f->n_batches3d);
        groups.reset(new ShipGroup[f->n_batches3d]);
        size_t acc = 0;
        for (uint32_t i = 0; i < f->n_batches3d; ++i) {
            if (i == 0 || acc >= target) {
                ShipGroup &g = groups[n_groups_ship++];
                g.v0 = g.v1 = b3[i].vert_base;
                g.t0 = g.t1 = b3[i].tri_base;
                acc = 0;
            }
            ShipGroup &g = groups[n_groups_ship - 1];
            g.v1 += f->batches3d[i].n_vertices;
            g.t1 += f->batches3d[i].n_triangles;
            g.left.store(g.left.load(std::memory_order_relaxed) + 1u, std::memory_order_relaxed);
            group_of[i] = (uint32_t)(n_groups_ship - 1);
            acc += (size_t)f->batches3d[i].n_vertices * 36 + (size_t)f->batches3d[i].n_triangles * 52;
        }
    }
    auto copy_batch = [&](size_t i) {
        const rxr_batch3d &b = f->batches3d[i];
        const size_t v0 = b3[i].vert_base, t0 = b3[i].tri_base;
        if (b.n_vertices) {
            memcpy(st + L.off_pv + v0 * 16, b.projected_vertices, (size_t)b.n_vertices * 16);
            memcpy(st + L.off_uv + v0 * 8, b.clipped_uvs, (size_t)b.n_vertices * 8);
            if (b.clipped_normals) memcpy(st + L.off_nrm + v0 * 12, b.clipped_normals, (size_t)b.n_vertices * 12);
        }
        if (b.n_triangles) {
            memcpy(st + L.off_idx + t0 * 12, b.clipped_indices, (size_t)b.n_triangles * 12);
            if (edgeless3d == 1) memcpy(st + L.off_edges + t0 * sizeof(uint32_t), b.edge_visible, (size_t)b.n_triangles * sizeof(uint32_t));
            else memcpy(st + L.off_edges + t0 * sizeof(rxr_edges), b.edges, (size_t)b.n_triangles * sizeof(rxr_edges));
        }
        if (n_groups_ship) groups[group_of[i]].left.fetch_sub(1u, std::memory_order_release);
    };
    bool arrays_shipped = false;  // the five pools are on their way: the final copy sends only what surrounds them
    ctx->last_upload_streamed = streamed ? (S.pinned ? 2 : 1) : 0;
    if (streamed) {
        arrays_shipped = true;  // ... since rxr_stream_batch3d
        if (!S.pinned && S.copy_next.load() < S.n) {
            // batches retired by the last hand-over calls and not copied yet: through the pool, one claim per job
            std::atomic<bool> ok{true};
            rxr_parallel::run(S.n - S.copy_next.load(), n_v3 + n_t3, [&](size_t) {
                if (!rxr_stream_copy_some(ctx, 1u)) ok.store(false);
            });
            if (!ok.load() || S.failed.load()) return fail(ctx, RXR_ERR_HIP, "rxr_upload_frame: " + (S.err.empty() ? std::string("a streamed batch could not be shipped") : S.err));
        }
    } else if (n_groups_ship) {
        hipError_t ship_err = hipSuccess;
        uint8_t *const dst = (uint8_t *)ctx->d_frame.p;
        auto ship = [&]() {
            for (size_t g = 0; g < n_groups_ship; ++g) {
                ShipGroup &G = groups[g];
                while (G.left.load(std::memory_order_acquire) != 0u) std::this_thread::yield();
                const size_t nv = G.v1 - G.v0, nt = G.t1 - G.t0, tri5 = edgeless3d == 1 ? sizeof(uint32_t) : sizeof(rxr_edges);
                const struct { size_t off, bytes; } r[5] = {{L.off_pv + G.v0 * 16, nv * 16}, {L.off_uv + G.v0 * 8, nv * 8}, {L.off_nrm + G.v0 * 12, nv * 12},
                                                            {L.off_idx + G.t0 * 12, nt * 12}, {L.off_edges + G.t0 * tri5, nt * tri5}};
                for (const auto &x : r)
                    if (x.bytes && ship_err == hipSuccess) ship_err = hipMemcpyAsync(dst + x.off, st + x.off, x.bytes, hipMemcpyHostToDevice, ctx->stream);
            }
        };
        arrays_shipped = rxr_parallel::run_with(f->n_batches3d, n_v3 + n_t3, copy_batch, ship);
        if (!arrays_shipped) {  // (a pool of one thread: copy, then one transfer at the end as for small frames)
            n_groups_ship = 0;
            for (size_t i = 0; i < f->n_batches3d; ++i) copy_batch(i);
        }
        if (ship_err != hipSuccess) return fail(ctx, RXR_ERR_HIP, std::string("rxr_upload_frame: hipMemcpyAsync: ") + hipGetErrorString(ship_err));
    } else {
        rxr_parallel::run(f->n_batches3d, n_v3 + n_t3, copy_batch);
    }

    ut_arrays = ut_ms();
    if (use_meshes) {
        // headers of the device-projected batches + the per-frame half of DevMesh (view * model, frustum reject)
        rvek::Mat4 view{}, proj{};
        memcpy(view.m, f->view, 64);
        memcpy(proj.m, f->projection, 64);
        const rvek::Mat4 pv_m = proj * view;
        std::vector<DevMesh> dm(ctx->meshes.size());
        for (uint32_t i = 0; i < n_b3; ++i) {
            const HostMesh &h = ctx->meshes[i];
            rvek::Mat4 model{};
            memcpy(model.m, f->mesh_transforms ? f->mesh_transforms + 16 * (size_t)i : h.transform, 64);
            const rvek::Mat4 mvp = pv_m * model;             // batch3d.rs:490
            const rvek::Mat4 view_model = view * model;      // :555
            bool rejected = false;
            if (h.has_vertices) {                            // :493-552
                bool out_l = true, out_r = true, out_b = true, out_t = true, out_n = true, out_f = true;
                for (int c = 0; c < 8; ++c) {
                    rvek::Vec4 corner{(c & 4) ? h.aabb_hi[0] : h.aabb_lo[0], (c & 2) ? h.aabb_hi[1] : h.aabb_lo[1],
                                      (c & 1) ? h.aabb_hi[2] : h.aabb_lo[2], 1.0f};
                    rvek::Vec4 v = mvp * corner;
                    const float w = v.w;
                    out_l &= v.x < -w;
                    out_r &= v.x > w;
                    out_b &= v.y < -w;
                    out_t &= v.y > w;
                    out_n &= v.z < -w;
                    out_f &= v.z > w;
                }
                rejected = out_l || out_r || out_b || out_t || out_n || out_f;
            }
            dm[i] = h.dev;
            dm[i].rejected = rejected ? 1u : 0u;
            memcpy(dm[i].view_model, view_model.m, 64);

            DevBatch d{};
            d.vert_base = h.dev.vout_base;
            d.tri_base = h.dev.tout_base;
            d.n_tris = 3u * h.dev.n_tris;
            d.n_verts = h.dev.n_verts + 4u * h.dev.n_tris;
            d.flags = DB_HAS_NORMALS;  // meshes with triangles must carry normals (batch3d.rs:605)
            if (h.has_profile_id) d.flags |= DB_HAS_PROFILE;
            if (h.list == RXR_LIST_CHUNK_OPACITY) d.flags |= DB_OPACITY_LIST;
            d.profile_id = h.profile_id;
            d.repeat_mode = h.repeat_mode;
            d.chunk = h.chunk;
            if ((rc = opacity_group_of(d, h.list == RXR_LIST_CHUNK_OPACITY, h.has_profile_id)) != RXR_OK) return rc;
            memcpy(d.ambient, h.ambient, 12);
            bool keep = !rejected && h.dev.n_tris > 0;       // the box reject itself happens on the device (dev_bbox)
            if ((rc = classify3d(d, h.source, h.chunk, h.shader, h.list, keep, "mesh")) != RXR_OK) return rc;
            if (!keep) d.flags |= DB_SKIP;
            else {
                any_3d_visible = true;
                if (d.program_plus1) uses_programs = any_3d_program = true;
                if ((d.flags & (DB_TERRAIN | DB_FULL_ALPHA)) || d.baked_plus1) uses_chunk_tex = true;
                if (h.list == RXR_LIST_CHUNK_OPACITY) {
                    if (first_opacity_chunk < 0) first_opacity_chunk = h.chunk;
                    else if (first_opacity_chunk != h.chunk) uses_chunk_tex = true;
                    if (seen_profiled_opaque) uses_chunk_tex = true;
                } else if (d.flags & DB_HAS_PROFILE) {
                    seen_profiled_opaque = true;
                }
            }
            b3[i] = d;
            base[i] = h.dev.tout_base;
        }
        base[n_b3] = (uint32_t)ctx->mesh_tris_out;
        n_t3 = ctx->mesh_tris_out;
        // the per-frame DevMesh array goes straight into the projection scratch (tiny: 96 B per mesh)
        if (n_b3) HIPCHK(ctx, hipMemcpyAsync((uint8_t *)ctx->d_proj_misc.p + ctx->pp_off_meshes, dm.data(), dm.size() * sizeof(DevMesh),
                                             hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // `dm` is a stack-lifetime source
    }

    if (f->n_lights) memcpy(st + L.off_lights, f->lights, f->n_lights * sizeof(rxr_light));
    for (uint32_t i = 0; i < f->n_lights; ++i) light_fast_record(f->lights[i], f->hash_anim, *reinterpret_cast<LightFast *>(st + L.off_lights_fast + i * sizeof(LightFast)));
    {
        rxr_occluder *oc = (rxr_occluder *)(st + L.off_occ);
        if (f->n_occluders) memcpy(oc, f->occluders, f->n_occluders * sizeof(rxr_occluder));
        ChunkRange *cr = (ChunkRange *)(st + L.off_chunks);
        size_t cur = f->n_occluders;
        for (uint32_t c = 0; c < f->n_chunks; ++c) {
            cr[c] = ChunkRange{};
            cr[c].occ_first = (uint32_t)cur;
            cr[c].occ_count = f->chunks[c].n_occluders;
            cr[c].terrain_tex = chunk_terrain[c];
            cr[c].origin_x = f->chunks[c].origin[0];
            cr[c].origin_y = f->chunks[c].origin[1];
            if (chunk_terrain[c] >= 0) {
                const int64_t wdt = (int32_t)f->chunks[c].terrain_texture->width;
                cr[c].pixels_per_tile = (int32_t)(wdt / f->chunks[c].size);  // `texture.width as i32 / self.size`, chunk.rs:138
            }
            if (f->chunks[c].n_occluders) memcpy(oc + cur, f->chunks[c].occluders, f->chunks[c].n_occluders * sizeof(rxr_occluder));
            cur += f->chunks[c].n_occluders;
        }
    }
    if (f->n_linedefs) memcpy(st + L.off_ld, f->linedefs, f->n_linedefs * sizeof(rxr_linedef));
    {
        // texture descriptor table of the frame: the resident textures, then this frame's chunk textures
        DevTexDesc *td = (DevTexDesc *)(st + L.off_tdesc);
        if (n_res_tex) memcpy(td, ctx->h_tex.data(), n_res_tex * sizeof(DevTexDesc));
        size_t cur = 0;
        for (size_t k = 0; k < local_tex.size(); ++k) {
            const rxr_texture &t = *local_tex[k].t;
            DevTexDesc &e = td[n_res_tex + k];
            e.offset = (uint32_t)cur;
            e.w = t.width;
            e.h = t.height;
            e.all_opaque = (local_tex[k].opaque ? 1u : 0u) | 2u;
            memcpy(st + L.off_ltex + cur * 4, t.rgba, (size_t)t.width * t.height * 4);
            cur += (size_t)t.width * t.height;
        }
    }

    DevBatch *b2 = (DevBatch *)(st + L.off_b2);
    Prim2D *p2 = (Prim2D *)(st + L.off_p2);
    size_t p2cur = 0, t2cur = 0;
    // `x as usize` after the clamp against the screen, as the device's sat_index (rasterizer.rs:631-634)
    auto sat_px = [](float x, uint32_t hi) -> uint32_t {
        if (!(x > 0.0f)) return 0u;
        if (x >= (float)hi) return hi;
        return (uint32_t)x;
    };
    if (use_meshes2d) {
        // device-projected 2D batches: the headers only -- the box reject, the Edges and the Prim2D records are the device's
        // (k_proj2d_*, rxr_project.hip); a batch that turns out to be off screen gets empty pixel boxes there
        for (uint32_t i = 0; i < n_b2; ++i) {
            const rxr_ctx::HostMesh2D &h = ctx->meshes2d[i];
            DevBatch d{};
            d.n_tris = h.n_tris;
            d.n_verts = h.n_verts;
            d.mode = h.mode;
            d.repeat_mode = h.repeat_mode;
            d.chunk = h.chunk;
            d.flags = h.receives_light ? DB_RECEIVES_LIGHT : 0u;
            d.program_plus1 = program_of(h.shader, h.chunk);
            if (d.program_plus1 && ctx->programs[d.program_plus1 - 1].shade_entry == 0xFFFFFFFFu) d.program_plus1 = 0;
            if (d.program_plus1) {
                d.flags |= DB_HAS_PROGRAM;
                uses_programs = true;
                reads_2d |= ctx->program_field_reads[d.program_plus1 - 1];
            }
            if (h.source.kind == RXR_SOURCE_TERRAIN && h.chunk >= 0 && chunk_terrain[h.chunk] >= 0) {  // :749-751
                d.tex = chunk_terrain[h.chunk];
                d.flags |= DB_TERRAIN;
                uses_chunk_tex = true;
            } else {
                rc = resolve_source(ctx, h.source, false, h.chunk, f->animation_frame, d.tex, d.pixel);
                if (rc != RXR_OK) return fail(ctx, rc, "mesh2d: tile without textures (the reference panics when the batch is on screen)");
            }
            b2[i] = d;
        }
        p2cur = ctx->meshes2d_prims;
        t2cur = ctx->meshes2d_tris;
    }
    for (uint32_t i = 0; i < f->n_batches2d; ++i) {
        const rxr_batch2d &b = f->batches2d[i];
        DevBatch d{};
        d.n_tris = b.n_triangles;
        d.n_verts = b.n_vertices;
        d.mode = b.mode;
        d.repeat_mode = b.repeat_mode;
        d.chunk = b.chunk;
        d.flags = b.receives_light ? DB_RECEIVES_LIGHT : 0u;
        d.program_plus1 = program_of(b.shader, b.chunk);
        if (d.program_plus1 && ctx->programs[d.program_plus1 - 1].shade_entry == 0xFFFFFFFFu) d.program_plus1 = 0;
        if (d.program_plus1) {
            d.flags |= DB_HAS_PROGRAM;
            uses_programs = true;
            reads_2d |= ctx->program_field_reads[d.program_plus1 - 1];
        }
        // batch-level box reject with pad 0.5, rasterizer.rs:594-600, against the whole screen
        bool keep = b.has_bounding_box != 0;
        if (keep) {
            const float *bb = b.bounding_box;
            const float pad = 0.5f;
            keep = bb[0] < W + pad && (bb[0] + bb[2]) > 0.0f - pad && bb[1] < H + pad && (bb[1] + bb[3]) > 0.0f - pad;
        }
        if (keep) {
            if (b.source.kind == RXR_SOURCE_TERRAIN && b.chunk >= 0 && chunk_terrain[b.chunk] >= 0) {  // :749-751
                d.tex = chunk_terrain[b.chunk];
                d.flags |= DB_TERRAIN;
                uses_chunk_tex = true;
            } else {
                rc = resolve_source(ctx, b.source, false, b.chunk, f->animation_frame, d.tex, d.pixel);
                if (rc != RXR_OK) return fail(ctx, rc, "batch2d: tile without textures (the reference panics)");
            }
        } else {
            d.flags |= DB_SKIP;
        }
        b2[i] = d;
        if (!keep) continue;
        // a batch whose box arithmetic the reference's per-tile test cannot be trusted with: its primitives only count inside the tiles
        // that pass it (pad 0.5, :594-600)
        uint32_t clip_x0 = 0, clip_x1 = f->width, clip_y0 = 0, clip_y1 = f->height;
        const bool risky2d = rxr_box_is_risky(b.bounding_box[0], b.bounding_box[1], b.bounding_box[2], b.bounding_box[3]);
        if (risky2d) {
            rxr_ref_tile_span(b.bounding_box[0], b.bounding_box[2], f->width, f->tile_size, 0.5f, clip_x0, clip_x1);
            rxr_ref_tile_span(b.bounding_box[1], b.bounding_box[3], f->height, f->tile_size, 0.5f, clip_y0, clip_y1);
        }
        const size_t p2_first = p2cur;
        if (b.mode == RXR_MODE_TRIANGLES) {
            for (uint32_t t = 0; t < b.n_triangles; ++t) {
                const uint32_t *ix = b.indices + 3 * (size_t)t;
                rxr_edges built;
                if (!b.edges) {
                    // ABI 5: Batch2D::project's Edges::new([v0,v1,v2], [v1,v2,v0], true) (src/batch/batch2d.rs:413-424, src/edge.rs:12-24)
                    // from the projected vertices, here on the host (this file is built -ffp-contract=off: the same floats)
                    const float *v[3] = {b.projected_vertices + 2 * (size_t)ix[0], b.projected_vertices + 2 * (size_t)ix[1], b.projected_vertices + 2 * (size_t)ix[2]};
                    for (int k = 0; k < 3; ++k) {
                        const float *p = v[k], *q = v[(k + 1) % 3];
                        built.a[k] = q[1] - p[1];
                        built.b[k] = p[0] - q[0];
                        built.c[k] = q[0] * p[1] - q[1] * p[0];
                    }
                    built.visible = 1u;
                }
                const rxr_edges &e = b.edges ? b.edges[t] : built;
                Prim2D T{};
                memcpy(T.ea, e.a, 12);
                memcpy(T.eb, e.b, 12);
                memcpy(T.ec, e.c, 12);
                T.v0x = b.projected_vertices[2 * ix[0]]; T.v0y = b.projected_vertices[2 * ix[0] + 1];
                T.v1x = b.projected_vertices[2 * ix[1]]; T.v1y = b.projected_vertices[2 * ix[1] + 1];
                T.v2x = b.projected_vertices[2 * ix[2]]; T.v2y = b.projected_vertices[2 * ix[2] + 1];
                T.u0 = b.uvs[2 * ix[0]]; T.v0 = b.uvs[2 * ix[0] + 1];
                T.u1 = b.uvs[2 * ix[1]]; T.v1 = b.uvs[2 * ix[1] + 1];
                T.u2 = b.uvs[2 * ix[2]]; T.v2 = b.uvs[2 * ix[2] + 1];
                T.batch_kind = (i << 2) | (e.visible ? 1u : 0u);
                // clamped pixel box of the triangle, rasterizer.rs:615-634 with tile = whole screen
                float min_xf = std::fmin(T.v0x, std::fmin(T.v1x, T.v2x)), max_xf = std::fmax(T.v0x, std::fmax(T.v1x, T.v2x));
                float min_yf = std::fmin(T.v0y, std::fmin(T.v1y, T.v2y)), max_yf = std::fmax(T.v0y, std::fmax(T.v1y, T.v2y));
                uint32_t min_x = sat_px(std::fmax(std::floor(min_xf), 0.0f), 0xFFFFu), max_x = sat_px(std::fmin(std::ceil(max_xf), W), 0xFFFFu);
                uint32_t min_y = sat_px(std::fmax(std::floor(min_yf), 0.0f), 0xFFFFu), max_y = sat_px(std::fmin(std::ceil(max_yf), H), 0xFFFFu);
                if (!(min_x < max_x && min_y < max_y) || !e.visible) min_x = max_x = min_y = max_y = 0;
                T.bx = min_x | (max_x << 16);
                T.by = min_y | (max_y << 16);
                p2[p2cur++] = T;
                ++t2cur;
            }
        } else {
            const uint8_t white[4] = {255, 255, 255, 255};
            uint32_t color = pack_px(b.source.kind == RXR_SOURCE_PIXEL ? b.source.pixel : white);  // :911-915
            auto push = [&](uint32_t ia, uint32_t ib) -> bool {
                int32_t x0, y0, x1, y1;
                if (!to_isize32(b.projected_vertices[2 * ia], x0) || !to_isize32(b.projected_vertices[2 * ia + 1], y0) ||
                    !to_isize32(b.projected_vertices[2 * ib], x1) || !to_isize32(b.projected_vertices[2 * ib + 1], y1))
                    return false;
                Prim2D Ln{};
                memcpy(&Ln.v0x, &x0, 4);
                memcpy(&Ln.v0y, &y0, 4);
                memcpy(&Ln.v1x, &x1, 4);
                memcpy(&Ln.v1y, &y1, 4);
                memcpy(&Ln.v2x, &color, 4);
                Ln.batch_kind = (i << 2) | 2u | 1u;
                // the walk never leaves the end-point box (the last point is not plotted, :1800)
                int64_t lx0 = std::min(x0, x1), lx1 = (int64_t)std::max(x0, x1) + 1, ly0 = std::min(y0, y1), ly1 = (int64_t)std::max(y0, y1) + 1;
                auto cl = [](int64_t v, int64_t hi) { return (uint32_t)std::min<int64_t>(std::max<int64_t>(v, 0), hi); };
                uint32_t min_x = cl(lx0, (int64_t)f->width), max_x = cl(lx1, (int64_t)f->width);
                uint32_t min_y = cl(ly0, (int64_t)f->height), max_y = cl(ly1, (int64_t)f->height);
                if (!(min_x < max_x && min_y < max_y)) min_x = max_x = min_y = max_y = 0;
                Ln.bx = min_x | (max_x << 16);
                Ln.by = min_y | (max_y << 16);
                p2[p2cur++] = Ln;
                return true;
            };
            bool ok = true;
            if (b.mode == RXR_MODE_LINES) {
                for (uint32_t t = 0; t < b.n_triangles && ok; ++t) ok = push(b.indices[3 * (size_t)t], b.indices[3 * (size_t)t + 1]);
            } else if (b.mode == RXR_MODE_LINE_STRIP) {
                for (uint32_t k = 0; k + 1 < b.n_vertices && ok; ++k) ok = push(k, k + 1);
            } else {
                for (uint32_t k = 0; k < b.n_vertices && ok; ++k) ok = push(k, (k + 1) % b.n_vertices);
            }
            if (!ok) return fail(ctx, RXR_ERR_UNSUPPORTED, "batch2d: line end point NaN or beyond +-2^30");
        }
        if (risky2d)
            for (size_t q = p2_first; q < p2cur; ++q) {
                Prim2D &T = p2[q];
                uint32_t x0 = std::max(T.bx & 0xFFFFu, clip_x0), x1 = std::min(T.bx >> 16, clip_x1), y0 = std::max(T.by & 0xFFFFu, clip_y0), y1 = std::min(T.by >> 16, clip_y1);
                if (!(x0 < x1 && y0 < y1)) x0 = x1 = y0 = y1 = 0u;
                T.bx = x0 | (x1 << 16);
                T.by = y0 | (y1 << 16);
            }
    }
    if (f->background_kind == RXR_BG_HOST_PIXELS) memcpy(st + L.off_bg, f->background_pixels, (size_t)f->width * f->height * 4);

    // ---- device scratch ------------------------------------------------------------------------
    uint32_t tiles_x = (f->width + RXR_TILE_W - 1) / RXR_TILE_W, tiles_y_all = (f->height + RXR_TILE_H - 1) / RXR_TILE_H;
    size_t n_bins = (size_t)tiles_x * tiles_y_all;
    const size_t n_blocks = (size_t)((tiles_x + 3u) / 4u) * ((tiles_y_all + 3u) / 4u);  // k_blockscan's blocks of 4 x 4 bins
    if ((rc = ensure(ctx, ctx->d_tri_setup, (n_t3 ? n_t3 : 1) * sizeof(TriSetup))) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_tri_shade, (n_t3 ? n_t3 : 1) * sizeof(TriShade))) != RXR_OK) return rc;
    const size_t n_groups = (n_t3 + RXR_BLOCKSCAN_GROUP - 1) / RXR_BLOCKSCAN_GROUP;  // (their bin-range unions live behind the boxes)
    if ((rc = ensure(ctx, ctx->d_tri_box, ((n_t3 ? n_t3 : 1) + n_groups + 1) * sizeof(uint2))) != RXR_OK) return rc;
    const size_t n_chunks = (n_bins + RXR_SCAN_CHUNK - 1) / RXR_SCAN_CHUNK + 1;
    {
        // bin_count lives in its OWN buffer: the invariant "all-zero between launches" (k_raster hands every
        // bin back cleared) must hold for whatever frame size comes next, so nothing else may share it
        void *before = ctx->d_bin_count.p;
        // (behind the bin counts: k_blockscan's group counts per block of 4 x 4 bins, zero between launches like them)
        if ((rc = ensure(ctx, ctx->d_bin_count, (n_bins + 1 + n_blocks) * sizeof(uint32_t))) != RXR_OK) return rc;
        if (ctx->d_bin_count.p != before) HIPCHK(ctx, hipMemsetAsync(ctx->d_bin_count.p, 0, ctx->d_bin_count.cap, ctx->stream));
    }
    if ((rc = ensure(ctx, ctx->d_bins, (2 * (n_bins + 1) + 2 * n_chunks + 8) * sizeof(uint32_t))) != RXR_OK) return rc;
    // (the large list doubles as k_blockscan's per-block group lists: the two are never used by the same launch)
    if ((rc = ensure(ctx, ctx->d_large, std::max<size_t>(n_t3 ? n_t3 : 1, n_blocks * RXR_BLOCKSCAN_BLOCK_GROUPS + n_groups + 1) * sizeof(uint32_t))) != RXR_OK) return rc;
    // 2D binning scratch (used only when the frame has more than RXR_STAGE_TRIS 2D primitives)
    const bool binned2d = p2cur > RXR_STAGE_TRIS;
    if (binned2d) {
        void *before = ctx->d_bin2d_count.p;
        if ((rc = ensure(ctx, ctx->d_bin2d_count, (n_bins + 1) * sizeof(uint32_t))) != RXR_OK) return rc;
        if (ctx->d_bin2d_count.p != before) HIPCHK(ctx, hipMemsetAsync(ctx->d_bin2d_count.p, 0, ctx->d_bin2d_count.cap, ctx->stream));
        if ((rc = ensure(ctx, ctx->d_bins2d, (2 * (n_bins + 1) + 2 * n_chunks + 8) * sizeof(uint32_t))) != RXR_OK) return rc;
        if ((rc = ensure(ctx, ctx->d_large2d, p2cur * sizeof(uint32_t))) != RXR_OK) return rc;
        size_t want2d = ctx->list_floor ? ctx->list_floor : std::max<size_t>(1u << 18, p2cur * 8);
        // k_blockscan2d: every bin its own run of RXR_BLOCKSCAN_CAP slots, lists in submission order (no sort per tile)
        {
            bool on = true;
            if (const char *bs = getenv("RXR_BLOCKSCAN2D")) on = bs[0] != '0';
            ctx->blockscan2d_off = !(on && !ctx->list_floor && p2cur * n_blocks <= RXR_BLOCKSCAN_MAX_WORK / 4u &&
                                     (size_t)n_bins * RXR_BLOCKSCAN_CAP <= (64u << 20)) ||
                                   ctx->blockscan2d_bad.has(p2cur, n_bins);
            if (!ctx->blockscan2d_off) want2d = std::max<size_t>(want2d, (size_t)n_bins * RXR_BLOCKSCAN_CAP);
        }
        if (want2d > ctx->list2d_capacity) {
            if ((rc = ensure(ctx, ctx->d_list2d, want2d * sizeof(uint32_t))) != RXR_OK) return rc;
            ctx->list2d_capacity = (uint32_t)std::min<size_t>(ctx->d_list2d.cap / sizeof(uint32_t), 0xFFFFFFF0u);
        }
    }
    {
        void *before = ctx->d_counters.p;
        if ((rc = ensure(ctx, ctx->d_counters, 4 * CNT_WORDS * sizeof(uint32_t))) != RXR_OK) return rc;  // 2 sets for 3D, 2 for 2D
        if (ctx->d_counters.p != before) HIPCHK(ctx, hipMemsetAsync(ctx->d_counters.p, 0, ctx->d_counters.cap, ctx->stream));
    }
    size_t want_list = ctx->list_floor ? ctx->list_floor : std::max<size_t>(1u << 20, n_t3 * 4);
    // mid-sized scenes: k_blockscan gives every bin its own run of slots (not with a list floor: that knob exists to make the
    // general pipeline's lists overflow in tests)
    ctx->blockscan_cap = RXR_BLOCKSCAN_CAP;  // (both knobs are read per upload: tests and A-B runs switch them on a live context)
    ctx->blockscan_enabled = true;
    if (const char *bs = getenv("RXR_BLOCKSCAN")) ctx->blockscan_enabled = bs[0] != '0';
    if (const char *bc = getenv("RXR_BLOCKSCAN_CAP")) {  // tests: few slots per bin make ordinary meshes overflow them
        const long v = atol(bc);
        if (v > 0 && v <= 4096) ctx->blockscan_cap = (uint32_t)v;
    }
    // (frames beyond 8K x 8K tiles: fewer slots per bin keep k_blockscan inside its list budget -- their bins are thinner too; a bin that
    // overflows sends the frame through the general pipeline as ever)
    if (!getenv("RXR_BLOCKSCAN_CAP"))
        while (ctx->blockscan_cap > 64u && (size_t)n_bins * ctx->blockscan_cap > (64u << 20)) ctx->blockscan_cap /= 2u;
    ctx->blockscan_off = !(ctx->blockscan_enabled && !ctx->list_floor && n_t3 > RXR_STAGE_TRIS &&
                           (n_groups > RXR_BLOCKSCAN_SCATTER_GROUPS || n_groups * n_blocks <= RXR_BLOCKSCAN_MAX_WORK) &&
                           (size_t)n_bins * ctx->blockscan_cap <= (64u << 20));
    if (!ctx->blockscan_off && ctx->blockscan_bad.has(n_t3, n_bins)) ctx->blockscan_off = true;
    if (!ctx->blockscan_off) want_list = std::max<size_t>(want_list, (size_t)n_bins * ctx->blockscan_cap);
    if (want_list > ctx->list_capacity) {
        if ((rc = ensure(ctx, ctx->d_list, want_list * sizeof(uint32_t))) != RXR_OK) return rc;
        ctx->list_capacity = (uint32_t)std::min<size_t>(ctx->d_list.cap / sizeof(uint32_t), 0xFFFFFFF0u);
    }
    if ((rc = ensure(ctx, ctx->d_fb, (size_t)f->width * f->height * 4)) != RXR_OK) return rc;

    if (arrays_shipped) {
        // the projected arrays [off_pv, off_tinfo) left while they were being copied; what surrounds them follows
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame.p, st, L.off_pv, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync((uint8_t *)ctx->d_frame.p + L.off_tinfo, st + L.off_tinfo, L.total - L.off_tinfo, hipMemcpyHostToDevice, ctx->stream));
    } else {
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame.p, st, L.total, hipMemcpyHostToDevice, ctx->stream));
    }

    // ---- parameter block -----------------------------------------------------------------------
    RasterParams &P = ctx->P;
    memset(&P, 0, sizeof(P));
    P.width = f->width;
    P.height = f->height;
    P.tiles_x = tiles_x;
    P.flags = f->flags;
    P.fwidth = W;
    P.fheight = H;
    memcpy(P.inv_view, f->inverse_view, 64);
    memcpy(P.inv_proj, f->inverse_projection, 64);
    memcpy(P.cam, f->camera_pos, 12);
    memcpy(P.translationd2, f->translationd2, 8);
    P.scaled2 = f->scaled2;
    P.hash_anim = f->hash_anim;
    P.sample_mode = f->sample_mode;
    P.background_color = pack_px(f->background_color);
    P.background_kind = f->background_kind;
    memcpy(P.bg_grid, f->background_grid, 16);
    P.has_brush = f->has_brush_preview ? 1u : 0u;
    memcpy(P.brush_pos, f->brush_position, 12);
    P.brush_radius = f->brush_radius;
    P.brush_falloff = f->brush_falloff;
    memcpy(P.ambient, f->ambient, 16);
    memcpy(P.sun_dir, f->sun_dir, 12);
    P.day_factor = f->day_factor;
    P.n_tris3d = (uint32_t)n_t3;
    P.n_batches3d = n_b3;
    P.n_lights = f->n_lights;
    P.n_occluders = f->n_occluders;
    P.any_occluders = n_occ_total ? 1u : 0u;
    P.n_linedefs = f->n_linedefs;
    P.n_prims2d = (uint32_t)p2cur;
    P.binned2d = binned2d ? 1u : 0u;
    // the 2D pass runs after the 3D passes on the SAME Execution (rasterizer.rs:310, :501): `normal` and `opacity.x` are
    // assigned by every 3D fragment and never by the 2D loop, `hitpoint.z` by every 3D fragment that runs a program
    if (any_3d_visible && (f->flags & RXR_FLAG_D3_ACTIVE) && (reads_2d & (PF_NORMAL | PF_OPACITY)))
        return fail(ctx, RXR_ERR_UNSUPPORTED, "a 2D batch's program reads normal / opacity, which in the reference hold whatever the tile's last 3D fragment left there");
    if (any_3d_program && (f->flags & RXR_FLAG_D3_ACTIVE) && (reads_2d & PF_HITPOINT))
        return fail(ctx, RXR_ERR_UNSUPPORTED, "a 2D batch's program reads hitpoint while 3D batches run programs: hitpoint.z would hold the tile's last 3D program fragment's");
    // `emissive` is only ever written by SetEmissive and never reset by the raster loops: every opaque 3D fragment adds whatever
    // the last SetEmissive executed in its TILE left behind (rasterizer.rs:310, :1323, :1394) -- a function of the tile size and of
    // the traversal order, which a per-fragment evaluation cannot (and should not) reproduce.  A frame is accepted when that
    // state cannot be observed: no program of a 3D batch on screen writes emissive, or EVERY opaque 3D batch on screen runs a
    // program that assigns emissive itself on every path before the fragment reads it.
    if (emissive_writer_3d && opaque_without_emissive && (f->flags & RXR_FLAG_D3_ACTIVE))
        return fail(ctx, RXR_ERR_UNSUPPORTED,
                    "a 3D batch's program writes emissive while another opaque 3D batch of the frame does not assign it on every path: in the "
                    "reference that batch's fragments would add the emissive of whichever fragment ran before them in the tile");
    ctx->frame_uses_programs = uses_programs;
    P.vm_code = (const uint32_t *)ctx->d_vm_code.p;
    // (the brush preview and the grid background are editor-only: they live in the feature levels >= 1 so that k_raster does not
    // carry them -- merely compiling the grid shader into it cost the bench frame 9 % more VALU instructions in SGPR spills)
    const bool editor_paths = (P.has_brush && (f->flags & RXR_FLAG_D3_ACTIVE)) || f->background_kind == RXR_BG_GRID;
    P.kernel_level = std::max(ctx->min_kernel_level, uses_programs ? 2u : ((uses_chunk_tex || editor_paths) ? 1u : 0u));
    ctx->frame_needs_chunk_paths = uses_chunk_tex || editor_paths || ctx->min_kernel_level >= 1u;  // (rxr_jit_launch: level 8 otherwise)
    P.plain_programs = (!ctx->frame_needs_chunk_paths && !getenv("RXR_NO_PLAIN_PROGRAMS")) ? 1u : 0u;
    {
        // RXR_LIGHT_MATH (read per frame; rxr_set_light_math sets the context's own default): "exact" -- the light loop in the
        // reference's correctly rounded operations; "relaxed" -- point lights through rsq / rcp products, within the 1-per-channel
        // tolerance of lit 3D fragments (shade3d_lights<X, true>; feature levels 0 and 1)
        const char *lm = getenv("RXR_LIGHT_MATH");
        P.relaxed_lights = lm ? (lm[0] == 'r' ? 1u : 0u) : (ctx->relaxed_lights ? 1u : 0u);
        // the fused point-light term multiplies where the reference branches (a light out of range contributes intensity * 0): with an
        // infinite or NaN intensity, colour, position or range that is NaN where the reference adds nothing -- and so it is with FINITE
        // parameters whose product overflows (colour -3e38 x flicker 3e38 = inf, times the 0 of an out-of-range fragment: found by
        // tools/fuzz_special2.py, seed 1036).  Such frames -- no real scene has them -- take the exact loop, which skips and branches
        // like the reference (light.rs:535-552): every factor of the fused term must stay below 1e9 in magnitude (their product below 1e36).
        for (uint32_t i = 0; i < f->n_lights && P.relaxed_lights; ++i) {
            const rxr_light &l = f->lights[i];
            const float v[] = {l.intensity, l.color[0], l.color[1], l.color[2], l.position[0], l.position[1], l.position[2], l.start_distance, l.end_distance, l.flicker};
            for (float x : v)
                if (!(std::fabs(x) <= 1.0e9f)) P.relaxed_lights = 0u;
        }
        P.rl_flip_guard = 1e-4f;
        if (const char *fg = getenv("RXR_RL_FLIP_GUARD")) {  // tests: a large guard sends every wave down the exact normal sequences
            const float v = (float)atof(fg);
            if (v >= 1e-4f) P.rl_flip_guard = v;
        }
    }
    if (P.kernel_level == 2u && uses_programs && ctx->programs_static) P.kernel_level = 3u;  // k_raster_vm_s: wave-uniform stack pointer
    // k_raster_vm_sv: ... and no program decides whether an opaque fragment is written, so the visibility loop is the one of
    // k_raster_chunk, without a call of the interpreter in it
    if (P.kernel_level == 3u && !vis_programs && !getenv("RXR_VM_VIS_CALLS")) P.kernel_level = 4u;
    else if (P.kernel_level == 2u && uses_programs && !vis_programs && !getenv("RXR_VM_VIS_CALLS")) P.kernel_level = 5u;  // k_raster_vm_v
    P.programs = (const DevProgram *)ctx->d_programs.p;
    P.patterns = (const DevPattern *)ctx->d_patterns.p;
    P.pattern_data = (const float *)ctx->d_pattern_data.p;
    P.palette = (const float *)ctx->d_palette.p;
    P.n_programs = (uint32_t)ctx->programs.size();
    P.n_patterns = ctx->n_patterns;
    P.n_normal_patterns = ctx->n_normal_patterns;
    P.n_palette = ctx->n_palette;
    P.vm_fault = ctx->d_host_status + HS_VM_FAULT;
    P.staircase_overflow = ctx->d_host_status + HS_STAIRCASE;
    P.time = f->time;
    {
        // tiles outside the union of the 2D pixel boxes skip the 2D pass without touching memory
        uint32_t bx0 = 0xFFFFu, bx1 = 0, by0 = 0xFFFFu, by1 = 0;
        for (size_t i = 0; i < (use_meshes2d ? 0 : p2cur); ++i) {  // (device-projected: the records do not exist yet -- d2_box_dev below)
            uint32_t a = p2[i].bx & 0xFFFFu, b = p2[i].bx >> 16, c = p2[i].by & 0xFFFFu, d = p2[i].by >> 16;
            if (a < b && c < d) {
                bx0 = std::min(bx0, a); bx1 = std::max(bx1, b); by0 = std::min(by0, c); by1 = std::max(by1, d);
            }
        }
        P.d2_box[0] = bx0; P.d2_box[1] = bx1; P.d2_box[2] = by0; P.d2_box[3] = by1;
        if (by0 < by1) {  // (the 2D pass only ever writes inside its primitives' pixel boxes)
            content_y0 = std::min(content_y0, by0);
            content_y1 = std::max(content_y1, std::min(by1, f->height));
            add_span(bx0, std::min(bx1, f->width), by0, std::min(by1, f->height));
        }
        P.d2_box_dev = use_meshes2d ? ctx->PP2.d2_box : nullptr;  // (the device builds these records: their boxes are not known here)
    }
    P.list2d_capacity = ctx->list2d_capacity;
    P.any_lights = f->n_lights ? 1u : 0u;
    P.has_opacity = has_opacity ? 1u : 0u;
    {
        // frames in which row mode has to work around some candidates (rxr_kernels.hip SPLITR): a kept batch of the opaque pass whose
        // fragments need their texel's alpha, or -- under an opacity pass -- one that carries a profile id
        bool split = false;
        const DevBatch *hb = (const DevBatch *)(st + L.off_b3);
        for (uint32_t i = 0; i < n_b3 && !split; ++i) {
            if (hb[i].flags & (DB_SKIP | DB_OPACITY_LIST)) continue;
            split = (hb[i].flags & (DB_ALPHA_TEST | DB_FULL_ALPHA)) != 0 || (has_opacity && (hb[i].flags & DB_HAS_PROFILE));
        }
        P.split_rounds = ((split && !getenv("RXR_NO_SPLIT_ROUNDS")) || getenv("RXR_FORCE_SPLIT_ROUNDS")) ? 1u : 0u;   // (the variables: A-B runs, tests)
    }
    P.list_capacity = ctx->list_capacity;
    uint8_t *d = (uint8_t *)ctx->d_frame.p;
    P.pv = (const float4 *)(d + L.off_pv);
    P.uv = (const float2 *)(d + L.off_uv);
    P.nrm = (const float *)(d + L.off_nrm);
    P.idx = (const uint32_t *)(d + L.off_idx);
    P.edges = (const rxr_edges *)(d + L.off_edges);
    P.edge_vis3d = edgeless3d == 1 ? (const uint32_t *)(d + L.off_edges) : nullptr;  // (the same pool, a word per triangle instead of a record)
    P.batches3d = (const DevBatch *)(d + L.off_b3);
    P.batch_tri_base = (const uint32_t *)(d + L.off_base);
    P.tri_info = with_tri_info ? (const uint2 *)(d + L.off_tinfo) : nullptr;
    P.batch_clip3d = any_risky3d ? (const uint4 *)(d + L.off_clip3d) : nullptr;
    P.ref_tile = f->tile_size;
    P.tri_setup = (TriSetup *)ctx->d_tri_setup.p;
    P.tri_shade = (TriShade *)ctx->d_tri_shade.p;
    P.tri_box = (uint2 *)ctx->d_tri_box.p;
    P.group_rng = P.tri_box + (n_t3 ? n_t3 : 1);
    P.blk_cnt = (uint32_t *)ctx->d_bin_count.p + n_bins + 1;
    P.blk_grp = (uint32_t *)ctx->d_large.p;
    P.blockscan_scatter = n_groups > RXR_BLOCKSCAN_SCATTER_GROUPS ? 1u : 0u;
    P.blk_wide_base = (uint32_t)(n_blocks * RXR_BLOCKSCAN_BLOCK_GROUPS);
    P.bin_count = (uint32_t *)ctx->d_bin_count.p;
    P.bin_offset = (uint32_t *)ctx->d_bins.p;
    P.bin_cursor = P.bin_offset + n_bins + 1;
    P.chunk_tot = P.bin_cursor + n_bins + 1;
    P.chunk_base = P.chunk_tot + n_chunks;
    P.host_status = ctx->d_host_status;
    P.bin_list = (uint32_t *)ctx->d_list.p;
    P.large_list = (uint32_t *)ctx->d_large.p;
    P.counters = (uint32_t *)ctx->d_counters.p;
    P.lights = (const rxr_light *)(d + L.off_lights);
    P.lights_fast = (const LightFast *)(d + L.off_lights_fast);
    P.occluders = (const rxr_occluder *)(d + L.off_occ);
    P.linedefs = (const rxr_linedef *)(d + L.off_ld);
    P.chunks = (const ChunkRange *)(d + L.off_chunks);
    P.batches2d = (const DevBatch *)(d + L.off_b2);
    P.prim2d = (const Prim2D *)(d + L.off_p2);
    if (binned2d) {
        P.bin2d_count = (uint32_t *)ctx->d_bin2d_count.p;
        P.bin2d_offset = (uint32_t *)ctx->d_bins2d.p;
        P.bin2d_cursor = P.bin2d_offset + n_bins + 1;
        P.chunk2d_tot = P.bin2d_cursor + n_bins + 1;
        P.chunk2d_base = P.chunk2d_tot + n_chunks;
        P.bin2d_list = (uint32_t *)ctx->d_list2d.p;
        P.large2d_list = (uint32_t *)ctx->d_large2d.p;
        P.host_status2d = ctx->d_host_status + CNT_WORDS;
    }
    P.texels = (const uint32_t *)ctx->d_texels.p;
    P.tex = (const DevTexDesc *)(d + L.off_tdesc);
    P.frame_texels = (const uint32_t *)(d + L.off_ltex);
    P.bg_pixels = (const uint32_t *)(d + L.off_bg);
    ctx->frame_uses_meshes2d = use_meshes2d;
    if (use_meshes2d) {
        Project2DParams &PP2 = ctx->PP2;
        PP2.has_matrix = ctx->has_matrix2d ? 1u : 0u;
        memcpy(PP2.m, ctx->matrix2d, sizeof(PP2.m));
        PP2.width = W;
        PP2.height = H;
        PP2.ref_tile = f->tile_size;
        PP2.out = (Prim2D *)(d + L.off_p2);
    }
    ctx->frame_uses_meshes = use_meshes;
    if (use_meshes) {
        // the raster pre-pass reads the pools the projection kernels write
        ProjectParams &PP = ctx->PP;
        memcpy(PP.projection, f->projection, 64);
        PP.width = W;
        PP.height = H;
        PP.meshes = (const DevMesh *)((uint8_t *)ctx->d_proj_misc.p + ctx->pp_off_meshes);
        P.pv = PP.pv;
        P.uv = PP.uv;
        P.nrm = PP.nrm;
        P.idx = PP.idx;
        P.edges = PP.edges;
        P.edge_vis3d = nullptr;
        P.dev_bbox = PP.bbox;
        P.mesh_live = PP.mesh_live;
        // the set-up builds the Edges records itself (k_proj_edges fused into make_setup): RXR_PROJ_FUSED_EDGES=0 keeps the pool
        static const bool fused_edges = !(getenv("RXR_PROJ_FUSED_EDGES") && atoi(getenv("RXR_PROJ_FUSED_EDGES")) == 0);
        PP.edges_in_setup = fused_edges ? 1u : 0u;
        P.pm_meshes = fused_edges ? PP.meshes : nullptr;
        P.pm_edge_vis = fused_edges ? PP.edge_vis : nullptr;
    }
    if (e2e_timing)
        fprintf(stderr, "rxr_e2e_timing upload: validate+size %.3f, wait for the previous frame %.3f, headers %.3f, arrays (copy + ship) %.3f, rest %.3f ms\n", ut_validated,
                ut_quiesced - ut_validated, ut_headers - ut_quiesced, ut_arrays - ut_headers, ut_ms() - ut_arrays);
    ctx->n_tris2d = (uint32_t)t2cur;
    // Known content rows: host-projected batches only (the device-projected ones have their boxes on the device), 3D mode (the miss
    // colour is then the constant [0,0,0,255], :420-461; in 2D mode the background is evaluated per pixel), no brush preview (it paints
    // missed pixels, :435-458).  RXR_CONTENT_ROWS=0 switches the clamp off (A-B runs, tests).
    const bool content_clamp = !(getenv("RXR_CONTENT_ROWS") && atoi(getenv("RXR_CONTENT_ROWS")) == 0);  // (read per upload: tests switch it on a live context)
    ctx->content_known = content_clamp && !use_meshes && !use_meshes2d && (f->flags & RXR_FLAG_D3_ACTIVE) && !P.has_brush && f->tile_size > 0;
    ctx->content_row0 = std::min(content_y0, content_y1);
    ctx->content_row1 = content_y1;
    // the spans pay when they leave out a good part of the content rows' tiles (a table look-up in front of every tile otherwise buys nothing)
    ctx->spans_active = false;
    if (ctx->content_known && span_budget >= 0 && n_tile_rows <= RXR_MAX_TILE_ROWS && ctx->content_row0 < ctx->content_row1 &&
        !(getenv("RXR_ROW_SPANS") && atoi(getenv("RXR_ROW_SPANS")) == 0)) {
        const uint32_t r0 = ctx->content_row0 / RXR_TILE_H, r1 = std::min((ctx->content_row1 + RXR_TILE_H - 1u) / RXR_TILE_H, n_tile_rows);
        size_t inside = 0;
        for (uint32_t r = 0; r < n_tile_rows; ++r) {
            const bool any = span_lo[r] < span_hi[r];
            ctx->h_row_spans[r] = make_uint2(any ? span_lo[r] : 0u, any ? span_hi[r] : 0u);
            if (r >= r0 && r < r1 && any) inside += span_hi[r] - span_lo[r];
        }
        if (inside * 100u <= (size_t)(r1 - r0) * n_tile_cols * 85u && (size_t)(r1 - r0) * n_tile_cols - inside >= content_min_tiles()) {
            int rc2;
            if ((rc2 = ensure(ctx, ctx->d_row_spans, RXR_MAX_TILE_ROWS * sizeof(uint2))) != RXR_OK) return rc2;
            HIPCHK(ctx, hipMemcpyAsync(ctx->d_row_spans.p, ctx->h_row_spans, n_tile_rows * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
            ctx->spans_active = true;
        }
    }
    // Device-projected 3D meshes: their boxes are made on the device, frame by frame -- the table goes there holding what is known here
    // (the pixel box of a host-projected 2D pass) and k_spans_from_meshes completes it behind the projections (render_impl).  The grid is not narrowed
    // (nothing here knows by how much); the workgroups outside the spans leave at once.  Large frames only: what a dense one pays is the
    // table's kernel, a fill launch that finds nothing to fill and the look-up in front of every tile.
    ctx->dev_spans = false;
    if (content_clamp && use_meshes && (f->flags & RXR_FLAG_D3_ACTIVE) && !P.has_brush && f->tile_size > 0 && span_budget >= 0 &&
        n_tile_rows <= RXR_MAX_TILE_ROWS && n_b3 && n_b3 <= rxr_span_meshes_max() && (size_t)n_tile_rows * n_tile_cols >= 4u * content_min_tiles() &&
        !(getenv("RXR_ROW_SPANS") && atoi(getenv("RXR_ROW_SPANS")) == 0)) {
        for (uint32_t r = 0; r < n_tile_rows; ++r) {
            const bool any = span_lo[r] < span_hi[r];
            ctx->h_row_spans[r] = make_uint2(any ? span_lo[r] : 0u, any ? span_hi[r] : 0u);
        }
        int rc2;
        if ((rc2 = ensure(ctx, ctx->d_row_spans, RXR_MAX_TILE_ROWS * sizeof(uint2))) != RXR_OK) return rc2;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_row_spans.p, ctx->h_row_spans, n_tile_rows * sizeof(uint2), hipMemcpyHostToDevice, ctx->stream));
        ctx->dev_spans = true;
    }
    ctx->has_frame = true;
    ctx->rendered = false;
    ctx->upload_ordered_on = nullptr;
    return RXR_OK;
}

// the rows [c0, c1) of a contiguous band spec that the resident frame can draw in, rounded out to tile rows (false: not known -- all of them)
static bool content_band(const rxr_ctx *ctx, const RenderSpec &spec, uint32_t &c0, uint32_t &c1) {
    c0 = spec.row0;
    c1 = spec.row1;
    if (!(ctx->content_known && spec.tile_stride == 1u && !spec.compact && spec.row0 < spec.row1)) return false;
    c0 = std::min(std::max(ctx->content_row0 / RXR_TILE_H * RXR_TILE_H, spec.row0), spec.row1);
    c1 = std::max(std::min((ctx->content_row1 + RXR_TILE_H - 1u) / RXR_TILE_H * RXR_TILE_H, spec.row1), c0);
    // (an empty tile costs the raster launch about half a nanosecond of the chip's time, a fill launch two to three microseconds: the teapot
    // at 1080p lost 5 us of set-up to save 3 of raster.  Fewer than content_min_tiles() empty tiles: the whole band is rastered)
    const size_t saved_tiles = (size_t)((c0 - spec.row0) + (spec.row1 - c1)) / RXR_TILE_H * ctx->P.tiles_x;
    if (saved_tiles < content_min_tiles()) {
        c0 = spec.row0;
        c1 = spec.row1;
        return false;
    }
    return true;
}

// n_raster_bands > 1 (contiguous band specs only): ONE pre-pass over the spec's rows, then the raster kernel in that many launches over
// consecutive groups of tile rows, an event of band_events recorded behind each -- the caller ships finished rows while the next ones
// render (rxr_render_download).  The bins are those of the one pre-pass (RasterParams.bin_row0); every bin is still handed back zeroed
// by its own tile's workgroup; the frame is byte-identical to one launch (tests/test_gpu_parity.py).
// spans_event (device-projected sparse frames): recorded behind k_spans_from_meshes, which then also writes the completed row-span
// table to the second half of ctx->h_row_spans; *spans_recorded says whether that happened in this call.
static int render_impl(rxr_ctx *ctx, const RenderSpec &spec, void *dev_pixels, hipStream_t s, bool retry = false, uint32_t n_raster_bands = 1,
                       hipEvent_t *band_events = nullptr, uint32_t *band_row_of = nullptr, hipEvent_t spans_event = nullptr,
                       bool *spans_recorded = nullptr) {
    if (!ctx) return RXR_ERR_INVALID;
    if (!ctx->has_frame) return fail(ctx, RXR_ERR_INVALID, "render: no frame uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // every launch sequence uses the context's one set of scratch buffers (records, bins, counters): a render on another
    // stream than the previous one is ordered behind it
    // (the event is recorded here, at the switch, on the PREVIOUS stream -- it then covers that stream's renders and whatever the
    // caller queued behind them, a superset -- and not behind every launch sequence: an event record is a barrier packet that idles
    // the GPU for microseconds, and a caller that stays on one stream never needs it)
    if (ctx->rendered && ctx->last_stream != s) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_render, ctx->last_stream));
        HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_render, 0));
    }
    if (retry) ctx->rerenders++;
    if (!ctx->upload_ordered_on || ctx->upload_ordered_on != s) {
        if (s != ctx->stream) {
            // the upload ran on the context stream: order the external stream behind it (once per upload)
            HIPCHK(ctx, hipEventRecord(ctx->ev_upload, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(s, ctx->ev_upload, 0));
        }
        ctx->upload_ordered_on = s;
    }
    RasterParams P = ctx->P;
    P.row0 = spec.row0;
    P.row1 = spec.row1;
    P.tile_y0 = spec.tile_y0;
    P.tile_stride = spec.tile_stride;
    P.tiles_y = spec.tiles_y;
    P.compact = spec.compact ? 1u : 0u;
    P.out = (uint32_t *)dev_pixels;
    P.out_row_stride = P.width;
    P.out_base_row = (spec.external && !spec.compact) ? (int64_t)spec.row0 : 0;
    // Rows that nothing of the frame can reach (ctx->content_row0 / 1) take the miss colour from a fill at memory speed; the pre-pass and
    // the raster kernel are launched over the tile rows in between only.  A sparse frame pays for its empty tiles otherwise: a workgroup
    // each that fetches its bin's length to learn that there is nothing to do -- on the 1 M-triangle grid (the 84 tile rows above the grid
    // are empty) 0.575 -> 0.531 ms for the band alone (tools/band_probe.py, profiles/r04).  Contiguous bands only (not the stripe
    // launches of a multi-GPU share).
    uint32_t fill_a0 = 0, fill_a1 = 0, fill_b0 = 0, fill_b1 = 0;  // rows [a0, a1) above and [b0, b1) below the content
    uint32_t c0 = 0, c1 = 0;
    if (content_band(ctx, spec, c0, c1)) {
        fill_a0 = spec.row0; fill_a1 = c0; fill_b0 = c1; fill_b1 = spec.row1;
        P.row0 = c0;
        P.row1 = c1;
        P.tile_y0 = c0 / RXR_TILE_H;
        P.tiles_y = c1 > c0 ? (c1 + RXR_TILE_H - 1u) / RXR_TILE_H - P.tile_y0 : 0u;
    }
    P.row_spans = nullptr;
    bool use_spans = ctx->spans_active && fill_a0 < fill_b1 /* (the clamp above applied) */ && P.tiles_y;
    auto widest_span = [&](uint32_t first_row, uint32_t n_rows) {
        uint32_t w = 1u;
        for (uint32_t r = first_row; r < first_row + n_rows && r < RXR_MAX_TILE_ROWS; ++r) w = std::max(w, ctx->h_row_spans[r].y - ctx->h_row_spans[r].x);
        return w;
    };
    const size_t n_bins = (size_t)P.tiles_x * P.tiles_y;

    // Kernel timing is opt-in (rxr_profile_begin): per-dispatch start / stop events (rxr_launch.h), no event record on the stream.
    // Without it rxr_stats' *_us fields stay zero.
    const bool timed = !retry && !ctx->prof.empty() && (ctx->prof_calls++ % ctx->prof_stride) == 0u;  // (a re-render after a list overflow takes no profiling slot)
    ProfSlot *slot = nullptr;
    if (timed) {
        slot = &ctx->prof[ctx->prof_next % ctx->prof.size()];
        slot->n = 0;
        slot->raster_first = RXR_PROF_MAX_KERNELS;
        ctx->prof_next++;
    }
    ctx->last_prof = slot;
    // (every kernel this thread launches until the guard goes takes a start / stop pair of the slot: rxr_launch.h)
    struct TimesGuard {
        explicit TimesGuard(ProfSlot *p) { rxr_launch_times = p; }
        ~TimesGuard() { rxr_launch_times = nullptr; }
    } times_guard(slot);
    // small scenes: one staging round of k_raster holds every triangle -> no set-up / binning launches at all
    const bool d3 = P.tiles_y && (P.flags & RXR_FLAG_D3_ACTIVE);
    P.fused_small = (d3 && P.n_tris3d <= RXR_STAGE_TRIS) ? ctx->small_mode : 0u;
    if (P.kernel_level && P.fused_small == 1u) P.fused_small = 2u;  // k_raster_vm reads the records k_setup3d writes
    if (d3 && P.fused_small) {
        if (ctx->frame_uses_meshes) rxr_launch_project(&ctx->PP, s);
        if (P.fused_small == 2u) rxr_launch_setup(&P, s);  // records only; no counters, bins or lists are touched
    }
    use_spans = use_spans && rxr_raster_takes_spans(&P) != 0;  // (the kernel this launch gets must be one that looks the table up)
    if (use_spans) {
        // the pixels to the left and right of each tile row's span, and -- in the same launch: their spans are empty -- the rows above
        // and below the content (one launch of 13 us where two fills of whole rows and one of row ends took 16 and two more gaps)
        P.row_spans = (const uint2 *)ctx->d_row_spans.p;
        RasterParams Pf = P;
        Pf.row0 = spec.row0;
        Pf.row1 = spec.row1;
        Pf.tile_y0 = spec.tile_y0;
        Pf.tiles_y = spec.tiles_y;
        rxr_launch_fill_outside_spans(&Pf, s);
    } else {
        for (int part = 0; part < 2; ++part) {  // the rows outside the content: the 3D miss colour [0, 0, 0, 255] (:420-461)
            const uint32_t a = part ? fill_b0 : fill_a0, b = part ? fill_b1 : fill_a1;
            if (a < b) rxr_launch_fill_words(P.out + (size_t)((int64_t)a - P.out_base_row) * P.out_row_stride, (uint64_t)(b - a) * P.out_row_stride, 0xFF000000u, s);
        }
    }
    const bool prepass = d3 && !P.fused_small;
    // (device-projected frames: the spans are completed on the device, behind the projection -- see rxr_upload_frame)
    const bool dev_spans = prepass && ctx->dev_spans && ctx->frame_uses_meshes && spec.tile_stride == 1u && !spec.compact && rxr_raster_takes_spans(&P) != 0;
    // device-projected 2D batches: Batch2D::project + the Prim2D records, before anything reads them
    // (a frame without 2D primitives never reads what they would write: no launch)
    const bool project2d = ctx->frame_uses_meshes2d && P.tiles_y && (P.flags & RXR_FLAG_D2_ACTIVE) && ctx->PP2.n_prims;
    bool projected2d = false;
    auto project_meshes = [&]() {
        rxr_launch_project(&ctx->PP, s);  // clip_and_project + Edges + boxes on the device
        if (dev_spans) {
            if (project2d) {  // (its union box belongs to the table)
                rxr_launch_project2d(&ctx->PP2, s);
                projected2d = true;
            }
            P.row_spans = (const uint2 *)ctx->d_row_spans.p;
            rxr_launch_spans_from_meshes(&P, (P.height + RXR_TILE_H - 1u) / RXR_TILE_H, project2d ? ctx->PP2.d2_box : nullptr,
                                         spans_event ? ctx->h_row_spans + RXR_MAX_TILE_ROWS : nullptr, s);
            if (spans_event && hipEventRecord(spans_event, s) == hipSuccess && spans_recorded) *spans_recorded = true;
            rxr_launch_fill_outside_spans(&P, s);
        }
    };
    if (prepass && ctx->scratch_dirty) {
        // a previous launch sequence was cut short: restore the all-zero invariants explicitly
        HIPCHK(ctx, hipMemsetAsync(ctx->d_bin_count.p, 0, ctx->d_bin_count.cap, s));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_counters.p, 0, ctx->d_counters.cap, s));
    }
    // mid-sized scenes: k_setup3d (records and boxes only) + k_blockscan; the counters are not touched (the clean set stays clean, the
    // large list stays empty: every block of bins looks at every triangle)
    const bool blockscan = prepass && !ctx->blockscan_off && (size_t)n_bins * ctx->blockscan_cap <= (size_t)P.list_capacity;
    ctx->last_used_blockscan = blockscan;
    if (blockscan) {
        ctx->scratch_dirty = true;
        P.blockscan_cap = ctx->blockscan_cap;
        // (the scatter form counts its wide groups in the clean counter set and clears the other one, as k_scan does)
        P.counters = (uint32_t *)ctx->d_counters.p + (size_t)ctx->parity * CNT_WORDS;
        P.counters_next = (uint32_t *)ctx->d_counters.p + (size_t)(ctx->parity ^ 1u) * CNT_WORDS;
        if (P.blockscan_scatter) ctx->parity ^= 1u;
        if (ctx->frame_uses_meshes) project_meshes();
        rxr_launch_setup(&P, s);
        rxr_launch_blockscan(&P, s);
    } else if (prepass) {
        ctx->scratch_dirty = true;
        // counter set `parity` is clean (cleared by the previous launch's k_scan); this launch's k_scan
        // clears the other set.  bin_count is clean because k_raster hands every bin back zeroed.
        P.counters = (uint32_t *)ctx->d_counters.p + (size_t)ctx->parity * CNT_WORDS;
        P.counters_next = (uint32_t *)ctx->d_counters.p + (size_t)(ctx->parity ^ 1u) * CNT_WORDS;
        ctx->parity ^= 1u;
        if (ctx->frame_uses_meshes) project_meshes();
        rxr_launch_setup(&P, s);
        ScanArgs A{};
        A.n = P.tiles_x * P.tiles_y;
        A.list_capacity = P.list_capacity;
        A.count = P.bin_count;
        A.offset = P.bin_offset;
        A.cursor = P.bin_cursor;
        A.chunk_tot = P.chunk_tot;
        A.chunk_base = P.chunk_base;
        A.counters = P.counters;
        A.counters_next = P.counters_next;
        A.host_status = P.host_status;
        rxr_launch_scan(&A, s);
        rxr_launch_fill(&P, s);
    }
    // (the pinned status words are never written by the host while launches may be in flight: earlier queued k_scan
    // launches write them; rxr_synchronize clears them once the streams have drained)
    (void)n_bins;
    if (project2d && !projected2d) rxr_launch_project2d(&ctx->PP2, s);
    // 2D binning pre-pass (many 2D primitives): count -> scan -> fill; k_raster sorts each tile's list
    const bool prepass2d = P.tiles_y && (P.flags & RXR_FLAG_D2_ACTIVE) && P.binned2d;
    if (prepass2d) {
        if (ctx->scratch2d_dirty) {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_bin2d_count.p, 0, ctx->d_bin2d_count.cap, s));
            HIPCHK(ctx, hipMemsetAsync((uint32_t *)ctx->d_counters.p + 2 * CNT_WORDS, 0, 2 * CNT_WORDS * sizeof(uint32_t), s));
        }
        ctx->scratch2d_dirty = true;
        const bool blockscan2d = !ctx->blockscan2d_off && (size_t)n_bins * RXR_BLOCKSCAN_CAP <= (size_t)P.list2d_capacity;
        ctx->last_used_blockscan2d = blockscan2d;
        if (blockscan2d) {  // the counters stay clean (no large list, no ticket); the raster kernel skips its per-tile sort
            P.blockscan2d_cap = RXR_BLOCKSCAN_CAP;
            P.counters2d = (uint32_t *)ctx->d_counters.p + (size_t)(2u + ctx->parity2d) * CNT_WORDS;
            rxr_launch_blockscan2d(&P, s);
        } else {
        P.counters2d = (uint32_t *)ctx->d_counters.p + (size_t)(2u + ctx->parity2d) * CNT_WORDS;
        P.counters2d_next = (uint32_t *)ctx->d_counters.p + (size_t)(2u + (ctx->parity2d ^ 1u)) * CNT_WORDS;
        ctx->parity2d ^= 1u;
        rxr_launch_bin2d_count(&P, s);
        ScanArgs A{};
        A.n = P.tiles_x * P.tiles_y;
        A.list_capacity = P.list2d_capacity;
        A.count = P.bin2d_count;
        A.offset = P.bin2d_offset;
        A.cursor = P.bin2d_cursor;
        A.chunk_tot = P.chunk2d_tot;
        A.chunk_base = P.chunk2d_base;
        A.counters = P.counters2d;
        A.counters_next = P.counters2d_next;
        A.host_status = P.host_status2d;
        rxr_launch_scan(&A, s);
        rxr_launch_bin2d_fill(&P, s);
        }
    }
    if (slot) slot->raster_first = slot->n;
    if (n_raster_bands > 1u && spec.tile_stride == 1u && !spec.compact) {
        if (band_row_of)
            for (uint32_t k = 0; k <= n_raster_bands; ++k) band_row_of[k] = k ? spec.row1 : spec.row0;  // (a frame without content: one band of filled rows)
        const uint32_t rows_all = P.tiles_y, first_row = P.tile_y0, r0_all = P.row0, r1_all = P.row1;
        for (uint32_t k = 0; k < n_raster_bands; ++k) {
            const uint32_t a = (uint32_t)((uint64_t)rows_all * k / n_raster_bands), b = (uint32_t)((uint64_t)rows_all * (k + 1u) / n_raster_bands);
            P.bin_row0 = a;
            P.tile_y0 = first_row + a;
            P.tiles_y = b - a;
            P.row0 = std::max(r0_all, (first_row + a) * (uint32_t)RXR_TILE_H);
            P.row1 = std::min(r1_all, (first_row + b) * (uint32_t)RXR_TILE_H);
            if (band_row_of) {  // (the first and the last band take the filled rows above / below the content with them)
                band_row_of[k] = k ? P.row0 : spec.row0;
                band_row_of[k + 1u] = k + 1u < n_raster_bands ? P.row1 : spec.row1;
            }
            if (P.tiles_y && !rxr_jit_launch(ctx, &P, s)) rxr_launch_raster_grid(&P, use_spans ? widest_span(P.tile_y0, P.tiles_y) : 0u, s);
            if (band_events) HIPCHK(ctx, hipEventRecord(band_events[k], s));
        }
    } else if (!rxr_jit_launch(ctx, &P, s)) rxr_launch_raster_grid(&P, use_spans ? widest_span(P.tile_y0, P.tiles_y) : 0u, s);
    HIPCHK(ctx, hipGetLastError());
    ctx->scratch_dirty = false;  // the raster launch that hands the bins back is queued
    ctx->scratch2d_dirty = false;
    ctx->rendered = true;
    ctx->last_had_prepass = prepass;
    ctx->last_had_prepass2d = prepass2d;
    ctx->launches_since_sync++;
    ctx->last_spec = spec;
    ctx->last_out = dev_pixels;
    ctx->last_stream = s;
    ctx->stats.tiles_x = P.tiles_x;
    ctx->stats.tiles_y = P.tiles_y;
    ctx->stats.n_triangles3d = P.n_tris3d;
    ctx->stats.n_triangles2d = ctx->n_tris2d;
    return RXR_OK;
}

static int band_spec(rxr_ctx *ctx, uint32_t row0, uint32_t row1, bool external, RenderSpec &spec) {
    if (!ctx->has_frame) return fail(ctx, RXR_ERR_INVALID, "render: no frame uploaded");
    if (row0 > row1 || row1 > ctx->P.height) return fail(ctx, RXR_ERR_INVALID, "render: bad row range");
    spec.row0 = row0;
    spec.row1 = row1;
    spec.tile_y0 = row0 / RXR_TILE_H;
    spec.tile_stride = 1;
    spec.tiles_y = row1 > row0 ? (row1 + RXR_TILE_H - 1) / RXR_TILE_H - spec.tile_y0 : 0;
    spec.compact = false;
    spec.external = external;
    return RXR_OK;
}

int rxr_render_rows(rxr_ctx *ctx, uint32_t row0, uint32_t row1) {
    if (!ctx) return RXR_ERR_INVALID;
    if (ctx->group) {
        (void)row0;
        (void)row1;
        return fail(ctx, RXR_ERR_UNSUPPORTED, "rxr_render_rows on a multi-device context: use rxr_rasterize / rxr_render_download / rxr_render_gather, or drive the members (rxr_member) yourself");
    }
    RenderSpec spec{};
    int rc = band_spec(ctx, row0, row1, false, spec);
    if (rc != RXR_OK) return rc;
    return render_impl(ctx, spec, ctx->d_fb.p, ctx->stream);
}

int rxr_render_rows_to(rxr_ctx *ctx, uint32_t row0, uint32_t row1, void *dev_pixels, void *hip_stream) {
    if (!ctx) return RXR_ERR_INVALID;
    if (!dev_pixels) return fail(ctx, RXR_ERR_INVALID, "rxr_render_rows_to: dev_pixels is NULL");
    if (ctx->group) return fail(ctx, RXR_ERR_UNSUPPORTED, "rxr_render_rows_to on a multi-device context: device pointers and streams belong to ONE device (use rxr_member)");
    RenderSpec spec{};
    int rc = band_spec(ctx, row0, row1, true, spec);
    if (rc != RXR_OK) return rc;
    return render_impl(ctx, spec, dev_pixels, hip_stream ? (hipStream_t)hip_stream : ctx->stream);
}

int rxr_render_stripes_to(rxr_ctx *ctx, uint32_t first, uint32_t stride, void *dev_pixels, void *hip_stream) {
    if (!ctx) return RXR_ERR_INVALID;
    if (!dev_pixels) return fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_to: dev_pixels is NULL");
    if (ctx->group) return fail(ctx, RXR_ERR_UNSUPPORTED, "rxr_render_stripes_to on a multi-device context: device pointers and streams belong to ONE device (use rxr_member)");
    if (!ctx->has_frame) return fail(ctx, RXR_ERR_INVALID, "render: no frame uploaded");
    if (stride == 0) return fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_to: stride 0");
    const uint32_t n_stripes = (ctx->P.height + RXR_TILE_H - 1) / RXR_TILE_H;
    RenderSpec spec{};
    spec.row0 = 0;
    spec.row1 = ctx->P.height;
    spec.tile_y0 = first;
    spec.tile_stride = stride;
    spec.tiles_y = first < n_stripes ? (n_stripes - first + stride - 1) / stride : 0;
    spec.compact = true;
    spec.external = true;
    return render_impl(ctx, spec, dev_pixels, hip_stream ? (hipStream_t)hip_stream : ctx->stream);
}

static int stripes_spec(rxr_ctx *ctx, uint32_t first, uint32_t stride, RenderSpec &spec) {
    if (!ctx->has_frame) return fail(ctx, RXR_ERR_INVALID, "render: no frame uploaded");
    if (stride == 0) return fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes: stride 0");
    const uint32_t n_stripes = (ctx->P.height + RXR_TILE_H - 1) / RXR_TILE_H;
    spec.row0 = 0;
    spec.row1 = ctx->P.height;
    spec.tile_y0 = first;
    spec.tile_stride = stride;
    spec.tiles_y = first < n_stripes ? (n_stripes - first + stride - 1) / stride : 0;
    spec.compact = true;
    spec.external = true;
    return RXR_OK;
}

int rxr_render_stripes_batch(rxr_ctx *ctx, uint32_t first, uint32_t stride, uint32_t n_frames, void *dev_pixels, size_t frame_stride_bytes,
                             void *hip_stream) {
    if (!ctx) return RXR_ERR_INVALID;
    if (!dev_pixels) return fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_batch: dev_pixels is NULL");
    if (ctx->group) return rxr_group_render_stripes_batch(ctx, first, stride, n_frames, dev_pixels, frame_stride_bytes, hip_stream);
    RenderSpec spec{};
    int rc = stripes_spec(ctx, first, stride, spec);
    if (rc != RXR_OK) return rc;
    if (n_frames > 1u && frame_stride_bytes < (size_t)spec.tiles_y * RXR_TILE_H * ctx->P.width * 4u)
        return fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_batch: frame_stride_bytes is smaller than one compact stripe buffer");
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    for (uint32_t k = 0; k < n_frames; ++k)
        if ((rc = render_impl(ctx, spec, (uint8_t *)dev_pixels + (size_t)k * frame_stride_bytes, s)) != RXR_OK) return rc;
    return RXR_OK;
}

int rxr_profile_begin(rxr_ctx *ctx, uint32_t max_frames) {
    if (!ctx) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_profile_begin(rxr_member(ctx, 0), max_frames);  // kernel timing of a multi-device context: member 0's
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = rxr_synchronize(ctx);
    if (rc != RXR_OK) return rc;
    for (ProfSlot &p : ctx->prof)
        for (hipEvent_t e : p.ev)
            if (e) (void)hipEventDestroy(e);
    ctx->prof.clear();
    ctx->last_prof = nullptr;
    ctx->prof_next = 0;
    ctx->prof_calls = 0;
    if (max_frames > 65536) max_frames = 65536;
    ctx->prof.assign(max_frames, ProfSlot{});  // (the events of a slot are created when a render first needs them: rxr_launch.h)
    return RXR_OK;
}

int rxr_profile_stride(rxr_ctx *ctx, uint32_t stride) {
    if (!ctx || stride == 0) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_profile_stride(rxr_member(ctx, 0), stride);
    ctx->prof_stride = stride;
    ctx->prof_calls = 0;
    return RXR_OK;
}

int rxr_profile_read(rxr_ctx *ctx, float *setup_us, float *raster_us, uint32_t capacity, uint32_t *n_out) {
    if (!ctx || !n_out) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_profile_read(rxr_member(ctx, 0), setup_us, raster_us, capacity, n_out);
    int rc = rxr_synchronize(ctx);
    if (rc != RXR_OK) return rc;
    uint32_t n = (uint32_t)std::min<size_t>(std::min<size_t>(ctx->prof_next, ctx->prof.size()), capacity);
    for (uint32_t i = 0; i < n; ++i) {
        float a = 0, b = 0;
        if (!rxr_prof_slot_us(ctx->prof[i], &a, &b)) return fail(ctx, RXR_ERR_HIP, "rxr_profile_read: a kernel's start / stop events could not be read");
        if (setup_us) setup_us[i] = a;
        if (raster_us) raster_us[i] = b;
    }
    *n_out = n;
    ctx->prof_next = 0;
    return RXR_OK;
}

int rxr_render_spec(rxr_ctx *ctx, const RenderSpec &spec, void *dev_pixels, hipStream_t s) { return render_impl(ctx, spec, dev_pixels, s); }

// Waits for everything this context has queued and reports what the launches since the previous call left in the pinned
// status words.  Those words are sticky (the device only sets / raises them), so an overflow or a program fault of ANY
// launch since the last call is seen, not only the last one's:
//   - a bin list overflowed: the lists are grown; the LAST launch is rendered again (its output is then complete); if
//     earlier launches were queued in between (an asynchronous caller that does not synchronize per frame) their frames
//     were shipped incomplete and the call returns RXR_ERR_OVERFLOW to say so;
//   - a fragment's program faulted, or an opacity staircase dropped an entry: an error, see the messages.
int rxr_synchronize(rxr_ctx *ctx) {
    if (!ctx) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_group_synchronize(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bool earlier_incomplete = false;
    for (int attempt = 0; attempt < 4; ++attempt) {
        int qrc = rxr_quiesce(ctx);
        if (qrc != RXR_OK) return qrc;
        const uint32_t launches = ctx->launches_since_sync;
        ctx->launches_since_sync = 0;
        if (!ctx->rendered) return RXR_OK;
        uint32_t *hc = ctx->h_counters;
        if (hc[HS_VM_FAULT] == VMF_JIT_PALETTE_MISS) {
            // not a fault of the program: a compiled set met a palette slot without a colour, where the reference pushes nothing -- only
            // the interpreter's dynamic stack follows that.  The set runs interpreted from now on; the last launch is rendered again, an
            // earlier one of this batch of launches was incomplete (reported like an overflow).
            hc[HS_VM_FAULT] = 0;
            ctx->jit_palette_miss = true;
            ctx->jit_info = "not compiled: a PaletteIndex met a missing or empty palette slot (the interpreter follows the reference's shorter stack)";
            if (launches > 1u) earlier_incomplete = true;
            int rc = render_impl(ctx, ctx->last_spec, ctx->last_out, ctx->last_stream, true);
            if (rc != RXR_OK) return rc;
            continue;
        }
        if (hc[HS_VM_FAULT]) {
            // a fragment's program did what makes the reference panic (rxr_vm.h, VMF_*)
            static const char *const what[] = {"", "stack underflow", "stack overflow", "local index out of range", "global index out of range",
                                               "call depth", "loop depth", "instruction limit (runaway loop)", "clamp with min > max",
                                               "call of a missing function", "bad opcode", "too many locals"};
            uint32_t code = hc[HS_VM_FAULT];
            hc[HS_VM_FAULT] = 0;
            return fail(ctx, RXR_ERR_INVALID, std::string("shader program fault: ") + (code < sizeof(what) / sizeof(what[0]) ? what[code] : "?"));
        }
        if (hc[HS_BAD_LINE2D]) {
            hc[HS_BAD_LINE2D] = 0;
            return fail(ctx, RXR_ERR_UNSUPPORTED, "batch2d: line end point NaN or beyond +-2^30");
        }
        if (hc[HS_STAIRCASE]) {
            hc[HS_STAIRCASE] = 0;
            return fail(ctx, RXR_ERR_UNSUPPORTED,
                        "four or more groups of opacity batches nest as prefix minima in one pixel: the device keeps three per pixel (surface_id, "
                        "rasterizer.rs:314-357, :1044-1048) and had to drop one; the frame may differ from the reference");
        }
        ctx->stats.n_bin_entries = (ctx->last_had_prepass && !ctx->last_used_blockscan) ? hc[CNT_ENTRIES] : 0u;  // (k_blockscan does not count its entries)
        const bool over3d = hc[CNT_OVERFLOW] != 0, over2d = hc[CNT_WORDS + CNT_OVERFLOW] != 0;
        if (!over3d && !over2d) {
            float a = 0, b = 0;
            if (ctx->last_prof && rxr_prof_slot_us(*ctx->last_prof, &a, &b)) {
                ctx->stats.setup_us = a;
                ctx->stats.raster_us = b;
                ctx->stats.total_us = a + b;
            }
            if (earlier_incomplete)
                return fail(ctx, RXR_ERR_OVERFLOW,
                            "a bin list overflowed in a launch that was not the last one before this rxr_synchronize: that frame was "
                            "incomplete (the lists have been grown and the last launch rendered again)");
            return RXR_OK;
        }
        if (launches > 1u) earlier_incomplete = true;
        int rc;
        if (over2d && ctx->last_used_blockscan2d) {
            // a block of bins or a bin had more 2D primitives than k_blockscan2d keeps: count / scan / fill (and the per-tile sort) for this frame
            ctx->blockscan2d_off = true;
            ctx->blockscan2d_bad.add(ctx->P.n_prims2d, (size_t)ctx->P.tiles_x * ((ctx->P.height + RXR_TILE_H - 1) / RXR_TILE_H));
            hc[CNT_WORDS + CNT_OVERFLOW] = hc[CNT_WORDS + HS_MAX_ENTRIES] = 0;
        } else if (over2d) {
            const size_t seen = std::max(hc[CNT_WORDS + HS_MAX_ENTRIES], hc[CNT_WORDS + CNT_ENTRIES]);
            if ((rc = ensure(ctx, ctx->d_list2d, (seen + seen / 2 + 1024) * sizeof(uint32_t))) != RXR_OK) return rc;
            ctx->list2d_capacity = (uint32_t)std::min<size_t>(ctx->d_list2d.cap / sizeof(uint32_t), 0xFFFFFFF0u);
            ctx->P.bin2d_list = (uint32_t *)ctx->d_list2d.p;
            ctx->P.list2d_capacity = ctx->list2d_capacity;
            hc[CNT_WORDS + CNT_OVERFLOW] = hc[CNT_WORDS + HS_MAX_ENTRIES] = 0;
        }
        if (over3d && ctx->last_used_blockscan) {
            // a block of bins or a bin had more candidates than k_blockscan keeps: this frame takes the general pipeline
            ctx->blockscan_off = true;
            ctx->blockscan_bad.add(ctx->P.n_tris3d, (size_t)ctx->P.tiles_x * ((ctx->P.height + RXR_TILE_H - 1) / RXR_TILE_H));
            hc[CNT_OVERFLOW] = hc[HS_MAX_ENTRIES] = 0;
        } else if (over3d) {
            const size_t seen = std::max(hc[HS_MAX_ENTRIES], hc[CNT_ENTRIES]);
            if ((rc = ensure(ctx, ctx->d_list, (seen + seen / 2 + 1024) * sizeof(uint32_t))) != RXR_OK) return rc;
            ctx->list_capacity = (uint32_t)std::min<size_t>(ctx->d_list.cap / sizeof(uint32_t), 0xFFFFFFF0u);
            ctx->P.bin_list = (uint32_t *)ctx->d_list.p;
            ctx->P.list_capacity = ctx->list_capacity;
            hc[CNT_OVERFLOW] = hc[HS_MAX_ENTRIES] = 0;
        }
        // the same launch again, now with room (the streams are idle: see rxr_quiesce above)
        if ((rc = render_impl(ctx, ctx->last_spec, ctx->last_out, ctx->last_stream, true)) != RXR_OK) return rc;
    }
    return fail(ctx, RXR_ERR_OOM, "bin list kept overflowing");
}

int rxr_download_rows(rxr_ctx *ctx, uint8_t *pixels, uint32_t row0, uint32_t row1) {
    if (!ctx || !pixels) return RXR_ERR_INVALID;
    if (ctx->group) return fail(ctx, RXR_ERR_UNSUPPORTED, "rxr_download_rows on a multi-device context: use rxr_rasterize / rxr_render_download");
    if (!ctx->has_frame || row0 > row1 || row1 > ctx->P.height) return fail(ctx, RXR_ERR_INVALID, "rxr_download_rows: bad row range or no frame");
    int rc = rxr_synchronize(ctx);
    if (rc != RXR_OK) return rc;
    size_t off = (size_t)row0 * ctx->P.width * 4, bytes = (size_t)(row1 - row0) * ctx->P.width * 4;
    if (bytes) {
        HIPCHK(ctx, hipMemcpyAsync(pixels + off, (uint8_t *)ctx->d_fb.p + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RXR_OK;
}

int rxr_rasterize(rxr_ctx *ctx, const rxr_frame *frame, uint8_t *pixels) {
    if (!ctx || !pixels) return RXR_ERR_INVALID;
    int rc = rxr_upload_frame(ctx, frame);
    if (rc != RXR_OK) return rc;
    return rxr_render_download(ctx, pixels);
}

int rxr_render_download(rxr_ctx *ctx, uint8_t *pixels) {
    if (!ctx || !pixels) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_group_render_download(ctx, pixels);
    if (!ctx->has_frame) return fail(ctx, RXR_ERR_INVALID, "rxr_render_download: no frame uploaded");
    int rc = RXR_OK;
    // The download of a 4K frame over PCIe takes longer than rendering it, so frames of 4 Mpixel and more are rastered in bands of tile
    // rows and every finished band travels to the caller's buffer while the next ones render; rendering in bands is byte-identical to
    // one launch (tested).  Measured: 3840x2160 0.90 -> 0.76 ms per call; at 1920x1080 the extra launches cost more than the overlap
    // gains (0.26 -> 0.31 ms), hence the threshold.  Rounds 1-3 made four whole launch sequences of it and therefore served frames
    // without a pre-pass only; since round 4 ONE pre-pass (projection, set-up, bins) is followed by four raster launches over its bins
    // (render_impl, RasterParams.bin_row0), which serves binned and device-projected frames as well: the 8K frame of 1 M triangles
    // downloads 133 MB in 2.4 ms behind 0.6 ms of kernels that used to come first.  A list overflow is repaired by rxr_synchronize as
    // ever (lists grown, the whole frame rendered again); the frame is then downloaded once more.
    const RasterParams &P = ctx->P;
    const uint32_t H = P.height;
    static const bool no_pipeline = getenv("RXR_NO_DOWNLOAD_PIPELINE") != nullptr;  // A-B runs
    if (no_pipeline || (size_t)P.width * H < (1u << 22)) {
        static const bool timing = getenv("RXR_E2E_TIMING") != nullptr;  // diagnostics (tools/e2e_probe.py): where a large frame's call goes
        if (timing) {
            using clk = std::chrono::steady_clock;
            HIPCHK(ctx, hipSetDevice(ctx->device));
            const auto t0 = clk::now();
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the tail of the upload's host->device copies
            const auto t1 = clk::now();
            rc = rxr_render_rows(ctx, 0, H);
            if (rc != RXR_OK) return rc;
            rc = rxr_synchronize(ctx);
            if (rc != RXR_OK) return rc;
            const auto t2 = clk::now();
            rc = rxr_download_rows(ctx, pixels, 0, H);
            const auto t3 = clk::now();
            auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "rxr_e2e_timing h2d_tail_ms=%.3f kernels_ms=%.3f download_ms=%.3f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
            return rc;
        }
        rc = rxr_render_rows(ctx, 0, H);
        if (rc != RXR_OK) return rc;
        return rxr_download_rows(ctx, pixels, 0, H);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    constexpr uint32_t n_bands = 4;
    uint32_t row_of[n_bands + 1] = {};
    RenderSpec spec{};
    rc = band_spec(ctx, 0, H, false, spec);
    if (rc != RXR_OK) return rc;
    const uint32_t rerenders_before = ctx->rerenders;
    bool spans_back = false;
    rc = render_impl(ctx, spec, ctx->d_fb.p, ctx->stream, false, n_bands, ctx->ev_band, row_of, ctx->ev_band[7], &spans_back);
    if (rc != RXR_OK) return rc;
    // Rows outside the frame's content are the miss colour on the device AND need not cross PCIe: the caller's rows are written here,
    // by the host, while the device renders and the content rows travel (the 8K frame of the box grid: 40 of 133 MB that are not
    // downloaded; the link is what bounds this call).
    uint32_t c0 = 0, c1 = H;
    bool clipped = content_band(ctx, spec, c0, c1);
    if (spans_back) {
        // device-projected meshes: the content is known once the projection has run -- the device hands its row-span table back a
        // fraction of a millisecond into the frame (k_spans_from_meshes writes a copy into page-locked memory, an event behind it),
        // long before the first band is rastered; this thread would only wait for the bands otherwise
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_band[7]));
        const uint2 *back = ctx->h_row_spans + RXR_MAX_TILE_ROWS;
        const uint32_t n_rows = (H + RXR_TILE_H - 1u) / RXR_TILE_H;
        uint32_t r0 = n_rows, r1 = 0;
        for (uint32_t r = 0; r < n_rows; ++r)
            if (back[r].x < back[r].y) {
                r0 = std::min(r0, r);
                r1 = r + 1u;
            }
        const uint32_t d0 = r0 < r1 ? r0 * RXR_TILE_H : 0u, d1 = r0 < r1 ? std::min(r1 * (uint32_t)RXR_TILE_H, H) : 0u;
        if ((size_t)(d0 + (H - d1)) / RXR_TILE_H * P.tiles_x >= content_min_tiles()) {  // (as content_band: a few rows are not worth the fills)
            c0 = d0;
            c1 = d1;
            clipped = true;
        }
    }
    // ... and inside the content rows the columns outside the tile rows' spans (host-projected frames: the table of rxr_upload_frame;
    // device-projected ones: the table the device has just handed back) are the miss colour as well: each band travels as two strips of
    // hipMemcpy2DAsync over the union of its tile rows' spans, the host writes what lies to the left and right.  Strips, not tile rows:
    // a 2D copy costs about 13 us of its own (tools/microbench/copy2d.hip: 91 MB of whole rows 1.64 ms; the same trapezoid in 4 / 24 /
    // 93 strips 1.18 / 1.33 / 2.05 ms).
    struct Rect {
        uint32_t r0, r1, x0, x1;  // pixel rows [r0, r1), pixel columns [x0, x1)
    };
    std::vector<Rect> fills, copies;
    const uint32_t W = P.width;
    if (clipped) {
        fills.push_back({0u, c0, 0u, W});
        fills.push_back({c1, H, 0u, W});
    }
    // (the table says where anything CAN be drawn whether or not the launches above used it: a frame whose content reaches from the
    // first row to the last is not clamped in rows, but its row ends are still the miss colour)
    const uint2 *table = spans_back ? ctx->h_row_spans + RXR_MAX_TILE_ROWS : (ctx->spans_active && ctx->content_known ? ctx->h_row_spans : nullptr);
    if (getenv("RXR_NO_COLUMN_TRIM")) table = nullptr;  // A-B runs, tests (read per call)
    constexpr uint32_t n_sub = 2;
    size_t trimmed_px = 0;
    for (uint32_t k = 0; k < n_bands; ++k) {
        const uint32_t a = std::max(row_of[k], c0), b = std::min(row_of[k + 1], c1);
        if (b <= a) continue;
        const uint32_t ta = a / RXR_TILE_H, tb = (b + RXR_TILE_H - 1u) / RXR_TILE_H;   // tile rows of the band
        for (uint32_t j = 0; j < (table ? n_sub : 1u); ++j) {
            const uint32_t t0 = table ? ta + (tb - ta) * j / n_sub : ta, t1 = table ? ta + (tb - ta) * (j + 1u) / n_sub : tb;
            const uint32_t r0 = std::max(a, t0 * (uint32_t)RXR_TILE_H), r1 = std::min(b, t1 * (uint32_t)RXR_TILE_H);
            if (r1 <= r0) continue;
            uint32_t x0 = 0u, x1 = W;
            if (table) {
                uint32_t lo = 0xFFFFFFFFu, hi = 0u;
                for (uint32_t t = t0; t < t1 && t < RXR_MAX_TILE_ROWS; ++t)
                    if (table[t].x < table[t].y) {
                        lo = std::min(lo, table[t].x);
                        hi = std::max(hi, table[t].y);
                    }
                if (lo >= hi) {  // nothing in these rows at all
                    fills.push_back({r0, r1, 0u, W});
                    trimmed_px += (size_t)(r1 - r0) * W;
                    continue;
                }
                x0 = std::min(lo * (uint32_t)RXR_TILE_W, W);
                x1 = std::min(hi * (uint32_t)RXR_TILE_W, W);
                if ((size_t)(x1 - x0) * 10u >= (size_t)W * 9u) {  // (nearly the whole width: one contiguous copy is cheaper than a strided one)
                    x0 = 0u;
                    x1 = W;
                }
                trimmed_px += (size_t)(r1 - r0) * (W - (x1 - x0));
            }
            copies.push_back({r0, r1, x0, x1});
        }
    }
    if (table && trimmed_px < content_min_tiles() * (size_t)(RXR_TILE_W * RXR_TILE_H)) {  // (8 MB by default) not worth the strided copies and the fills: whole rows, as without a table
        copies.clear();
        fills.resize(clipped ? 2u : 0u);
        for (uint32_t k = 0; k < n_bands; ++k) {
            const uint32_t a = std::max(row_of[k], c0), b = std::min(row_of[k + 1], c1);
            if (b > a) copies.push_back({a, b, 0u, W});
        }
    }
    for (const Rect &c : copies)
        if (!(c.x0 == 0u && c.x1 == W)) {
            fills.push_back({c.r0, c.r1, 0u, c.x0});
            fills.push_back({c.r0, c.r1, c.x1, W});
        }
    // [0, 0, 0, 255] per pixel (:420-461), written by helper threads while the device renders and the link is busy (they start BEFORE the
    // copies are queued: a copy into pageable memory keeps the calling thread until it has landed); the caller's buffer may be unaligned:
    // bytes then.  Rows are dealt round robin.
    size_t n_px = 0;
    for (const Rect &f : fills) n_px += f.r1 > f.r0 && f.x1 > f.x0 ? (size_t)(f.r1 - f.r0) * (f.x1 - f.x0) : 0u;
    const uint32_t n_threads = n_px >= (12u << 20) ? 8u : (n_px >= (4u << 20) ? 4u : (n_px ? 1u : 0u));
    auto part = [&fills, pixels, W](uint32_t me, uint32_t of) {
        const bool aligned = ((uintptr_t)pixels & 3u) == 0u;
        uint32_t turn = 0;
        for (const Rect &f : fills) {
            if (f.r1 <= f.r0 || f.x1 <= f.x0) continue;
            for (uint32_t r = f.r0; r < f.r1; ++r, ++turn) {
                if (turn % of != me) continue;
                uint8_t *p = pixels + ((size_t)r * W + f.x0) * 4;
                const size_t n = f.x1 - f.x0;
                if (aligned) std::fill((uint32_t *)p, (uint32_t *)p + n, 0xFF000000u);
                else
                    for (size_t i = 0; i < n; ++i) { p[4 * i] = 0; p[4 * i + 1] = 0; p[4 * i + 2] = 0; p[4 * i + 3] = 255; }
            }
        }
    };
    ctx->last_download_bytes = 0;
    for (const Rect &c : copies) ctx->last_download_bytes += (uint64_t)(c.r1 - c.r0) * (c.x1 - c.x0) * 4u;
    ctx->last_host_fill_bytes = (uint64_t)n_px * 4u;
    std::vector<std::thread> helpers;
    helpers.reserve(n_threads);
    for (uint32_t t = 0; t < n_threads; ++t) {
        try {
            helpers.emplace_back(part, t, n_threads);
        } catch (...) {  // (no thread to be had: this share is written here and now -- nothing may leave through the C ABI)
            part(t, n_threads);
        }
    }
    struct JoinAll {  // (every return path below, errors included, waits for the helpers: they write the caller's buffer)
        std::vector<std::thread> &h;
        ~JoinAll() { for (std::thread &th : h) if (th.joinable()) th.join(); }
    } join_all{helpers};
    {
        size_t ci = 0;
        for (uint32_t k = 0; k < n_bands; ++k) {
            HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_band[k], 0));
            for (; ci < copies.size() && copies[ci].r0 < row_of[k + 1]; ++ci) {  // (ascending rows; every strip lies inside one band)
                const Rect &c = copies[ci];
                const size_t off = ((size_t)c.r0 * W + c.x0) * 4;
                if (c.x0 == 0u && c.x1 == W) {
                    HIPCHK(ctx, hipMemcpyAsync(pixels + off, (uint8_t *)ctx->d_fb.p + off, (size_t)(c.r1 - c.r0) * W * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
                } else {
                    HIPCHK(ctx, hipMemcpy2DAsync(pixels + off, (size_t)W * 4, (uint8_t *)ctx->d_fb.p + off, (size_t)W * 4, (size_t)(c.x1 - c.x0) * 4, c.r1 - c.r0,
                                                 hipMemcpyDeviceToHost, ctx->copy_stream));
                }
            }
        }
    }
    for (std::thread &th : helpers) th.join();  // (before anything below can write the same bytes again: the repaired frame's download)
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    rc = rxr_synchronize(ctx);  // (program faults and list overflows are reported / repaired here)
    if (rc != RXR_OK) return rc;
    if (ctx->rerenders != rerenders_before) return rxr_download_rows(ctx, pixels, 0, H);  // the bands that travelled were incomplete
    return RXR_OK;
}

int rxr_get_stats(rxr_ctx *ctx, rxr_stats *out) {
    if (!ctx || !out) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_group_get_stats(ctx, out);
    *out = ctx->stats;
    return RXR_OK;
}

void *rxr_device_framebuffer(rxr_ctx *ctx) { return (ctx && !ctx->group) ? ctx->d_fb.p : nullptr; }

// ---- Rusteria programs: NodeOp tree (include/rxr.h) -> jump code (rxr_device.h) ------------------
namespace {

struct Flattener {
    std::vector<uint32_t> code;
    std::vector<std::pair<size_t, uint32_t>> call_patches;  // (position of the target word, function index)
    std::vector<size_t> return_patches;                     // positions to fill with the current function's ENDFN address
    uint32_t n_functions = 0;
    bool writes_opacity = false, writes_emissive = false;
    std::string err;
    int status = RXR_OK;

    bool bad(int st, const std::string &m) {
        if (status == RXR_OK) {
            status = st;
            err = m;
        }
        return false;
    }

    // one block of the serialised tree; for_depth > 0 inside a For
    bool block(const uint32_t *w, size_t n, int for_depth, int depth) {
        if (depth > 64) return bad(RXR_ERR_INVALID, "program nested too deeply");
        size_t i = 0;
        auto need = [&](size_t k) { return i + k <= n; };
        while (i < n) {
            const uint32_t op = w[i++];
            if (op >= RXR_NODE_COUNT) return bad(RXR_ERR_INVALID, "unknown NodeOp opcode");
            switch (op) {
                case RXR_NODE_LOAD_GLOBAL:
                case RXR_NODE_STORE_GLOBAL:
                case RXR_NODE_LOAD_LOCAL:
                case RXR_NODE_STORE_LOCAL:
                    if (!need(1)) return bad(RXR_ERR_INVALID, "truncated program");
                    code.push_back(op);
                    code.push_back(w[i++]);
                    break;
                case RXR_NODE_GET_COMPONENTS:
                case RXR_NODE_SET_COMPONENTS: {
                    if (!need(1)) return bad(RXR_ERR_INVALID, "truncated program");
                    uint32_t k = w[i++];
                    if (!need(k)) return bad(RXR_ERR_INVALID, "truncated program");
                    if (k > 12) return bad(RXR_ERR_UNSUPPORTED, "swizzle with more than 12 components");
                    uint32_t enc = k;
                    for (uint32_t j = 0; j < k; ++j) enc |= (w[i + j] > 2u ? 3u : w[i + j]) << (4u + 2u * j);
                    i += k;
                    code.push_back(op == RXR_NODE_GET_COMPONENTS ? (uint32_t)VM_GETC : (uint32_t)VM_SETC);
                    code.push_back(enc);
                    break;
                }
                case RXR_NODE_IF: {
                    if (!need(3)) return bad(RXR_ERR_INVALID, "truncated program");
                    const uint32_t tl = w[i], he = w[i + 1], el = w[i + 2];
                    i += 3;
                    if (!need((size_t)tl + el)) return bad(RXR_ERR_INVALID, "truncated program");
                    code.push_back(VM_JZ);
                    const size_t jz = code.size();
                    code.push_back(0);
                    if (!block(w + i, tl, for_depth, depth + 1)) return false;
                    i += tl;
                    if (he) {
                        code.push_back(VM_JMP);
                        const size_t jend = code.size();
                        code.push_back(0);
                        code[jz] = (uint32_t)code.size();
                        if (!block(w + i, el, for_depth, depth + 1)) return false;
                        code[jend] = (uint32_t)code.size();
                    } else {
                        code[jz] = (uint32_t)code.size();
                    }
                    i += el;
                    break;
                }
                case RXR_NODE_FOR: {  // execution.rs:251-278
                    if (!need(4)) return bad(RXR_ERR_INVALID, "truncated program");
                    const uint32_t l[4] = {w[i], w[i + 1], w[i + 2], w[i + 3]};
                    i += 4;
                    if (!need((size_t)l[0] + l[1] + l[2] + l[3])) return bad(RXR_ERR_INVALID, "truncated program");
                    const uint32_t *init = w + i, *cond = init + l[0], *incr = cond + l[1], *body = incr + l[2];
                    i += (size_t)l[0] + l[1] + l[2] + l[3];
                    code.push_back(VM_FOR_ENTER);
                    if (!block(init, l[0], for_depth + 1, depth + 1)) return false;
                    code.push_back(VM_FOR_TRUNC);
                    const uint32_t top = (uint32_t)code.size();
                    if (!block(cond, l[1], for_depth + 1, depth + 1)) return false;
                    code.push_back(VM_FOR_COND);
                    const size_t jexit = code.size();
                    code.push_back(0);
                    code.push_back(VM_FOR_TRUNC);
                    if (!block(body, l[3], for_depth + 1, depth + 1)) return false;
                    code.push_back(VM_FOR_TRUNC);
                    if (!block(incr, l[2], for_depth + 1, depth + 1)) return false;
                    code.push_back(VM_FOR_TRUNC);
                    code.push_back(VM_JMP);
                    code.push_back(top);
                    code[jexit] = (uint32_t)code.size();
                    code.push_back(VM_FOR_EXIT);
                    break;
                }
                case RXR_NODE_PUSH:
                    if (!need(3)) return bad(RXR_ERR_INVALID, "truncated program");
                    // peephole: a constant followed by a component-wise binary operation becomes one instruction
                    // ("tos = tos op c"), the commonest pair in compiled expressions
                    if (i + 3 < n) {
                        const uint32_t nx = w[i + 3];
                        int fused = -1;
                        switch (nx) {
                            case RXR_NODE_ADD: fused = VM_BINC_ADD; break;
                            case RXR_NODE_SUB: fused = VM_BINC_SUB; break;
                            case RXR_NODE_MUL: fused = VM_BINC_MUL; break;
                            case RXR_NODE_DIV: fused = VM_BINC_DIV; break;
                            case RXR_NODE_MIN: fused = VM_BINC_MIN; break;
                            case RXR_NODE_MAX: fused = VM_BINC_MAX; break;
                            case RXR_NODE_MOD: fused = VM_BINC_MOD; break;
                            case RXR_NODE_LT: fused = VM_BINC_LT; break;
                            case RXR_NODE_LE: fused = VM_BINC_LE; break;
                            case RXR_NODE_GT: fused = VM_BINC_GT; break;
                            case RXR_NODE_GE: fused = VM_BINC_GE; break;
                            case RXR_NODE_EQ: fused = VM_BINC_EQ; break;
                            case RXR_NODE_NE: fused = VM_BINC_NE; break;
                            default: break;
                        }
                        if (fused >= 0) {
                            code.push_back((uint32_t)VM_BINC | ((uint32_t)fused << 8));
                            code.push_back(w[i]);
                            code.push_back(w[i + 1]);
                            code.push_back(w[i + 2]);
                            i += 4;
                            break;
                        }
                    }
                    code.push_back(op);
                    code.push_back(w[i]);
                    code.push_back(w[i + 1]);
                    code.push_back(w[i + 2]);
                    i += 3;
                    break;
                case RXR_NODE_FUNCTION_CALL: {
                    if (!need(3)) return bad(RXR_ERR_INVALID, "truncated program");
                    const uint32_t arity = w[i], total = w[i + 1], index = w[i + 2];
                    i += 3;
                    if (index >= n_functions) {  // program.user_functions[index] panics when reached
                        code.push_back(VM_FAULT);
                        code.push_back(VMF_BAD_CALL);
                        break;
                    }
                    if (total > RXR_VM_LOCALS) return bad(RXR_ERR_UNSUPPORTED, "function with more locals than the device VM holds");
                    code.push_back(VM_CALL);
                    code.push_back(arity);
                    code.push_back(total);
                    call_patches.emplace_back(code.size(), index);
                    code.push_back(0);
                    break;
                }
                case RXR_NODE_RETURN:
                    // the reference's For keeps iterating after a Return unwound its body (execution.rs:258-277):
                    // it pops the ENCLOSING frame's values as loop conditions
                    if (for_depth > 0) return bad(RXR_ERR_UNSUPPORTED, "Return inside For (the reference unwinds it incorrectly)");
                    code.push_back(VM_RETURN);
                    return_patches.push_back(code.size());
                    code.push_back(0);
                    break;
                case RXR_NODE_ALLOC:
                case RXR_NODE_ITERATE:
                case RXR_NODE_SAVE:
                    return bad(RXR_ERR_UNSUPPORTED, "Alloc / Iterate / Save (texture baking) are not part of per-fragment shading");
                case RXR_NODE_SET_EMISSIVE:
                    // `emissive` is never reset by the raster loops: every opaque 3D fragment adds whatever the LAST SetEmissive
                    // of its tile left (rasterizer.rs:1323, :1394).  Whether a frame can run such a program without that leak
                    // is decided per frame (rxr_upload_frame: every visible opaque 3D batch must assign it itself)
                    writes_emissive = true;
                    code.push_back(op);
                    break;
                case RXR_NODE_SET_OPACITY:
                    writes_opacity = true;
                    code.push_back(op);
                    break;
                default: code.push_back(op); break;
            }
        }
        return true;
    }
};

// Purity check (definite assignment).  The reference keeps ONE Execution per tile: `shade`'s locals are resized, not
// cleared (execution.rs:747), and globals are never reset, so a read that the same invocation has not written before
// sees the previous fragment's value.  A program is accepted only if every LoadLocal of `shade` and every LoadGlobal
// anywhere is definitely preceded by a store in the same invocation: stores count from their position onwards within a
// block, an If contributes what BOTH branches store, a For what its init and its first condition evaluation store;
// called functions get fresh zeroed locals (execution.rs:188) -- their global reads are checked against what is
// assigned at the call, their global stores are not credited.
struct Assigned {
    uint64_t locals = 0;
    uint32_t globals = 0;
    uint32_t fields = 0;  // PF_*: Execution fields this invocation has written so far
};


struct PurityCheck {
    const rxr_program &p;
    bool ok = true;
    uint32_t reads_unassigned = 0;  // PF_* read before this invocation wrote them
    uint32_t writes = 0;            // PF_* written anywhere in the program
    uint32_t exit_fields = ~0u;     // PF_* assigned at EVERY `Return` of shade itself (a Return leaves before the code behind it)
    std::vector<char> visiting;

    explicit PurityCheck(const rxr_program &prog) : p(prog), visiting(prog.n_functions, 0) {}

    // walks one block; `in_shade`: LoadLocal / StoreLocal refer to shade's (leaky) locals
    Assigned block(const uint32_t *w, size_t n, Assigned a, bool in_shade, int depth) {
        size_t i = 0;
        while (i < n && ok && depth < 64) {
            const uint32_t op = w[i++];
            switch (op) {
                case RXR_NODE_LOAD_LOCAL:
                    if (in_shade && (w[i] >= 64 || !((a.locals >> w[i]) & 1ull))) ok = false;
                    i += 1;
                    break;
                case RXR_NODE_STORE_LOCAL:
                    if (in_shade && w[i] < 64) a.locals |= 1ull << w[i];
                    i += 1;
                    break;
                case RXR_NODE_LOAD_GLOBAL:
                    if (w[i] >= 32 || !((a.globals >> w[i]) & 1u)) ok = false;
                    i += 1;
                    break;
                case RXR_NODE_STORE_GLOBAL:
                    if (w[i] < 32) a.globals |= 1u << w[i];  // (what a callee stores is not credited to its caller, see FunctionCall)
                    i += 1;
                    break;
                case RXR_NODE_UV: reads_unassigned |= PF_UV & ~a.fields; break;
                case RXR_NODE_ROUGHNESS: reads_unassigned |= PF_ROUGHNESS & ~a.fields; break;
                case RXR_NODE_METALLIC: reads_unassigned |= PF_METALLIC & ~a.fields; break;
                case RXR_NODE_OPACITY: reads_unassigned |= PF_OPACITY & ~a.fields; break;
                case RXR_NODE_BUMP: reads_unassigned |= PF_BUMP & ~a.fields; break;
                case RXR_NODE_NORMAL: reads_unassigned |= PF_NORMAL & ~a.fields; break;
                case RXR_NODE_HITPOINT: reads_unassigned |= PF_HITPOINT; break;
                case RXR_NODE_EMISSIVE: reads_unassigned |= PF_EMISSIVE & ~a.fields; break;
                case RXR_NODE_SET_EMISSIVE: a.fields |= PF_EMISSIVE; writes |= PF_EMISSIVE; break;
                case RXR_NODE_RETURN:
                    if (in_shade) exit_fields &= a.fields;
                    break;
                case RXR_NODE_SET_UV: a.fields |= PF_UV; writes |= PF_UV; break;
                case RXR_NODE_SET_ROUGHNESS: a.fields |= PF_ROUGHNESS; writes |= PF_ROUGHNESS; break;
                case RXR_NODE_SET_METALLIC: a.fields |= PF_METALLIC; writes |= PF_METALLIC; break;
                case RXR_NODE_SET_OPACITY: a.fields |= PF_OPACITY; writes |= PF_OPACITY; break;
                case RXR_NODE_SET_BUMP: a.fields |= PF_BUMP; writes |= PF_BUMP; break;
                case RXR_NODE_SET_NORMAL: a.fields |= PF_NORMAL; writes |= PF_NORMAL; break;
                case RXR_NODE_GET_COMPONENTS:
                case RXR_NODE_SET_COMPONENTS: i += 1 + w[i]; break;
                case RXR_NODE_PUSH: i += 3; break;
                case RXR_NODE_IF: {
                    const uint32_t tl = w[i], he = w[i + 1], el = w[i + 2];
                    i += 3;
                    Assigned t = block(w + i, tl, a, in_shade, depth + 1);
                    Assigned e = he ? block(w + i + tl, el, a, in_shade, depth + 1) : a;
                    a.locals = t.locals & e.locals;
                    a.globals = t.globals & e.globals;
                    a.fields = t.fields & e.fields;
                    i += (size_t)tl + el;
                    break;
                }
                case RXR_NODE_FOR: {
                    const uint32_t l0 = w[i], l1 = w[i + 1], l2 = w[i + 2], l3 = w[i + 3];
                    i += 4;
                    const uint32_t *init = w + i, *cond = init + l0, *incr = cond + l1, *body = incr + l2;
                    a = block(init, l0, a, in_shade, depth + 1);
                    a = block(cond, l1, a, in_shade, depth + 1);       // the condition runs at least once
                    Assigned b = block(body, l3, a, in_shade, depth + 1);
                    (void)block(incr, l2, b, in_shade, depth + 1);
                    i += (size_t)l0 + l1 + l2 + l3;
                    break;
                }
                case RXR_NODE_FUNCTION_CALL: {
                    const uint32_t index = w[i + 2];
                    i += 3;
                    if (index < p.n_functions && !visiting[index]) {
                        visiting[index] = 1;
                        Assigned callee;
                        callee.globals = a.globals;
                        callee.fields = a.fields;
                        (void)block(p.functions[index].words, p.functions[index].n_words, callee, false, depth + 1);
                        visiting[index] = 0;
                    }
                    break;
                }
                default: break;
            }
        }
        return a;
    }
};

// `assigned_at_exit`: the PF_* fields that `shade` has written itself on every path to its end (its last instruction or a Return)
bool program_is_pure(const rxr_program &p, uint32_t &reads_unassigned, uint32_t &writes, uint32_t &assigned_at_exit) {
    reads_unassigned = writes = assigned_at_exit = 0;
    if (p.shade_index < 0 || (uint32_t)p.shade_index >= p.n_functions) return true;
    PurityCheck c(p);
    c.visiting[p.shade_index] = 1;
    const Assigned end = c.block(p.functions[p.shade_index].words, p.functions[p.shade_index].n_words, Assigned{}, true, 0);
    reads_unassigned = c.reads_unassigned;
    writes = c.writes;
    assigned_at_exit = end.fields & c.exit_fields;
    return c.ok;
}

}  // namespace

namespace {

// validates every program of the set and flattens them into one code stream; no device involved.
// Returns RXR_OK or the status + message rxr_set_shaders / rxr_check_shaders report.
int flatten_programs(const rxr_shader_set *set, std::vector<uint32_t> &code, std::vector<DevProgram> &progs, std::vector<uint32_t> &field_reads,
                     std::string &err) {
    auto bad = [&](int st, const std::string &m) {
        err = m;
        return st;
    };
    if (set->n_programs && !set->programs) return bad(RXR_ERR_INVALID, "NULL program array");
    Flattener fl;
    uint32_t field_writes = 0;
    for (uint32_t pi = 0; pi < set->n_programs; ++pi) {
        const rxr_program &p = set->programs[pi];
        if (p.n_functions && !p.functions) return bad(RXR_ERR_INVALID, "NULL function array");
        for (uint32_t k = 0; k < p.n_functions; ++k)
            if (p.functions[k].n_words && !p.functions[k].words) return bad(RXR_ERR_INVALID, "NULL function body");
        DevProgram d{};
        d.shade_entry = 0xFFFFFFFFu;
        d.shade_locals = p.shade_locals;
        d.n_globals = p.n_globals;
        uint32_t ru = 0, wr = 0, at_exit = 0;
        if (p.shade_index >= 0) {
            if ((uint32_t)p.shade_index >= p.n_functions)  // program.user_functions[index] would panic on the first fragment
                return bad(RXR_ERR_INVALID, "shade_index out of range");
            if (p.n_globals > RXR_VM_GLOBALS) return bad(RXR_ERR_UNSUPPORTED, "more globals than the device VM holds");
            if (p.shade_locals > RXR_VM_LOCALS) return bad(RXR_ERR_UNSUPPORTED, "more locals than the device VM holds");
            // structural check first (lengths), so that the purity walk below cannot run off the arrays
            fl.n_functions = p.n_functions;
            fl.writes_opacity = fl.writes_emissive = false;
            fl.call_patches.clear();
            std::vector<uint32_t> entries(p.n_functions);
            for (uint32_t k = 0; k < p.n_functions; ++k) {
                entries[k] = (uint32_t)fl.code.size();
                fl.return_patches.clear();
                if (!fl.block(p.functions[k].words, p.functions[k].n_words, 0, 0)) return bad(fl.status, fl.err);
                const uint32_t endfn = (uint32_t)fl.code.size();
                fl.code.push_back(VM_ENDFN);
                for (size_t pos : fl.return_patches) fl.code[pos] = endfn;
            }
            for (auto &cp : fl.call_patches) fl.code[cp.first] = entries[cp.second];
            if (!program_is_pure(p, ru, wr, at_exit))
                return bad(RXR_ERR_UNSUPPORTED, "a local of `shade` or a global is read before this invocation wrote it (the reference would read the previous fragment's value)");
            field_writes |= wr;
            d.shade_entry = entries[p.shade_index];
            d.flags = (fl.writes_opacity ? PG_WRITES_OPACITY : 0u) | (fl.writes_emissive ? PG_WRITES_EMISSIVE : 0u) |
                      ((at_exit & PF_EMISSIVE) ? PG_ASSIGNS_EMISSIVE : 0u);
        }
        field_reads.push_back(ru);
        progs.push_back(d);
    }
    // uv.z, roughness.yz, metallic.yz, opacity.yz and bump are never assigned by the raster loops: once ANY program of the
    // set writes such a field, a read that its own invocation has not preceded by a write would see an earlier fragment's lanes
    for (uint32_t m : field_reads)
        if (m & field_writes & (PF_UV | PF_ROUGHNESS | PF_METALLIC | PF_OPACITY | PF_BUMP | PF_EMISSIVE))
            return bad(RXR_ERR_UNSUPPORTED, "a program reads uv / roughness / metallic / opacity / bump / emissive before writing it while a program of the set writes that field (lanes the raster loops never reset would leak between fragments)");
    for (int k = 0; k < 4; ++k) fl.code.push_back(VM_ENDFN);  // the interpreter reads one word ahead of every opcode
    if (fl.code.size() >= (1ull << 31)) return bad(RXR_ERR_INVALID, "programs too large");
    code = std::move(fl.code);
    return RXR_OK;
}

}  // namespace

// Static stack depths.  For a program without calls and without PaletteIndex (the one opcode that pushes or not depending on
// data) the depth of the value stack is a function of the program counter alone.  This pass proves it by abstract
// interpretation of the flat code (state = depth + the For-loop bases; every join must agree; every instruction must have its
// operands and its room) and writes the depth BEFORE each instruction into bits 16..23 of its opcode word.  When that works
// for every program of a set the interpreter runs with a wave-uniform stack pointer read from the code stream (rxr_vm.h,
// kernel k_raster_vm_s): the per-lane stack bookkeeping becomes scalar.  Returns false when any program stays dynamic.
static bool tag_static_depths(std::vector<uint32_t> &code, const std::vector<DevProgram> &progs) {
    struct State {
        int depth;
        std::vector<int> loops;
        bool operator==(const State &o) const { return depth == o.depth && loops == o.loops; }
    };
    std::vector<int> seen(code.size(), -1);          // index into states, per pc
    std::vector<State> states;
    std::vector<std::pair<uint32_t, State>> work;
    auto length_of = [](uint32_t op) -> uint32_t {
        switch (op) {
            case RXR_NODE_LOAD_GLOBAL: case RXR_NODE_STORE_GLOBAL: case RXR_NODE_LOAD_LOCAL: case RXR_NODE_STORE_LOCAL:
            case VM_GETC: case VM_SETC: case VM_JMP: case VM_JZ: case VM_FOR_COND: case VM_RETURN: case VM_FAULT: return 2;
            case RXR_NODE_PUSH: case VM_BINC: case VM_CALL: return 4;
            default: return 1;
        }
    };
    for (const DevProgram &p : progs) {
        if (p.shade_entry == 0xFFFFFFFFu) continue;
        work.clear();
        work.push_back({p.shade_entry, State{0, {}}});
        while (!work.empty()) {
            auto [pc, st] = work.back();
            work.pop_back();
            for (;;) {
                if (pc >= code.size()) return false;
                const uint32_t w = code[pc], op = w & 0xFFu;
                if (op == VM_ENDFN) break;  // end of `shade`: the stack is not looked at any more
                if (seen[pc] >= 0) {
                    if (!(states[(size_t)seen[pc]] == st)) return false;  // two paths arrive with different stacks
                    break;
                }
                seen[pc] = (int)states.size();
                states.push_back(st);
                if (st.depth < 0 || st.depth > 255) return false;
                const uint32_t len = length_of(op);
                if (pc + len > code.size()) return false;
                int need = 0, delta = 0;
                bool room = false;
                switch (op) {
                    case RXR_NODE_LOAD_GLOBAL: case RXR_NODE_LOAD_LOCAL: case RXR_NODE_PUSH:
                    case RXR_NODE_UV: case RXR_NODE_NORMAL: case RXR_NODE_HITPOINT: case RXR_NODE_TIME: case RXR_NODE_COLOR:
                    case RXR_NODE_ROUGHNESS: case RXR_NODE_METALLIC: case RXR_NODE_EMISSIVE: case RXR_NODE_OPACITY: case RXR_NODE_BUMP:
                        room = true; delta = 1; break;
                    case RXR_NODE_STORE_GLOBAL: case RXR_NODE_STORE_LOCAL: case RXR_NODE_PRINT:
                    case RXR_NODE_SET_UV: case RXR_NODE_SET_NORMAL: case RXR_NODE_SET_COLOR: case RXR_NODE_SET_ROUGHNESS:
                    case RXR_NODE_SET_METALLIC: case RXR_NODE_SET_OPACITY: case RXR_NODE_SET_BUMP: case RXR_NODE_SET_EMISSIVE:
                    case VM_JZ: case VM_FOR_COND:
                        need = 1; delta = -1; break;
                    case RXR_NODE_SWAP: need = 2; break;
                    case VM_GETC: case VM_BINC:
                    case RXR_NODE_LENGTH: case RXR_NODE_LENGTH2: case RXR_NODE_LENGTH3: case RXR_NODE_ABS: case RXR_NODE_SIN: case RXR_NODE_SIN1:
                    case RXR_NODE_SIN2: case RXR_NODE_COS: case RXR_NODE_COS1: case RXR_NODE_COS2: case RXR_NODE_TAN: case RXR_NODE_ATAN:
                    case RXR_NODE_NORMALIZE: case RXR_NODE_FLOOR: case RXR_NODE_CEIL: case RXR_NODE_ROUND: case RXR_NODE_FRACT:
                    case RXR_NODE_DEGREES: case RXR_NODE_RADIANS: case RXR_NODE_SQRT: case RXR_NODE_LOG: case RXR_NODE_NOT: case RXR_NODE_NEG:
                        need = 1; break;
                    case VM_SETC: case RXR_NODE_PACK2:
                    case RXR_NODE_ADD: case RXR_NODE_SUB: case RXR_NODE_MUL: case RXR_NODE_DIV: case RXR_NODE_ATAN2: case RXR_NODE_ROTATE2D:
                    case RXR_NODE_DOT: case RXR_NODE_DOT2: case RXR_NODE_DOT3: case RXR_NODE_CROSS: case RXR_NODE_MOD: case RXR_NODE_MIN:
                    case RXR_NODE_MAX: case RXR_NODE_STEP: case RXR_NODE_POW: case RXR_NODE_EQ: case RXR_NODE_NE: case RXR_NODE_LT:
                    case RXR_NODE_LE: case RXR_NODE_GT: case RXR_NODE_GE: case RXR_NODE_AND: case RXR_NODE_OR:
                    case RXR_NODE_SAMPLE: case RXR_NODE_SAMPLE_NORMAL:
                        need = 2; delta = -1; break;
                    case RXR_NODE_PACK3: case RXR_NODE_MIX: case RXR_NODE_SMOOTHSTEP: case RXR_NODE_CLAMP:
                        need = 3; delta = -2; break;
                    case RXR_NODE_CLEAR: delta = st.depth > 0 ? -1 : 0; break;
                    case RXR_NODE_DUP:
                        if (st.depth > 0) { room = true; delta = 1; }
                        break;
                    case VM_RETURN: delta = st.depth > 0 ? -1 : 0; break;   // (no calls: the function is `shade`)
                    case VM_JMP: case VM_FOR_ENTER: case VM_FOR_TRUNC: case VM_FOR_EXIT: case VM_FAULT: break;
                    default: return false;   // VM_CALL, PaletteIndex, anything unknown: dynamic
                }
                if (st.depth < need) return false;                       // a stack underflow must be reported by the dynamic interpreter
                if (room && st.depth >= (int)RXR_VM_STACK) return false;  // ... and so must an overflow
                code[pc] = (w & 0xFF00FFFFu) | ((uint32_t)st.depth << 16);
                State nx = st;
                nx.depth += delta;
                if (op == VM_FOR_ENTER) {
                    if (nx.loops.size() >= RXR_VM_LOOPS) return false;
                    nx.loops.push_back(st.depth);
                } else if (op == VM_FOR_TRUNC) {
                    if (nx.loops.empty()) return false;
                    nx.depth = std::min(nx.depth, nx.loops.back());
                } else if (op == VM_FOR_EXIT) {
                    if (nx.loops.empty()) return false;
                    nx.loops.pop_back();
                }
                if (op == VM_FAULT) break;                                // the lane stops here
                if (op == VM_JMP || op == VM_RETURN) {
                    pc = code[pc + 1];
                    st = nx;
                    continue;
                }
                if (op == VM_JZ || op == VM_FOR_COND) work.push_back({code[pc + 1], nx});
                pc += len;
                st = nx;
            }
        }
    }
    return true;
}

int rxr_check_shaders(const rxr_shader_set *set, uint32_t *code_words, char *message, uint32_t message_capacity) {
    std::vector<uint32_t> code, reads;
    std::vector<DevProgram> progs;
    std::string err;
    int rc = set ? flatten_programs(set, code, progs, reads, err) : RXR_ERR_INVALID;
    if (code_words) *code_words = (uint32_t)code.size();
    if (message && message_capacity) {
        size_t n = std::min<size_t>(err.size(), message_capacity - 1);
        memcpy(message, err.data(), n);
        message[n] = 0;
    }
    return rc;
}

int rxr_set_shaders(rxr_ctx *ctx, const rxr_shader_set *set) {
    if (!ctx) return RXR_ERR_INVALID;
    if (ctx->group) return rxr_group_set_shaders(ctx, set);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        int qrc = rxr_quiesce(ctx);  // a render (possibly on the caller's stream) may still run the old programs
        if (qrc != RXR_OK) return qrc;
    }
    ctx->has_frame = false;  // the resident frame's batch headers refer to the old programs
    ctx->programs.clear();
    ctx->programs_static = false;
    ctx->program_field_reads.clear();
    ctx->n_patterns = ctx->n_normal_patterns = ctx->n_palette = 0;
    if (!set) return RXR_OK;
    if ((set->n_programs && !set->programs) || (set->n_patterns && !set->patterns) || (set->n_normal_patterns && !set->normal_patterns) ||
        (set->n_palette && !set->palette_rgb))
        return fail(ctx, RXR_ERR_INVALID, "rxr_set_shaders: NULL array");

    // ---- validate + flatten every program into one code stream
    struct {
        std::vector<uint32_t> code;
    } fl;
    std::vector<DevProgram> progs;
    std::vector<uint32_t> field_reads;  // per program: PF_* read before written
    {
        std::string err;
        int frc = flatten_programs(set, fl.code, progs, field_reads, err);
        if (frc != RXR_OK) return fail(ctx, frc, "rxr_set_shaders: " + err);
    }
    {
        const bool no_static = getenv("RXR_VM_NO_STATIC") != nullptr;  // A-B runs / tests: always the dynamic interpreter
        std::vector<uint32_t> tagged = fl.code;
        ctx->programs_static = !no_static && !progs.empty() && tag_static_depths(tagged, progs);
        if (ctx->programs_static) fl.code.swap(tagged);   // (a failed attempt leaves partial tags behind: keep the clean stream then)
    }

    // ---- patterns + palette
    std::vector<DevPattern> pats;
    size_t n_floats = 0;
    auto add_patterns = [&](const rxr_pattern *src, uint32_t n) -> int {
        for (uint32_t i = 0; i < n; ++i) {
            if (!src[i].rgb || src[i].width == 0 || src[i].height == 0 || src[i].width > 32768 || src[i].height > 32768) return RXR_ERR_INVALID;
            DevPattern d{};
            d.offset = (uint32_t)n_floats;
            d.w = src[i].width;
            d.h = src[i].height;
            pats.push_back(d);
            n_floats += (size_t)3 * d.w * d.h;
            if (n_floats >= (1ull << 31)) return RXR_ERR_INVALID;
        }
        return RXR_OK;
    };
    int rc;
    if ((rc = add_patterns(set->patterns, set->n_patterns)) != RXR_OK) return fail(ctx, rc, "rxr_set_shaders: bad pattern");
    if ((rc = add_patterns(set->normal_patterns, set->n_normal_patterns)) != RXR_OK) return fail(ctx, rc, "rxr_set_shaders: bad normal pattern");

    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        int r = ensure(ctx, b, bytes ? bytes : 16);
        if (r != RXR_OK) return r;
        if (bytes) HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return RXR_OK;
    };
    if ((rc = up(ctx->d_vm_code, fl.code.data(), fl.code.size() * 4)) != RXR_OK) return rc;
    if ((rc = up(ctx->d_programs, progs.data(), progs.size() * sizeof(DevProgram))) != RXR_OK) return rc;
    if ((rc = up(ctx->d_patterns, pats.data(), pats.size() * sizeof(DevPattern))) != RXR_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_pattern_data, n_floats ? n_floats * 4 : 16)) != RXR_OK) return rc;
    {
        size_t k = 0;
        auto copy_pats = [&](const rxr_pattern *src, uint32_t n) -> int {
            for (uint32_t i = 0; i < n; ++i, ++k)
                HIPCHK(ctx, hipMemcpyAsync((float *)ctx->d_pattern_data.p + pats[k].offset, src[i].rgb, (size_t)12 * pats[k].w * pats[k].h,
                                           hipMemcpyHostToDevice, ctx->stream));
            return RXR_OK;
        };
        if ((rc = copy_pats(set->patterns, set->n_patterns)) != RXR_OK) return rc;
        if ((rc = copy_pats(set->normal_patterns, set->n_normal_patterns)) != RXR_OK) return rc;
    }
    std::vector<float> pal((size_t)set->n_palette * 4);
    for (uint32_t i = 0; i < set->n_palette; ++i) {
        pal[4 * i] = set->palette_rgb[3 * i];
        pal[4 * i + 1] = set->palette_rgb[3 * i + 1];
        pal[4 * i + 2] = set->palette_rgb[3 * i + 2];
        pal[4 * i + 3] = (!set->palette_present || set->palette_present[i]) ? 1.0f : 0.0f;
    }
    if ((rc = up(ctx->d_palette, pal.data(), pal.size() * 4)) != RXR_OK) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the staging vectors die here
    // opt-in: the set as straight-line kernels compiled now (rxr_jit.hip); sets with calls or PaletteIndex keep the interpreter
    rxr_jit_drop(ctx);
    ctx->jit_info.clear();
    {
        // RXR_SHADER_JIT: unset / "async" -- a child process compiles while the interpreter renders (the default); "1" -- compiled when
        // first needed, the caller waits (measurements); "0" -- interpreted only
        const char *jit = getenv("RXR_SHADER_JIT");
        const char mode = jit ? jit[0] : 'a';
        ctx->jit_async = mode == 'a';
        if ((mode == '1' || mode == 'a') && !progs.empty()) {
            if ((rc = rxr_jit_build(ctx, fl.code, progs)) != RXR_OK) return rc;  // (does its own analysis: calls are covered, PaletteIndex / recursion are not)
        }
    }
    ctx->programs = std::move(progs);
    ctx->program_field_reads = std::move(field_reads);
    ctx->n_patterns = set->n_patterns;
    ctx->n_normal_patterns = set->n_normal_patterns;
    ctx->n_palette = set->n_palette;
    return RXR_OK;
}

int rxr_selftest_math(rxr_ctx *ctx, uint64_t n_tuples, uint64_t seed, uint64_t mismatches[RXR_MATH_KINDS]) {
    if (!ctx || !mismatches) return fail(ctx, RXR_ERR_INVALID, "rxr_selftest_math: NULL argument");
    if (ctx->group) return rxr_selftest_math(rxr_member(ctx, 0), n_tuples, seed, mismatches);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint32_t iters = 256;
    uint64_t per_block = 256ull * iters;
    uint64_t blocks64 = (n_tuples + per_block - 1) / per_block;
    if (blocks64 == 0) blocks64 = 1;
    if (blocks64 > (1ull << 22)) return fail(ctx, RXR_ERR_INVALID, "rxr_selftest_math: n_tuples too large (max 2^38)");
    unsigned long long *d = nullptr;
    HIPCHK(ctx, hipMalloc(&d, RXR_MATH_KINDS * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, RXR_MATH_KINDS * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        rxr_launch_selftest_math(seed, (uint32_t)blocks64, iters, d, ctx->stream);
        e = hipGetLastError();
    }
    unsigned long long h[RXR_MATH_KINDS] = {0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, RXR_ERR_HIP, std::string("rxr_selftest_math: ") + hipGetErrorString(e));
    for (int k = 0; k < RXR_MATH_KINDS; ++k) mismatches[k] = h[k];
    return RXR_OK;
}

}  // extern "C"

// what the run-time compiler did with the last program set of this context (rxr_jit.hip); "" when it was not asked
// tests: how the resident frame's 3D arrays arrived -- 0 plain rxr_upload_frame, 1 streamed and copied, 2 streamed out of page-locked memory
// tests: what rxr_upload_frame found out about the resident frame's extent: out[0] = content rows known, out[1], out[2] = the rows, out[3] = row spans in use (1: from the host's boxes, 2: completed on the device)
extern "C" int rxr_debug_content(rxr_ctx *ctx, uint32_t *out) {
    if (!ctx || ctx->group || !out) return -1;
    out[0] = ctx->content_known ? 1u : 0u;
    out[1] = ctx->content_row0;
    out[2] = ctx->content_row1;
    out[3] = ctx->spans_active ? 1u : (ctx->dev_spans ? 2u : 0u);  // (2: completed on the device from the meshes' boxes)
    return 0;
}
// tests (host only, no device): rxr_ref_tile_span (quick == 0) / rxr_ref_tile_span_quick of n boxes [lo, lo + extent] on one axis; out: p0, p1 per box
extern "C" void rxr_debug_tile_spans(const float *lo, const float *extent, uint32_t n, uint32_t size, uint32_t ts, float pad, int quick, uint32_t *out) {
    for (uint32_t i = 0; i < n; ++i) {
        if (quick) rxr_ref_tile_span_quick(lo[i], extent[i], size, ts, pad, out[2 * i], out[2 * i + 1]);
        else rxr_ref_tile_span(lo[i], extent[i], size, ts, pad, out[2 * i], out[2 * i + 1]);
    }
}
// tests: what the last banded rxr_render_download sent over PCIe and what the host wrote itself (bytes)
extern "C" int rxr_debug_download_bytes(rxr_ctx *ctx, uint64_t *out2) {
    if (!ctx || ctx->group || !out2) return -1;
    out2[0] = ctx->last_download_bytes;
    out2[1] = ctx->last_host_fill_bytes;
    return 0;
}
extern "C" int rxr_debug_stream_info(rxr_ctx *ctx) { return (ctx && !ctx->group) ? ctx->last_upload_streamed : -1; }

// tests: how many launch sequences rxr_synchronize has rendered again after a list overflow (a plain context or a member)
extern "C" uint32_t rxr_debug_rerenders(rxr_ctx *ctx) { return (ctx && !ctx->group) ? ctx->rerenders : 0u; }

extern "C" const char *rxr_debug_jit_info(rxr_ctx *ctx) {
    if (!ctx) return "";
    if (ctx->group) return rxr_member(ctx, 0) ? rxr_member(ctx, 0)->jit_info.c_str() : "";
    return ctx->jit_info.c_str();
}

// device-free half of the run-time compiler, for tests without a GPU: validates + flattens the set like rxr_check_shaders, generates
// the C++ of its programs (copied to `source`, truncated to its capacity) and, with compile != 0, runs hiprtc for gfx950.
// Returns RXR_OK, the validation status, or RXR_ERR_UNSUPPORTED with the reason in `message` when the set is not covered.
extern "C" int rxr_debug_jit_generate(const rxr_shader_set *set, int compile, char *source, uint32_t source_capacity, char *message, uint32_t message_capacity) {
    std::vector<uint32_t> code, reads;
    std::vector<DevProgram> progs;
    std::string err, gen;
    auto say = [&](const std::string &m) {
        if (message && message_capacity) snprintf(message, message_capacity, "%s", m.c_str());
    };
    say("");
    int rc = set ? flatten_programs(set, code, progs, reads, err) : RXR_ERR_INVALID;
    if (rc != RXR_OK) {
        say(err);
        return rc;
    }
    if (progs.empty()) {
        say("no programs");
        return RXR_ERR_UNSUPPORTED;
    }
    if (!rxr_jit_generate(code, progs, gen, err)) {
        say(err);
        return RXR_ERR_UNSUPPORTED;
    }
    if (source && source_capacity) snprintf(source, source_capacity, "%s", gen.c_str());
    if (compile) {
        std::vector<char> obj;
        double seconds = 0.0;
        double total = 0.0;
        size_t bytes = 0;
        for (int level : {2, 7, 8}) {  // (the three template levels rxr_jit_launch may ask for)
            if (!rxr_jit_compile(gen, "gfx950", level, obj, seconds, err)) {
                say(err);
                return RXR_ERR_HIP;
            }
            total += seconds;
            bytes += obj.size();
        }
        seconds = total;
        char m[96];
        snprintf(m, sizeof m, "compiled in %.2f s, %zu bytes", seconds, bytes);
        say(m);
    }
    return RXR_OK;
}
