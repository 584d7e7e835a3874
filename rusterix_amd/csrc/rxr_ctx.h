// rxr_ctx.h -- the context object behind the C ABI (private to rxr_api.hip and rxr_multi.hip).
//
// A plain context is one HIP device: its streams, the resident textures / meshes / programs, the frame blob and the
// device scratch.  A multi-device context (rxr_create_multi) is a handle whose `group` lists one plain context per
// device plus one host worker thread per member; it owns no device memory itself (rxr_multi.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rxr_device.h"
#include "rxr_launch.h"
#include "rxr_project.h"

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};
#define RXR_MAX_TILE_ROWS 2048u   // frames of at most 32768 rows

struct TileRange {
    uint32_t first, n;
};

// host-side record of one registered mesh (rxr_set_meshes)
struct HostMesh {
    DevMesh dev;              // static part (bases, counts, cull mode); view_model / rejected are per frame
    float transform[16];
    float aabb_lo[3], aabb_hi[3];
    bool has_vertices;
    uint32_t repeat_mode;
    rxr_source source;
    float ambient[3];
    int32_t shader;
    uint32_t has_profile_id, profile_id, list;
    int32_t chunk;
};

// one sampled render of the rxr_profile_begin ring: a (start, stop) event pair per launched kernel (rxr_launch.h)
using ProfSlot = LaunchTimes;

// one render launch sequence.  Band mode: rows [row0,row1), stride 1.  Stripe mode: every `stride`-th
// 16-row stripe from `first`, into a compact buffer (compact) or at its place in a whole frame (!compact).
struct RenderSpec {
    uint32_t row0, row1;
    uint32_t tile_y0, tile_stride, tiles_y;
    bool compact, external;
};

// pinned host words the device writes (rxr_ctx.h_counters): [0, CNT_WORDS) 3D bins, [CNT_WORDS, 2 CNT_WORDS) 2D bins,
// then the words below.  The overflow flags and the maxima are STICKY: the device only ever sets / raises them, the host
// clears them in rxr_synchronize after both streams have drained.
enum : uint32_t {
    HS_VM_FAULT = 2 * CNT_WORDS,        // a non-zero VMF_* code if any fragment's program faulted
    HS_STAIRCASE = 2 * CNT_WORDS + 1,   // != 0: a pixel's opacity staircase (rxr_kernels.hip front_insert) had to drop an entry
    HS_BAD_LINE2D = 2 * CNT_WORDS + 2,  // != 0: device-projected 2D batches: a visible segment's end point lies beyond +-2^30 (k_proj2d_prims)
    HS_WORDS = 2 * CNT_WORDS + 4,
};
// inside a set of CNT_WORDS host words: CNT_ENTRIES = entries of the LAST launch, CNT_OVERFLOW = sticky flag,
// CNT_LARGE (unused by the host otherwise) = the largest entry count seen by a launch that overflowed
#define HS_MAX_ENTRIES CNT_LARGE

struct rxr_group;

// Streaming hand-over of a large frame's projected 3D batches (rxr_stream_begin / rxr_stream_batch3d, include/rxr.h): the host
// hands batch i over from whatever thread projected it; batches are RETIRED in index order (which fixes their dense offsets in
// the vertex / triangle pools, i.e. the submission order the depth tie-break needs), copied into the pinned staging blob by the
// retiring thread and shipped to the device per group of consecutive batches -- all while later batches are still being projected.
struct FrameStream {
    bool active = false;
    uint32_t n = 0;
    struct Rec {
        const float *pv, *uv, *nrm;
        const uint32_t *idx;
        const void *edges;        // rxr_edges records, or (S.edgeless) the uint32 `visible` words
        uint32_t nv, nt;
        size_t v0, t0;
        const void *dev[5];  // pinned mode: the DEVICE addresses of pv, uv, nrm, idx, edges (hipPointerGetAttributes: for memory registered with
                             // hipHostRegister they need not equal the host addresses; the pull kernel reads through these)
    };
    std::vector<Rec> rec;
    std::vector<uint32_t> cap_v, cap_t;
    std::unique_ptr<std::atomic<uint8_t>[]> done;
    std::unique_ptr<std::atomic<uint32_t>[]> group_left;
    uint32_t group_size = 1, n_groups = 0;
    std::mutex mu;       // retirement: next / vcur / tcur
    std::mutex ship_mu;  // the HIP calls of a group's transfers
    uint32_t next = 0;                      // batches [0, next) are retired (under mu)
    std::atomic<uint32_t> retired{0};       // ... the same, readable without the lock
    std::atomic<uint32_t> copy_next{0};     // copy mode: batches [0, copy_next) have been claimed for their copy into pinned memory
    size_t vcur = 0, tcur = 0;
    size_t total_cap_v = 0, total_cap_t = 0;
    size_t off_pv = 0, off_uv = 0, off_nrm = 0, off_idx = 0, off_edges = 0, off_after = 0;  // the blob's layout up to the end of the edges
    size_t blob_capacity = 0;  // staging / device blob bytes that exist without a reallocation
    // pinned mode (rxr_stream_begin_pinned): no host copy; per group a table of (source, destination, bytes) in pinned memory and one
    // k_gather_host launch that pulls the arrays over PCIe
    bool pinned = false;
    struct GatherEntry {
        const void *src;
        uint64_t dst_off;       // bytes from the blob's start
        uint32_t bytes, first_piece;
    };
    GatherEntry *table = nullptr;   // pinned, 5 entries per batch
    size_t table_cap = 0;
    std::atomic<uint32_t> handed{0};
    std::atomic<int> failed{0};
    std::atomic<int> edgeless{-1};  // -1 until the first batch with triangles arrives; then 1: batches come without Edges records (ABI 5), 0: with
    size_t tri5() const { return edgeless.load() == 1 ? sizeof(uint32_t) : sizeof(rxr_edges); }  // bytes per triangle in the fifth pool
    std::string err;     // (under mu)
};

struct rxr_ctx {
    int device = 0;
    rxr_group *group = nullptr;          // != nullptr: multi-device handle (every other member below is unused)
    hipStream_t stream = nullptr;
    uint32_t prof_stride = 1, prof_calls = 0;  // rxr_profile_stride: every prof_stride-th render records events
    hipStream_t copy_stream = nullptr;   // rxr_rasterize: downloads of finished bands overlap the rendering of the next ones
    hipEvent_t ev_band[8] = {};
    std::string err;

    // textures
    DevBuf d_tex, d_texels;
    std::vector<DevTexDesc> h_tex;
    std::vector<TileRange> tiles_static, tiles_dynamic;

    // frame blob
    void *h_stage = nullptr;
    size_t h_stage_cap = 0;
    DevBuf d_frame;
    size_t last_blob_tail = 0;       // bytes of the last frame blob behind the projected arrays (rxr_stream_begin leaves room for twice that)
    DevBuf d_tri_setup, d_tri_shade, d_tri_box, d_bin_count, d_bins, d_list, d_large, d_counters, d_fb;
    DevBuf d_bin2d_count, d_bins2d, d_list2d, d_large2d;
    DevBuf d_stripes;                // multi-device member: this device's stripes, compact (rxr_multi.hip)
    uint32_t list2d_capacity = 0, parity2d = 0;
    uint32_t *h_counters = nullptr;  // pinned, HS_WORDS; written by the device through d_host_status
    uint32_t *d_host_status = nullptr;
    uint32_t list_capacity = 0;
    size_t list_floor = 0;           // RXR_LIST_CAPACITY_FLOOR (tests): initial size of the bin lists instead of the generous default
    uint32_t parity = 0;             // counter set of the next launch
    bool scratch_dirty = false;      // a pre-pass was queued without its raster launch
    bool scratch2d_dirty = false;
    uint32_t min_kernel_level = 0;   // RXR_MIN_KERNEL_LEVEL (tuning)
    // run-time compiled kernels of the current program set (rxr_jit.hip; opt-in RXR_SHADER_JIT=1), else null: the interpreter runs
    // (one module per template level 2 / 7 / 8, compiled when the first frame that needs it is launched; jit_source: the generated
    // programs of the current set, empty when the set is not covered)
    void *jit_module[3] = {nullptr, nullptr, nullptr}, *jit_fn[3] = {nullptr, nullptr, nullptr};
    void *jit_fn_cut[3] = {nullptr, nullptr, nullptr};   // k_raster_jit_cut of the same module (levels 7 / 8), or null
    bool jit_failed[3] = {false, false, false};
    bool jit_palette_miss = false;   // a compiled frame met a PaletteIndex without a colour: this set runs interpreted from now on (VMF_JIT_PALETTE_MISS)
    // background mode (the default): the compilation of a level runs in a child process (rxr_jitc); the interpreter renders until it is done
    bool jit_async = false;
    std::string jit_wait_key[3];           // the background compilation (rxr_jit.hip registry) this context is attached to, per level
    std::string jit_source, jit_arch;
    // mid-sized scenes: bin lists by k_blockscan (rxr_device.h RXR_BLOCKSCAN_*).  blockscan_off: the current frame overflowed a block or a
    // bin and goes through the general pipeline (reset by the next upload); RXR_BLOCKSCAN=0 turns the mode off, RXR_BLOCKSCAN_CAP sets
    // the slots per bin (tests)
    bool blockscan_enabled = true, blockscan_off = false, last_used_blockscan = false;
    // the (primitives, bins) of the last frames that overflowed k_blockscan / k_blockscan2d (a ring of eight each): frames of such a shape
    // do not try again -- a caller that uploads every frame would pay the failed attempt and the second rendering every time
    struct BadShapes {
        size_t prims[8] = {}, bins[8] = {};
        uint32_t next = 0;
        bool has(size_t p, size_t b) const {
            for (int i = 0; i < 8; ++i)
                if (prims[i] == p && bins[i] == b && p) return true;
            return false;
        }
        void add(size_t p, size_t b) {
            if (has(p, b)) return;
            prims[next % 8u] = p;
            bins[next % 8u] = b;
            ++next;
        }
    } blockscan_bad, blockscan2d_bad;
    bool blockscan2d_off = false, last_used_blockscan2d = false;  // the same for the 2D bins (k_blockscan2d); RXR_BLOCKSCAN2D=0 turns it off
    uint32_t blockscan_cap = 0;           // RXR_BLOCKSCAN_CAP in effect
    bool relaxed_lights = true;           // rxr_set_light_math / RXR_LIGHT_MATH: the 3D light loop in relaxed arithmetic (RasterParams.relaxed_lights)
    bool frame_needs_chunk_paths = true;  // the uploaded frame uses what feature level 1 adds (terrain / baked textures / staircase / editor paths)
    std::string jit_info;                // what happened to the last set ("compiled: ...", "not compiled: <why>", empty: not asked)
    bool programs_static = false;    // every program of the set has a stack depth that is a function of the pc (tag_static_depths)
    uint32_t small_mode = 2;         // RasterParams.fused_small for frames with <= RXR_STAGE_TRIS triangles;
                                     // RXR_SMALL_MODE=0|1|2 overrides it (tests / A-B runs)

    // device-side projection (rxr_set_meshes)
    std::vector<HostMesh> meshes;
    DevBuf d_obj, d_proj_out, d_proj_misc;
    ProjectParams PP{};
    size_t mesh_verts_out = 0, mesh_tris_out = 0;
    size_t pp_off_meshes = 0;  // byte offset of the per-frame DevMesh array inside d_proj_misc
    bool frame_uses_meshes = false;
    // ... and its 2D half (rxr_set_meshes2d / rxr_set_projection2d)
    struct HostMesh2D {
        uint32_t vin_base, n_verts, n_tris, n_prims, prim_base, mode, repeat_mode, receives_light;
        rxr_source source;
        int32_t shader, chunk;
    };
    std::vector<HostMesh2D> meshes2d;
    DevBuf d_obj2d, d_proj2d_misc;
    Project2DParams PP2{};
    size_t meshes2d_prims = 0, meshes2d_tris = 0;
    bool has_matrix2d = false;
    float matrix2d[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool frame_uses_meshes2d = false;

    // Rusteria programs (rxr_set_shaders)
    DevBuf d_vm_code, d_programs, d_patterns, d_pattern_data, d_palette;
    std::vector<DevProgram> programs;
    std::vector<uint32_t> program_field_reads;  // PF_* each program reads before writing (see rxr_set_shaders)
    std::vector<uint32_t> program_flags;        // PG_* per program (see rxr_set_shaders)
    uint32_t n_patterns = 0, n_normal_patterns = 0, n_palette = 0;
    bool frame_uses_programs = false;

    FrameStream fstream;    // rxr_stream_begin .. rxr_upload_frame
    int last_upload_streamed = 0;  // 0 plain, 1 streamed (copied), 2 streamed out of page-locked arrays

    bool has_frame = false;
    // Rows [content_row0, content_row1) hold everything the resident frame can draw (rxr_upload_frame; content_known false: unknown, the
    // whole frame): outside them every pixel is the 3D miss colour, which render_impl writes with a fill instead of launching tiles there
    bool content_known = false;
    uint32_t content_row0 = 0, content_row1 = 0;
    // ... and per frame tile row the tile columns [x, y) it can draw in (RasterParams.row_spans): built by rxr_upload_frame from the batch
    // boxes in page-locked memory, copied to the device and used when they leave out enough of the content rows (spans_active)
    uint2 *h_row_spans = nullptr;    // RXR_MAX_TILE_ROWS entries, page-locked
    DevBuf d_row_spans;
    bool spans_active = false;
    bool dev_spans = false;          // device-projected meshes: the table is completed on the device behind the projection (k_spans_from_meshes)
    uint64_t last_download_bytes = 0, last_host_fill_bytes = 0;  // the last banded rxr_render_download: over PCIe / written by the host (tests)
    RasterParams P{};       // template for the resident frame (pointers resolved)
    uint32_t n_tris2d = 0;

    // last render
    bool rendered = false;
    bool last_had_prepass = false, last_had_prepass2d = false;
    uint32_t launches_since_sync = 0;   // renders queued since rxr_synchronize last drained the streams
    uint32_t rerenders = 0;             // launches rendered again by rxr_synchronize after a list overflow
    RenderSpec last_spec{};
    void *last_out = nullptr;
    hipStream_t last_stream = nullptr;
    hipStream_t upload_ordered_on = nullptr;  // stream already ordered behind the last upload
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev_upload = nullptr;
    hipEvent_t ev_render = nullptr;     // recorded on the previous stream when a render moves to ANOTHER stream (orders it behind the earlier ones)
    ProfSlot *last_prof = nullptr;  // the slot of the last render, if it was sampled (rxr_get_stats)
    std::vector<ProfSlot> prof;  // rxr_profile_begin ring
    size_t prof_next = 0;
    rxr_stats stats{};
};

// ---- helpers shared by the two translation units (defined in rxr_api.hip) -------------------------
int rxr_fail(rxr_ctx *ctx, int code, const std::string &msg);
int rxr_ensure(rxr_ctx *ctx, DevBuf &b, size_t bytes);
// waits until nothing queued by this context -- on its own streams or on the caller's stream of the last render -- is
// still running: the precondition for rewriting the staging blob, the frame blob or any scratch buffer
int rxr_quiesce(rxr_ctx *ctx);
// one render launch sequence of a plain context (rxr_api.hip)
extern "C" int rxr_render_spec(rxr_ctx *ctx, const RenderSpec &spec, void *dev_pixels, hipStream_t s);

#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return rxr_fail(ctx, RXR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

// ---- multi-device handles (rxr_multi.hip); every function expects ctx->group != nullptr -------------
void rxr_group_destroy(rxr_ctx *ctx);
int rxr_group_set_textures(rxr_ctx *ctx, const rxr_tile *static_tiles, uint32_t n_static, const rxr_tile *dynamic_tiles, uint32_t n_dynamic);
int rxr_group_set_meshes(rxr_ctx *ctx, const rxr_mesh3d *meshes, uint32_t n_meshes);
int rxr_group_set_meshes2d(rxr_ctx *ctx, const rxr_mesh2d *meshes, uint32_t n_meshes);
int rxr_group_set_projection2d(rxr_ctx *ctx, const float *mat3);
int rxr_group_set_shaders(rxr_ctx *ctx, const rxr_shader_set *set);
int rxr_group_upload_frame(rxr_ctx *ctx, const rxr_frame *frame);
int rxr_group_render(rxr_ctx *ctx);                          // every member renders its stripes (asynchronous)
int rxr_group_download(rxr_ctx *ctx, uint8_t *pixels);       // ... and ships them to host `pixels`; blocks
int rxr_group_render_download(rxr_ctx *ctx, uint8_t *pixels);
int rxr_group_synchronize(rxr_ctx *ctx);
int rxr_group_get_stats(rxr_ctx *ctx, rxr_stats *out);
int rxr_group_render_stripes_batch(rxr_ctx *ctx, uint32_t first, uint32_t stride, uint32_t n_frames, void *dev_pixels, size_t frame_stride_bytes,
                                   void *hip_stream);

// rxr_jit.hip: program sets compiled at run time
bool rxr_jit_generate(const std::vector<uint32_t> &code, const std::vector<DevProgram> &progs, std::string &src, std::string &why);
bool rxr_jit_compile(const std::string &gen, const std::string &arch, int level, std::vector<char> &obj, double &seconds, std::string &err);
int rxr_jit_build(rxr_ctx *ctx, const std::vector<uint32_t> &code, const std::vector<DevProgram> &progs);
void rxr_jit_drop(rxr_ctx *ctx);
bool rxr_jit_launch(rxr_ctx *ctx, const RasterParams *P, hipStream_t s);

// kernel durations of one sampled render from its slot: the sum over its set-up kernels and the raster kernel, microseconds
static inline bool rxr_prof_slot_us(const ProfSlot &p, float *setup_us, float *raster_us) {
    float a = 0.0f, b = 0.0f;
    for (uint32_t k = 0; k < p.n; ++k) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.ev[2u * k], p.ev[2u * k + 1u]) != hipSuccess) return false;
        (k < p.raster_first ? a : b) += ms * 1000.0f;
    }
    *setup_us = a;
    *raster_us = b;
    return true;
}
