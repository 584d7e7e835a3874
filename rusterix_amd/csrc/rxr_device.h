// rxr_device.h -- HBM data layout shared by the C-ABI implementation (rxr_api.hip) and the
// kernels (rxr_kernels.hip).  See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>

#ifdef RXR_JIT
#include "rxr.h"  // (hiprtc: the sources are in-memory headers with plain names, rxr_jit.hip)
#else
#include "../../include/rxr.h"
#endif

// GPU tile = binning granule = one 256-thread workgroup.  16x16 pixels; a wave covers 16x4.
#define RXR_TILE_W 16
#define RXR_TILE_H 16
// (Both neighbours were measured in round 2, profiles/r02/s_tile32.txt and t_tile8.txt.  16 x 32 tiles, 512-thread workgroups:
// the pre-pass gains 12 %, the raster kernels lose 10-20 % (barriers over eight waves).  16 x 8 tiles, 128-thread workgroups:
// the 1 M-triangle grid 561 -> 855 us, the bench frame 205 -> 257 us AT THE SAME VALU INSTRUCTION COUNT (1 182 against 1 178 per
// wave): with ~17 KB of LDS per workgroup only nine 2-wave workgroups fit a CU -- 4.5 waves per SIMD instead of 8 -- and the
// per-wave latency chain (records, winner's record, texel) is no longer covered.)
#define RXR_TILE_THREADS (RXR_TILE_W * RXR_TILE_H)
// a triangle whose clamped pixel box touches more bins than this goes to the "large" list that
// every tile scans (with a scalar box reject) instead of being inserted into each bin
#define RXR_LARGE_BINS 48

// one texture frame; texels live in one RGBA8 pool
struct DevTexDesc {
    uint32_t offset;  // in texels (uint32 units) from the pool base
    uint32_t w, h;
    uint32_t all_opaque;  // bit 0: every texel has alpha == 255; bit 1: texels live in the frame blob (RasterParams.frame_texels)
};

enum : uint32_t {
    DB_HAS_NORMALS = 1u << 0,
    DB_HAS_PROFILE = 1u << 1,
    DB_ALPHA_TEST = 1u << 2,  // texel alpha may be != 255: the z-write rule (rasterizer.rs:1408) needs a per-fragment sample
    DB_SKIP = 1u << 3,        // nothing of this batch can ever be written (no bbox / NaN bbox / constant alpha != 255)
    DB_RECEIVES_LIGHT = 1u << 4,
    DB_OPACITY_LIST = 1u << 5,
    DB_HAS_PROGRAM = 1u << 6,     // the batch runs a Rusteria program per fragment (DevBatch.program_plus1)
    DB_FULL_ALPHA = 1u << 7,      // the encoded alpha of a fragment needs the whole front half of the fragment block: a program
                                  // that may write `opacity`, a terrain texel (sampled by world position) or a baked shader texture
    DB_TERRAIN = 1u << 8,         // PixelSource::Terrain in a chunk: texel by world position (chunk.rs:133-151)
    // bits 16..31 of an OPACITY batch's flags: its opacity group (rxr_upload_frame; rxr_kernels.hip front_insert)
    DB_GROUP_SHIFT = 16,
};

// flattened Batch3D / Batch2D header.  The texel source is resolved on the host at upload time:
// tex >= 0 -> sample DevTexDesc[tex]; tex < 0 -> constant texel `pixel`.
struct DevBatch {
    uint32_t vert_base;  // into the vertex pools
    uint32_t tri_base;   // global index of this batch's first triangle (submission order)
    uint32_t n_tris;
    uint32_t flags;
    int32_t tex;
    uint32_t pixel;  // RGBA8 little endian (r | g<<8 | b<<16 | a<<24)
    uint32_t repeat_mode;
    uint32_t profile_id;
    float ambient[3];
    int32_t chunk;
    uint32_t mode;  // 2D: RXR_MODE_*
    uint32_t n_verts;
    uint32_t program_plus1;  // 0: no program; else 1 + index into RasterParams.programs
    uint32_t baked_plus1;    // 0: none; else 1 + DevTexDesc index of the chunk's baked shader texture (rasterizer.rs:1226-1267)
};  // 64 B

// per-triangle record for the visibility loop (written by k_setup3d).  96 B = 6 x 16 B.
struct TriSetup {
    float ea[3], eb[3], ec[3];  // Edges a/b/c (edge.rs:2-8)
    float v0x, v0y, v1x, v1y, v2x, v2y;
    float area;               // ac.x*ab.y - ac.y*ab.x (rasterizer.rs:1767)
    float iz0, iz1, iz2;      // 1.0 / v.z (rasterizer.rs:1054-1055)
    uint32_t batch;
    uint32_t bx;              // pixel box: min_x | max_x << 16 (whole screen clamp, exclusive max)
    uint32_t by;              // min_y | max_y << 16
    uint32_t bflags;          // copy of the batch's DevBatch.flags (saves a dependent load per candidate)
    uint32_t profile_id;      // copy of the batch's profile id
};
static_assert(sizeof(TriSetup) == 96, "TriSetup is staged through LDS as 6 x 16 B");
// candidates staged per round in k_raster (LDS: RXR_STAGE_TRIS * 96 B)
#ifndef RXR_STAGE_TRIS
#define RXR_STAGE_TRIS 128
#endif
// frames of at most this many triangles carry a (batch, vert_base) pair per triangle (RasterParams.tri_info)
#define RXR_TRI_INFO_MAX 16384u

// per-triangle record for shading the winning fragment.  80 B = 5 x 16 B.
struct TriShade {
    float iw0, iw1, iw2;                     // 1.0 / v.w
    float u0w, u1w, u2w, v0w, v1w, v2w;      // uv / w (rasterizer.rs:1062-1067)
    float n0[3], n1[3], n2[3];
    uint32_t pad[2];                         // the batch's texture descriptor, packed (rxr_kernels.hip TS_DESC_VALID), or zeros
};

// 2D primitive in submission order (flattened on the host at upload): a triangle of a Triangles batch
// or one Bresenham segment of a Lines / LineStrip / LineLoop batch (rasterizer.rs:602-955).  96 B, staged
// through LDS like TriSetup.  Order matters (alpha blending, :876-895), so the per-tile lists are sorted
// by primitive index before they are walked.
struct Prim2D {
    float ea[3], eb[3], ec[3];               // triangle: Edges a/b/c
    float v0x, v0y, v1x, v1y, v2x, v2y;      // triangle: vertices;  line: x0, y0, x1, y1 as int bits (`as isize`, :1785-1788), colour in v2x
    float u0, v0, u1, v1, u2, v2;            // triangle: uvs
    uint32_t batch_kind;                     // batch index << 2 | is_line << 1 | visible
    uint32_t bx, by;                         // pixel box (whole screen clamp, exclusive max), as TriSetup
};
static_assert(sizeof(Prim2D) == 96, "Prim2D is staged through LDS as 6 x 16 B");
// most primitives a tile may list before k_raster falls back to walking all of them (LDS sort capacity)
#ifndef RXR_SORT2D_MAX
#define RXR_SORT2D_MAX 1024
#endif

// what one k_scan launch works on (the 3D bins and the 2D bins use the same kernel)
struct ScanArgs {
    uint32_t n;                 // number of bins
    uint32_t list_capacity;
    const uint32_t *count;
    uint32_t *offset, *cursor, *chunk_tot, *chunk_base;
    uint32_t *counters, *counters_next, *host_status;
};

struct ChunkRange {
    uint32_t occ_first, occ_count;
    int32_t terrain_tex;      // DevTexDesc index of chunk.terrain_texture, -1 for None
    int32_t origin_x, origin_y;
    int32_t pixels_per_tile;  // texture.width as i32 / chunk.size
    uint32_t pad[2];
};

// ---- Rusteria programs on the device (rxr_vm.h) --------------------------------------------------
// The NodeOp trees are flattened by rxr_set_shaders into one word stream of jump code.  A word's low 8
// bits are the opcode: the RXR_NODE_* value for data operations, one of the VM_* control opcodes below
// otherwise; immediates follow in the next words.
enum : uint32_t {
    VM_JMP = 128,        // target
    VM_JZ = 129,         // target: pop; jump if x == 0.0 (NodeOp::If; NaN takes the then-branch)
    VM_FOR_ENTER = 130,  // push the current stack height on the loop stack (execution.rs:252)
    VM_FOR_TRUNC = 131,  // stack.truncate(base)
    VM_FOR_COND = 132,   // exit target: pop z; if z.x == 0.0 leave the loop
    VM_FOR_EXIT = 133,   // pop the loop stack
    VM_CALL = 134,       // arity, total_locals, target (NodeOp::FunctionCall)
    VM_ENDFN = 135,      // end of a function body: return to the caller, or halt for `shade`
    VM_RETURN = 136,     // address of the enclosing function's VM_ENDFN (NodeOp::Return)
    VM_FAULT = 137,      // statically known panic (e.g. a call of a function that does not exist)
    VM_GETC = 138,       // GetComponents: next word = n | idx0 << 4 | idx1 << 6 ... (2 bits each, 3 = "not x/y/z")
    VM_SETC = 139,       // SetComponents, same encoding
    VM_BINC = 140,       // fused "Push c; <binary op>": bits 8..15 = which one (VM_BINC_*), then the f32 bits of c.x, c.y, c.z
};
// the binary operations VM_BINC fuses, numbered densely (the interpreter dispatches on them through a balanced tree)
enum { VM_BINC_ADD = 0, VM_BINC_SUB, VM_BINC_MUL, VM_BINC_DIV, VM_BINC_MIN, VM_BINC_MAX, VM_BINC_MOD, VM_BINC_LT, VM_BINC_LE, VM_BINC_GT,
       VM_BINC_GE, VM_BINC_EQ, VM_BINC_NE, VM_BINC_COUNT };

#define RXR_VM_STACK 64
#define RXR_VM_LOCALS 48
#define RXR_VM_GLOBALS 16
#define RXR_VM_FRAMES 8
#define RXR_VM_LOOPS 8
#define RXR_VM_MAX_STEPS (1u << 20)
// fault codes (what the reference answers with a panic)
enum : uint32_t {
    VMF_STACK_UNDERFLOW = 1, VMF_STACK_OVERFLOW, VMF_LOCAL_INDEX, VMF_GLOBAL_INDEX, VMF_CALL_DEPTH, VMF_LOOP_DEPTH,
    VMF_STEP_LIMIT, VMF_CLAMP_BOUNDS, VMF_BAD_CALL, VMF_BAD_OPCODE, VMF_LOCALS_OVERFLOW,
    // not a fault of the program: a COMPILED set met a PaletteIndex whose slot is missing or empty (the reference then pushes nothing,
    // which only the interpreter's dynamic stack can follow) -- rxr_synchronize sends the set back to the interpreter and renders again
    VMF_JIT_PALETTE_MISS,
};
struct DevProgram {
    uint32_t shade_entry;   // word offset of the shade function in vm_code; 0xFFFFFFFF: shade_index is None
    uint32_t shade_locals;
    uint32_t n_globals;
    uint32_t flags;         // bit 0: writes opacity
};
struct DevPattern {
    uint32_t offset;        // in floats from pattern_data
    uint32_t w, h, pad;
};

// device counters.  Two sets: launch i uses set i&1 and k_scan's last block clears the other one for
// launch i+1, so no per-frame memset is needed.
enum { CNT_LARGE = 0, CNT_ENTRIES = 1, CNT_OVERFLOW = 2, CNT_TICKET = 3, CNT_WORDS = 4 };
// bins per k_scan workgroup (256 threads x 8)
#define RXR_SCAN_CHUNK 2048u
// Binned scenes whose triangle order is spatially coherent (meshes: consecutive triangles are neighbours on the screen): the bin
// lists are built without atomics on global memory, without a scan and without a second pass over all triangles -- k_blockscan
// (rxr_kernels.hip).  k_setup3d leaves, per group of 64 consecutive triangles (a wave), the union of their bin ranges; one workgroup
// per block of 4 x 4 bins keeps the groups whose range meets its block, then those groups' triangles whose range does (ids
// and packed ranges in LDS), and deals them to its 16 bins, each bin owning RasterParams.blockscan_cap list slots.  A block that
// would have to look into more than RXR_BLOCKSCAN_BLOCK_GROUPS groups or keep more than RXR_BLOCKSCAN_BLOCK_TRIS triangles, or
// a bin with more candidates than slots, raises the overflow word: the frame is then rendered again through the general
// pipeline (count / scan / fill), which the rest of the upload's launches use as well.
#define RXR_BLOCKSCAN_GROUP 64u               // triangles per group = a wave of k_setup3d
#define RXR_BLOCKSCAN_BLOCK_GROUPS 256u       // groups a block of bins looks into; more (an incoherent triangle order, a block under
                                              // a dense cloud of small meshes) raises the overflow word
#define RXR_BLOCKSCAN_GROUP_BLOCKS 64u        // scatter form: blocks of bins a group's range may meet; more raises the overflow word
#define RXR_BLOCKSCAN_SCATTER_GROUPS 256u     // more groups than this: the scatter form
#define RXR_BLOCKSCAN_MAX_WORK (48u << 20)    // groups x blocks of bins: phase 0 looks at every pair (~0.25 wave-instructions each)
#define RXR_BLOCKSCAN_BLOCK_TRIS 2048u
#define RXR_BLOCKSCAN_CAP 256u

// ---- the reference's per-tile batch box test, for batches whose box arithmetic cannot be trusted ---------------------------------
// d3_rasterize / d2_rasterize skip a batch for every tile its bounding box does not meet (rasterizer.rs:978-983, :594-600):
//     bbox.x < (tile.x + tile.width) as f32 + pad  &&  (bbox.x + bbox.width) > tile.x as f32 - pad        (and the same in y)
// A primitive's pixels lie inside its batch's box, so for ordinary coordinates the test never cuts anything a primitive covers and
// the device evaluates it once, against the whole screen.  It stops being harmless when `bbox.x + bbox.width` -- where
// width is itself max - min, rounded -- no longer lands within half a pixel of the true maximum: a batch with a vertex at -3e38 and
// another at +100 has x + width == 0, and the reference draws it in its leftmost tile column only.  The set of tiles that pass is an
// interval of tile columns times an interval of tile rows, found here with the reference's own float expressions (a binary search per
// bound: both conditions are monotone in the tile index); the device clips the pixel boxes of such a batch's primitives to it.
// RISKY = some |coordinate| or extent is not below 2^21 (ulp 0.25: the sum then stays within half a pixel) -- or not a number.
#if defined(__HIPCC__) || defined(RXR_JIT)
#define RXR_HD __host__ __device__
#else
#define RXR_HD
#endif
RXR_HD inline bool rxr_box_is_risky(float x, float y, float w, float h) {
    const float lim = 2097152.0f;
    return !(x > -lim && x < lim && y > -lim && y < lim && w > -lim && w < lim && h > -lim && h < lim);   // (NaN: risky)
}
// pixels [p0, p1) of one axis in which the reference draws a batch with box [lo, lo + extent] (as it computes them): the union of the
// tiles of size `ts` (the last one clipped to `size`, rasterizer.rs:256-270) that pass the test; p0 == p1: none
RXR_HD inline void rxr_ref_tile_span(float lo, float extent, uint32_t size, uint32_t ts, float pad, uint32_t &p0, uint32_t &p1) {
    p0 = p1 = 0u;
    if (ts == 0u || size == 0u) return;
    const uint32_t n = (size + ts - 1u) / ts;
    const float hi = lo + extent;   // the reference's `bbox.x + bbox.width`
    auto below = [&](uint32_t c) { const uint32_t t0 = c * ts, tw = (size - t0 < ts) ? size - t0 : ts; return lo < (float)(t0 + tw) + pad; };   // true from some column on
    auto above = [&](uint32_t c) { return hi > (float)(c * ts) - pad; };                                                                   // true up to some column
    if (!below(n - 1u) || !above(0u)) return;   // (also every NaN case)
    uint32_t a = 0u, b = n - 1u;                // first column with below(c)
    while (a < b) {
        const uint32_t m = (a + b) >> 1;
        if (below(m)) b = m;
        else a = m + 1u;
    }
    const uint32_t first = a;
    a = 0u, b = n - 1u;                         // last column with above(c)
    while (a < b) {
        const uint32_t m = (a + b + 1u) >> 1;
        if (above(m)) a = m;
        else b = m - 1u;
    }
    const uint32_t last = a;
    if (first > last) return;
    p0 = first * ts;
    p1 = ((last + 1u) * ts < size) ? (last + 1u) * ts : size;
}

// The same interval for a box that is NOT risky, without the searches (k_spans_from_meshes runs this once per mesh and frame in a single
// workgroup: eighteen dependent search steps were most of its 15 us): both bounds are guessed by a division and walked to the place where
// the reference's own predicate flips -- the predicates are monotone in the tile index, so whatever the guess, the walk ends on the
// bound the search finds (tests/test_host_and_abi.py compares the two on random and boundary boxes).  Risky boxes take the search.
RXR_HD inline void rxr_ref_tile_span_quick(float lo, float extent, uint32_t size, uint32_t ts, float pad, uint32_t &p0, uint32_t &p1) {
    const float lim = 2097152.0f;
    if (!(lo > -lim && lo < lim && extent > -lim && extent < lim) || ts == 0u || size == 0u) {
        rxr_ref_tile_span(lo, extent, size, ts, pad, p0, p1);
        return;
    }
    p0 = p1 = 0u;
    const uint32_t n = (size + ts - 1u) / ts;
    const float hi = lo + extent;
    auto below = [&](uint32_t c) { const uint32_t t0 = c * ts, tw = (size - t0 < ts) ? size - t0 : ts; return lo < (float)(t0 + tw) + pad; };
    auto above = [&](uint32_t c) { return hi > (float)(c * ts) - pad; };
    if (!below(n - 1u) || !above(0u)) return;
    const float fts = (float)ts;
    const float ga = (lo - pad) / fts, gb = (hi + pad) / fts;   // (|.| < 2^22: the conversions below are defined)
    uint32_t first = ga > 0.0f ? (uint32_t)ga : 0u;
    if (first > n - 1u) first = n - 1u;
    while (first > 0u && below(first - 1u)) --first;
    while (first < n - 1u && !below(first)) ++first;
    uint32_t last = gb > 0.0f ? (uint32_t)gb : 0u;
    if (last > n - 1u) last = n - 1u;
    while (last < n - 1u && above(last + 1u)) ++last;
    while (last > 0u && !above(last)) --last;
    if (first > last) return;
    p0 = first * ts;
    p1 = ((last + 1u) * ts < size) ? (last + 1u) * ts : size;
}

// kernel parameter block (passed by value; lives in the kernarg segment -> scalar loads)
// Per-light constants of the relaxed point-light term (shade3d_lights<X, true>; RXR_LIGHT_MATH=relaxed).  Everything in the term that
// depends on the light and the frame but not on the fragment -- colour x intensity x flicker, the smoothstep's reciprocal
// 1 / (start - end) and its offset -end / (start - end) -- is a function of the inputs of rxr_upload_frame (the light records and
// hash_anim), so the host evaluates it once per frame next to the light records instead of every wave once per light in VALU
// instructions on wave-uniform operands (gfx950's scalar unit has no float arithmetic): the loop reads this record through the
// scalar cache.  ss_r == 0: the light is not a well-formed emitting point light (or a parameter is outside the exact-math window)
// and takes the general path, which reads the rxr_light record itself.
struct LightFast {
    float pos[3];
    float ss_r;    // 1 / (start_distance - end_distance), or 0
    float cfi[3];  // colour * intensity * (1 - flicker value)
    float c0;      // -end_distance * ss_r:  t = clamp(distance * ss_r + c0, 0, 1)
    // (the loop reads the 32 bytes above with one scalar load; the wave-level culling step in front of it reads position, ss_r and
    // the two words below -- everything it needs of a light, in one round of loads -- for EVERY light, fast or not)
    float end_distance;
    uint32_t cull_kind;  // 1: a type with a range (point, spot, area, daylight) -- culled against the wave's bounding sphere
    uint32_t pad[2];
};
static_assert(sizeof(LightFast) == 48, "LightFast: 32 bytes for the loop + 16 for the culling step");

struct RasterParams {
    uint32_t width, height;
    uint32_t row0, row1;           // band of rows this launch renders
    uint32_t tiles_x, tiles_y;     // tile grid of the band
    uint32_t tile_y0;              // first tile row of the launch (in whole-frame tile coordinates)
    uint32_t tile_stride;          // launch tile row l is frame tile row tile_y0 + l*tile_stride (1 = contiguous band)
    uint32_t compact;              // 1: output row = l*RXR_TILE_H + ly (stripe buffer); 0: frame addressing
    uint32_t flags;                // RXR_FLAG_*
    float fwidth, fheight;
    float inv_view[16], inv_proj[16];
    float cam[3];
    float translationd2[2];
    float scaled2;
    uint32_t hash_anim;
    uint32_t sample_mode;
    uint32_t background_color;     // RGBA8 packed
    uint32_t background_kind;
    float ambient[4];
    float sun_dir[3];
    float day_factor;

    uint32_t n_tris3d, n_batches3d, n_lights, n_occluders, n_linedefs, n_prims2d, any_lights, has_opacity;
    uint32_t binned2d;             // 1: the 2D primitives were binned (n_prims2d > RXR_STAGE_TRIS); 0: implicit ordered list
    uint32_t d2_box[4];            // union of the 2D primitives' pixel boxes: min_x, max_x, min_y, max_y (max exclusive)
    const uint32_t *d2_box_dev;    // device-projected 2D batches: the same four words, written by k_proj2d_prims (else nullptr)
    uint32_t list2d_capacity;
    uint32_t list_capacity;
    uint32_t ref_tile;             // the reference's tile_size (rxr_frame.tile_size): only the batch box test of RISKY batches depends on it
    const uint4 *batch_clip3d;     // host-projected frames with a risky 3D batch: per batch the pixels (x0, x1, y0, y1) the reference draws it in
                                   // (rxr_ref_tile_span; ordinary batches: the whole frame); NULL otherwise.  make_setup clips the pixel boxes
    uint32_t fused_small;          // small-scene mode (whole frame <= RXR_STAGE_TRIS triangles): 0 = binned pipeline,
                                   // 1 = fully fused (k_raster_fused builds the records itself, no pre-pass launch),
                                   // 2 = implicit list (k_setup3d writes the records, no scan / fill / bins; default)

    // geometry inputs (indexed, as handed over by the host)
    const float4 *pv;              // projected_vertices
    const float2 *uv;              // clipped_uvs
    const float *nrm;              // clipped_normals, 3 floats per vertex
    const uint32_t *idx;           // clipped_indices, 3 per triangle
    const rxr_edges *edges;
    const DevBatch *batches3d;
    const uint32_t *batch_tri_base;  // n_batches3d + 1 prefix array for the triangle -> batch search
    const uint2 *tri_info;           // small frames (<= RXR_TRI_INFO_MAX triangles, host-projected): (batch, its vert_base) per triangle, written
                                     // at upload -- k_setup3d of such a frame is one workgroup of pure latency, and the search above is five
                                     // dependent loads of it; nullptr otherwise
    const struct DevBBox *dev_bbox;  // device-projection path: per-batch boxes accumulated on the device (else NULL)
    const struct DevMesh *pm_meshes; // device-projection path, edges fused into the set-up (RXR_PROJ_FUSED_EDGES): the frame's mesh headers and
    const uint8_t *pm_edge_vis;      // edge_visibility per original triangle -- make_setup builds the Edges record itself (else NULL: P.edges)
    const uint32_t *mesh_live;       // device-projection path: per batch, the triangle slots in use (rxr_project.h); slots behind them
                                     // hold no triangle and nothing may be read from their records (else NULL)

    // set-up outputs
    TriSetup *tri_setup;
    TriShade *tri_shade;
    uint2 *group_rng;                  // k_blockscan: per wave of k_setup3d (64 consecutive triangles) the union of its triangles' bin ranges,
                                       // packed like a triangle's (bx0 | bx1 << 16, l0 | l1 << 16); (0xFFFF, 0xFFFF) -- first bin 65535, last bin 0 -- when no triangle of the group has one
    uint2 *tri_box;                    // (bx, by) of every TriSetup once more, densely: k_fill reads these 8 bytes instead of a 96-byte stride
    uint32_t *bin_count;           // tiles_x * tiles_y; all-zero between launches (k_raster clears its own bin)
    uint32_t *bin_offset;          // chunk-local exclusive scan of bin_count; add chunk_base[bin / RXR_SCAN_CHUNK]
    uint32_t *bin_cursor;
    uint32_t *chunk_tot;           // per k_scan workgroup: number of list entries of its chunk
    uint32_t *chunk_base;          // exclusive scan of chunk_tot (written by the last k_scan workgroup)
    uint32_t *counters_next;       // the other counter set, cleared for the next launch
    uint32_t plain_programs;       // host only: the frame runs programs but needs none of level 1's chunk paths (k_raster_vm_p instead of k_raster_vm_sv)
    uint32_t relaxed_lights;       // host only: RXR_LIGHT_MATH=relaxed -- feature levels 0 and 1 launch the kernels whose 3D light loop uses the
                                   // relaxed arithmetic (k_raster_rl, k_raster_rows_rl, k_raster_chunk_rl; shade3d_lights<X, true>)
    uint32_t blockscan_scatter;    // k_blockscan's phase 0 the other way round (many groups): every group has appended itself to the lists of the
                                   // blocks its range meets (k_setup3d, at most RXR_BLOCKSCAN_GROUP_BLOCKS of them) instead of every block
                                   // reading every group's range
    uint32_t blk_wide_base;        // first slot of the list of "wide" groups (ranges over more than RXR_BLOCKSCAN_GROUP_BLOCKS blocks) in blk_grp
    uint32_t *blk_cnt, *blk_grp;   // per block of bins: number of groups (all-zero between launches: k_blockscan hands it back) and their ids,
                                   // RXR_BLOCKSCAN_BLOCK_GROUPS slots each
    uint32_t blockscan2d_cap;      // != 0: the 2D bin lists of this launch come from k_blockscan2d -- slots per bin; the lists are SORTED by primitive
                                   // index (submission order) and there is no list of large primitives: the raster kernel skips its per-tile sort
    uint32_t blockscan_cap;        // != 0: this launch bins with k_blockscan (k_setup3d counts nothing); list slots per bin
    uint32_t any_occluders;        // the frame has an occluder somewhere (mapmini's or a chunk's): get_occlusion compares world positions with their boxes
    float rl_flip_guard;           // relaxed light mode: the smallest |n.v| for which the normal's flip toward the camera is decided from the
                                   // relaxed values (shade3d_begin; 1e-4, a hundred times their error; RXR_RL_FLIP_GUARD for tests)
    uint32_t *host_status;         // pinned host words (device-visible), indexed like the counters: [CNT_ENTRIES], [CNT_OVERFLOW]
    uint32_t *bin_list;
    uint32_t *large_list;
    uint32_t *counters;

    const rxr_light *lights;
    const LightFast *lights_fast;      // one per light, made by the host with the frame (rxr_upload_frame): shade3d_lights<X, true>
    const rxr_occluder *occluders;     // mapmini occluders first, then the chunks'
    const rxr_linedef *linedefs;
    const ChunkRange *chunks;

    const DevBatch *batches2d;
    const Prim2D *prim2d;
    uint32_t *bin2d_count;         // own zero-invariant buffer, like bin_count
    uint32_t *bin2d_offset, *bin2d_cursor, *chunk2d_tot, *chunk2d_base;
    uint32_t *bin2d_list, *large2d_list;
    uint32_t *counters2d, *counters2d_next;
    uint32_t *host_status2d;

    // Rusteria programs (rxr_set_shaders)
    uint32_t kernel_level;             // 0: k_raster; 1: k_raster_chunk (a visible batch uses a terrain / baked texture);
                                       // 2: k_raster_vm (a visible batch runs a program); 3: k_raster_vm_s (all programs have static
                                       // stack depths); 4: k_raster_vm_sv (3, and no opaque-pass program writes `opacity`); 5: k_raster_vm_v (2, likewise)
    const uint32_t *vm_code;
    const DevProgram *programs;
    const DevPattern *patterns;        // n_patterns colour patterns, then n_normal_patterns normal patterns
    const float *pattern_data;
    const float *palette;              // n_palette x 4: r, g, b, present (0 / 1)
    uint32_t n_programs, n_patterns, n_normal_patterns, n_palette;
    uint32_t *vm_fault;                // pinned host word: a non-zero VMF_* code if any fragment's program faulted
    uint32_t *staircase_overflow;      // pinned host word: set when a pixel's opacity staircase had to drop an entry (front_insert)
    float time;                        // Rasterizer.time
    float bg_grid[4];                  // RXR_BG_GRID: grid_size, subdivisions, offset.x, offset.y
    uint32_t has_brush;                // Rasterizer.brush_preview (feature level >= 1)
    float brush_pos[3], brush_radius, brush_falloff;

    const DevTexDesc *tex;             // resident textures first, then this frame's chunk textures
    const uint32_t *frame_texels;      // texel base of the latter (inside the frame blob)
    const uint32_t *texels;
    const uint32_t *bg_pixels;     // RXR_BG_HOST_PIXELS
    uint32_t *out;                 // framebuffer; row `row0` of the band is at out + out_row0_offset
    uint64_t out_row_stride;       // in pixels
    int64_t out_base_row;          // row index that `out` points at (0 for the context framebuffer, row0 for external)
    uint32_t bin_row0;             // raster launches only: launch tile row l reads the bins of pre-pass row bin_row0 + l (0 unless ONE pre-pass
                                   // over a band is followed by several raster launches over parts of it: rxr_render_download's pipeline)
    const uint32_t *edge_vis3d;    // host-projected frames handed over WITHOUT Edges records (rxr_batch3d.edges == NULL, ABI 5): per triangle != 0 iff
                                   // its record's `visible`; make_setup builds the record from the projected vertices under the batch's cull
                                   // mode (DevBatch.mode), as it does for device-projected frames.  NULL: P.edges holds the records
    const uint2 *row_spans;        // per FRAME tile row the tile columns [x, y) that anything of the frame can reach (rxr_upload_frame, from the
                                   // batch boxes), or NULL: the raster grid is as wide as the widest span of the launch, workgroup (bx, ty)
                                   // takes column x + bx and leaves at once behind y; the pixels outside the spans get the miss colour from
                                   // k_fill_outside_spans.  Contiguous, non-compact launches only.
    uint32_t split_rounds;         // some kept opaque-pass batch is a cut-out (DB_ALPHA_TEST / DB_FULL_ALPHA) or, under an opacity pass, carries a
                                   // profile id: k_raster_rows_cut[_rl] (rxr_upload_frame)
    uint32_t pad_tail;
};
