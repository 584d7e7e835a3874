// rxr_project.hip -- device-side Batch3D::clip_and_project + Edges::new + bounding box
// (reference src/batch/batch3d.rs:482-768, src/edge.rs:12-24).
//
// Same exactness rules as rxr_kernels.hip: -ffp-contract=off, IEEE division, fmaf only where vek's
// Mat4 * Vec4 fuses.  The host half that remains per frame is two Mat4 * Mat4 products and the 8-corner
// AABB frustum test per mesh (rxr_api.hip); everything per vertex / per triangle happens here.
//
// Ordering: the reference appends the vertices / fan triangles created by near-plane clipping after
// the originals, in the order of the original triangles (:627-686).  Triangle order is the tie-break
// of the depth test, so it is reproduced exactly: k_clip_count counts what each triangle emits, an
// exclusive scan turns the counts into append offsets, k_clip_emit writes at those offsets.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "rxr_launch.h"
#include "rxr_project.h"

#ifndef RXR_VEK_FUSED_MATVEC
#define RXR_VEK_FUSED_MATVEC 1
#endif

namespace {

__device__ __forceinline__ float madd(float a, float b, float c) {
#if RXR_VEK_FUSED_MATVEC
    return fmaf(a, b, c);
#else
    return a * b + c;
#endif
}

// vek column-major Mat4 * Vec4 (m[c*4+r])
__device__ __forceinline__ float4 mat4_mul(const float *m, float4 v) {
    float4 o;
    o.x = madd(m[12], v.w, madd(m[8], v.z, madd(m[4], v.y, m[0] * v.x)));
    o.y = madd(m[13], v.w, madd(m[9], v.z, madd(m[5], v.y, m[1] * v.x)));
    o.z = madd(m[14], v.w, madd(m[10], v.z, madd(m[6], v.y, m[2] * v.x)));
    o.w = madd(m[15], v.w, madd(m[11], v.z, madd(m[7], v.y, m[3] * v.x)));
    return o;
}

// order-preserving f32 -> u32 (so that integer atomicMin/Max implement f32 min/max)
__device__ __forceinline__ uint32_t enc(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// largest b in [0, n) with prefix[b] <= i  (prefix has n + 1 entries, meshes may be empty)
__device__ __forceinline__ uint32_t find_mesh(const uint32_t *prefix, uint32_t n, uint32_t i) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The same for the 64 consecutive items of a wave (i = first + lane, lanes in item order; call with all lanes that have an
// item): they nearly always belong to one mesh, so the search runs once, for the wave's first item, with scalar loads through
// the constant address space (the prefix is read-only for the launch) -- not nine dependent vector-memory round trips at the
// head of every thread.  Only a wave that straddles a mesh boundary searches per lane.
__device__ __forceinline__ uint32_t find_mesh_wave(const uint32_t *prefix, uint32_t n, uint32_t i) {
    typedef const uint32_t __attribute__((address_space(4))) *cptr;
    const cptr cp = (cptr)prefix;
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (cp[mid] <= first) lo = mid;
        else hi = mid;
    }
    if (lo + 1u < n && cp[lo + 1u] <= first + 63u) return find_mesh(prefix, n, i);   // wave-uniform branch
    return lo;
}

// projection to the screen, batch3d.rs:689-700
__device__ __forceinline__ float4 to_screen(const ProjectParams &P, float4 vs) {
    float4 r = mat4_mul(P.projection, vs);
    float w = r.w;
    float4 o;
    o.x = ((r.x / w) * 0.5f + 0.5f) * P.width;
    o.y = ((-r.y / w) * 0.5f + 0.5f) * P.height;
    o.z = r.z / w;
    o.w = w;
    return o;
}

// f32::min / f32::max drop NaN (batch3d.rs:755-760): a NaN coordinate does not contribute
__device__ __forceinline__ void bbox_add(DevBBox *bb, float x, float y) {
    if (x == x) {
        atomicMin(&bb->min_x, enc(x));
        atomicMax(&bb->max_x, enc(x));
    }
    if (y == y) {
        atomicMin(&bb->min_y, enc(y));
        atomicMax(&bb->max_y, enc(y));
    }
}

// Wave-cooperative form for kernels where a wave's lanes almost always belong to one mesh: reduce the
// encoded min / max across the wave (integer min/max of the order-preserving encoding == f32 min/max,
// NaN lanes contribute the neutral element) and let one lane issue the four atomics.  Must be called
// by every lane of the wave; `active` marks lanes that carry a vertex of mesh `b`.
__device__ __forceinline__ void bbox_add_wave(DevBBox *boxes, uint32_t b, bool active, float x, float y) {
    const uint32_t b0 = __shfl(b, __ffsll((long long)__ballot(active)) - 1, 64);
    const bool uniform = __ballot(active && b != b0) == 0ull;
    if (!uniform) {  // a mesh boundary inside the wave: plain per-lane atomics
        if (active) bbox_add(&boxes[b], x, y);
        return;
    }
    uint32_t mnx = (active && x == x) ? enc(x) : 0xFFFFFFFFu, mxx = (active && x == x) ? enc(x) : 0u;
    uint32_t mny = (active && y == y) ? enc(y) : 0xFFFFFFFFu, mxy = (active && y == y) ? enc(y) : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mnx = min(mnx, (uint32_t)__shfl_xor((int)mnx, d, 64));
        mxx = max(mxx, (uint32_t)__shfl_xor((int)mxx, d, 64));
        mny = min(mny, (uint32_t)__shfl_xor((int)mny, d, 64));
        mxy = max(mxy, (uint32_t)__shfl_xor((int)mxy, d, 64));
    }
    if ((threadIdx.x & 63u) == 0u) {
        DevBBox *bb = &boxes[b0];
        // neutral values mean "no finite-or-infinite coordinate seen": skip them (the box keeps +-inf)
        if (mnx != 0xFFFFFFFFu) atomicMin(&bb->min_x, mnx);
        if (mxx != 0u) atomicMax(&bb->max_x, mxx);
        if (mny != 0xFFFFFFFFu) atomicMin(&bb->min_y, mny);
        if (mxy != 0u) atomicMax(&bb->max_y, mxy);
    }
}

// The same one level up, for kernels whose WORKGROUPS almost always lie inside one mesh (k_proj_vertices: a mesh of the 1 M-triangle
// grid is 27 workgroups): the four waves' partial boxes meet in LDS and ONE lane issues the atomics for every run of waves with the same
// mesh -- a quarter of the same-line atomics that are this kernel's run time.  Must be called by every thread of the workgroup
// (a barrier inside); `active` marks lanes that carry a vertex of mesh `b`.
__device__ __forceinline__ void bbox_add_block(DevBBox *boxes, uint32_t b, bool active, float x, float y) {
    __shared__ uint32_t s_part[4][5];   // per wave: mesh (0xFFFFFFFF: nothing to merge), min_x, max_x, min_y, max_y (encoded)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long act = __ballot(active);
    uint32_t mesh = 0xFFFFFFFFu, mnx = 0xFFFFFFFFu, mxx = 0u, mny = 0xFFFFFFFFu, mxy = 0u;
    if (act) {  // wave-uniform
        const uint32_t b0 = __shfl(b, __ffsll((long long)act) - 1, 64);
        if (__ballot(active && b != b0) != 0ull) {  // a mesh boundary inside the wave: plain per-lane atomics
            if (active) bbox_add(&boxes[b], x, y);
        } else {
            mesh = b0;
            mnx = (active && x == x) ? enc(x) : 0xFFFFFFFFu; mxx = (active && x == x) ? enc(x) : 0u;
            mny = (active && y == y) ? enc(y) : 0xFFFFFFFFu; mxy = (active && y == y) ? enc(y) : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                mnx = min(mnx, (uint32_t)__shfl_xor((int)mnx, d, 64));
                mxx = max(mxx, (uint32_t)__shfl_xor((int)mxx, d, 64));
                mny = min(mny, (uint32_t)__shfl_xor((int)mny, d, 64));
                mxy = max(mxy, (uint32_t)__shfl_xor((int)mxy, d, 64));
            }
        }
    }
    if (lane == 0u) {
        s_part[wave][0] = mesh; s_part[wave][1] = mnx; s_part[wave][2] = mxx; s_part[wave][3] = mny; s_part[wave][4] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t cur = 0xFFFFFFFFu, a = 0xFFFFFFFFu, c = 0u, e = 0xFFFFFFFFu, g = 0u;
        auto flush = [&]() {
            if (cur == 0xFFFFFFFFu) return;
            DevBBox *bb = &boxes[cur];
            // neutral values mean "no finite-or-infinite coordinate seen": skip them (the box keeps +-inf)
            if (a != 0xFFFFFFFFu) atomicMin(&bb->min_x, a);
            if (c != 0u) atomicMax(&bb->max_x, c);
            if (e != 0xFFFFFFFFu) atomicMin(&bb->min_y, e);
            if (g != 0u) atomicMax(&bb->max_y, g);
        };
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t m = s_part[w][0];
            if (m == 0xFFFFFFFFu) continue;
            if (m != cur) {
                flush();
                cur = m; a = 0xFFFFFFFFu; c = 0u; e = 0xFFFFFFFFu; g = 0u;
            }
            a = min(a, s_part[w][1]); c = max(c, s_part[w][2]); e = min(e, s_part[w][3]); g = max(g, s_part[w][4]);
        }
        flush();
    }
    __syncthreads();  // (s_part is rewritten by the caller's next round)
}

struct Clip {
    int nv;            // emitted vertices (0, 3 or 4)
    bool edge_vis;     // edge_visibility[triangle] (:582-618)
};

// classification of one original triangle, batch3d.rs:586-623.  z0..z2 are view-space z.
__device__ __forceinline__ Clip classify(const DevMesh &M, float4 v0, float4 v1, float4 v2) {
    Clip c;
    c.nv = 0;
    c.edge_vis = true;
    if (M.cull_mode != RXR_CULL_OFF) {  // :592-600 -- skips the clip step, leaves edge_visibility true
        float orient = (v1.x - v0.x) * (v2.y - v0.y) - (v1.y - v0.y) * (v2.x - v0.x);
        bool is_front = orient > 0.0f;
        if (M.cull_mode == RXR_CULL_BACK && is_front) return c;
        if (M.cull_mode == RXR_CULL_FRONT && !is_front) return c;
    }
    const float near_plane = 0.1f;
    bool in0 = v0.z < -near_plane, in1 = v1.z < -near_plane, in2 = v2.z < -near_plane;
    if (in0 && in1 && in2) return c;
    c.edge_vis = false;
    if (!in0 && !in1 && !in2) return c;
    // :630-669: one vertex per inside corner plus one per edge that crosses the plane
    c.nv = (int)in0 + (int)in1 + (int)in2 + (int)(in0 != in1) + (int)(in1 != in2) + (int)(in2 != in0);
    return c;
}

}  // namespace

// one-time (per rxr_set_meshes): originals of clipped_indices / clipped_uvs / clipped_normals
// (batch3d.rs:566-574) at their fixed slots, appended slots zeroed
extern "C" __global__ void __launch_bounds__(256) k_proj_static(ProjectParams P) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n_verts_in) {
        uint32_t b = find_mesh(P.vin_prefix, P.n_meshes, i);
        const DevMesh &M = P.meshes[b];
        uint32_t o = M.vout_base + (i - M.vin_base);
        P.uv[o] = P.obj_uvs[i];
        P.nrm[3 * (size_t)o + 0] = P.obj_normals[3 * (size_t)i + 0];
        P.nrm[3 * (size_t)o + 1] = P.obj_normals[3 * (size_t)i + 1];
        P.nrm[3 * (size_t)o + 2] = P.obj_normals[3 * (size_t)i + 2];
    }
    if (i < P.n_tris_in) {
        uint32_t b = find_mesh(P.tin_prefix, P.n_meshes, i);
        const DevMesh &M = P.meshes[b];
        uint32_t o = M.tout_base + (i - M.tin_base);
        P.idx[3 * (size_t)o + 0] = P.obj_idx[3 * (size_t)i + 0];
        P.idx[3 * (size_t)o + 1] = P.obj_idx[3 * (size_t)i + 1];
        P.idx[3 * (size_t)o + 2] = P.obj_idx[3 * (size_t)i + 2];
    }
}

// per frame: reset the per-mesh boxes to (+inf, +inf, -inf, -inf), batch3d.rs:750-753
namespace {
// P.ticket[1]: does ANY triangle of the frame append vertices / fan triangles (near-plane clip)?  Cleared here, raised by k_clip_count.
// A frame in which none does -- the usual case -- has an all-zero append table: its scan is the table itself and there is nothing to
// emit, so k_proj_scan and k_clip_emit leave at once (the 1 M-triangle frame: 20.5 + 10.7 us of the device-projected pre-pass).
__device__ __forceinline__ void proj_init_item(const ProjectParams &P, uint32_t b) {
    if (b == 0u) P.ticket[1] = 0u;
    if (b >= P.n_meshes) return;
    DevBBox bb;
    bb.min_x = bb.min_y = enc(INFINITY);
    bb.max_x = bb.max_y = enc(-INFINITY);
    P.bbox[b] = bb;
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_proj_init(ProjectParams P) { proj_init_item(P, blockIdx.x * blockDim.x + threadIdx.x); }

// per frame: view transform (:555-560) and screen projection (:689-700) of the ORIGINAL vertices
namespace {
// (called by whole workgroups whose threads hold consecutive items: find_mesh_wave, bbox_add_block)
__device__ __forceinline__ void proj_vertices_item(const ProjectParams &P, uint32_t i) {
    bool active = i < P.n_verts_in;
    uint32_t b = 0;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (active) {
        b = find_mesh_wave(P.vin_prefix, P.n_meshes, i);
        const DevMesh &M = P.meshes[b];
        active = !M.rejected;
        if (active) {
            uint32_t o = M.vout_base + (i - M.vin_base);
            float4 vs = mat4_mul(M.view_model, P.obj_verts[i]);
            P.view_verts[o] = vs;
            s = to_screen(P, vs);
            P.pv[o] = s;
        }
    }
    bbox_add_block(P.bbox, b, active, s.x, s.y);  // (every thread of the workgroup)
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_proj_vertices(ProjectParams P) { proj_vertices_item(P, blockIdx.x * blockDim.x + threadIdx.x); }

// per frame: what each original triangle appends (:586-681)
namespace {
__device__ __forceinline__ void clip_count_item(const ProjectParams &P, uint32_t t) {
    if (t > P.n_tris_in) return;
    if (t == P.n_tris_in) {  // sentinel: after the scan it holds the grand total
        P.append[t] = 0ull;
        return;
    }
    uint32_t b = find_mesh_wave(P.tin_prefix, P.n_meshes, t);
    const DevMesh &M = P.meshes[b];
    if (M.rejected) {
        P.append[t] = 0ull;
        P.edge_vis[t] = 0;
        return;
    }
    const uint32_t *ix = P.obj_idx + 3 * (size_t)t;
    float4 v0 = P.view_verts[M.vout_base + ix[0]], v1 = P.view_verts[M.vout_base + ix[1]], v2 = P.view_verts[M.vout_base + ix[2]];
    Clip c = classify(M, v0, v1, v2);
    P.edge_vis[t] = c.edge_vis ? 1 : 0;
    uint32_t nt = c.nv >= 3 ? (uint32_t)(c.nv - 2) : 0u;
    P.append[t] = (AppendCount)(uint32_t)c.nv | ((AppendCount)nt << 32);
    if (c.nv != 0) __hip_atomic_store(&P.ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (rare; every writer stores the same value)
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_clip_count(ProjectParams P) { clip_count_item(P, blockIdx.x * blockDim.x + threadIdx.x); }

// exclusive scan of P.append[0 .. n_tris_in] (n_tris_in + 1 entries) in place: chunk-local prefixes,
// chunk bases written by the workgroup that finishes last (same scheme as k_scan in rxr_kernels.hip)
extern "C" __global__ void __launch_bounds__(256) k_proj_scan(ProjectParams P) {
    __shared__ AppendCount wave_tot[4];
    __shared__ uint32_t s_last;
    const uint32_t n = P.n_tris_in + 1u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (__hip_atomic_load(&P.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {  // (uniform) nothing is appended: the table of zeros is its own scan
        if (tid == 0) P.chunk_base[blockIdx.x] = 0ull;
        return;
    }
    constexpr uint32_t PER = RXR_PROJ_SCAN_CHUNK / 256u;
    const uint32_t i0 = blockIdx.x * RXR_PROJ_SCAN_CHUNK + tid * PER;
    AppendCount v[PER];
    AppendCount sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        uint32_t i = i0 + k;
        v[k] = (i < n) ? P.append[i] : 0ull;
        sum += v[k];
    }
    AppendCount inc = sum;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        AppendCount o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    AppendCount wave_off = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        AppendCount wt = wave_tot[w];
        if (w < wave) wave_off += wt;
        total += wt;
    }
    AppendCount run = wave_off + (inc - sum);
#pragma unroll
    for (uint32_t k = 0; k < PER; ++k) {
        uint32_t i = i0 + k;
        if (i < n) P.append[i] = run;
        run += v[k];
    }
    if (tid == 0) {
        __hip_atomic_store(&P.chunk_tot[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        uint32_t ticket = atomicAdd(P.ticket, 1u);
        s_last = (ticket == gridDim.x - 1u) ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || wave != 0) return;
    __threadfence();
    AppendCount carry = 0;
    for (uint32_t base = 0; base < gridDim.x; base += 64u) {
        uint32_t c = base + lane;
        AppendCount t = (c < gridDim.x) ? __hip_atomic_load(&P.chunk_tot[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        AppendCount ic = t;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            AppendCount o = __shfl_up(ic, d, 64);
            if (lane >= d) ic += o;
        }
        if (c < gridDim.x) P.chunk_base[c] = carry + ic - t;
        carry += __shfl(ic, 63, 64);
    }
    if (lane == 0) *P.ticket = 0u;  // ready for the next frame
}

namespace {
__device__ __forceinline__ AppendCount prefix_at(const ProjectParams &P, uint32_t i) {
    return P.chunk_base[i / RXR_PROJ_SCAN_CHUNK] + P.append[i];
}
}  // namespace

// per frame: Sutherland-Hodgman against z = -0.1 for the mixed triangles, appended vertices and fan
// triangles written at their scanned offsets (:626-686), new vertices projected (:689-700)
namespace {
// per mesh: how many of its 3 * n_tris triangle slots are in use (the pools are capacity based: an unclipped scene leaves two thirds
// of them unused) -- written by the thread of the mesh's FIRST triangle in k_clip_emit (a mesh without triangles has no slots)
__device__ __forceinline__ void proj_live_item(const ProjectParams &P, uint32_t b) {
    if (b >= P.n_meshes) return;
    const DevMesh &M = P.meshes[b];
    uint32_t live = 0;
    if (!M.rejected) {
        const AppendCount tot = prefix_at(P, M.tin_base + M.n_tris) - prefix_at(P, M.tin_base);
        live = M.n_tris + (uint32_t)(tot >> 32);
    }
    P.mesh_live[b] = live;
}
__device__ __forceinline__ void clip_emit_item(const ProjectParams &P, uint32_t t) {
    if (t >= P.n_tris_in) return;
    uint32_t b = find_mesh_wave(P.tin_prefix, P.n_meshes, t);
    const DevMesh &M = P.meshes[b];
    if (t == M.tin_base) proj_live_item(P, b);
    if (M.rejected) return;
    if (__hip_atomic_load(&P.ticket[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // (uniform) no triangle of the frame emits anything
    const uint32_t *ix = P.obj_idx + 3 * (size_t)t;
    const uint32_t gi[3] = {M.vout_base + ix[0], M.vout_base + ix[1], M.vout_base + ix[2]};
    float4 v[3] = {P.view_verts[gi[0]], P.view_verts[gi[1]], P.view_verts[gi[2]]};
    Clip c = classify(M, v[0], v[1], v[2]);
    if (c.nv < 3) return;

    AppendCount rel = prefix_at(P, t) - prefix_at(P, M.tin_base);  // no borrow: both halves are monotone
    uint32_t voff = (uint32_t)(rel & 0xFFFFFFFFull), toff = (uint32_t)(rel >> 32);
    const uint32_t first_local = M.n_verts + voff;               // mesh-local index of the first emitted vertex
    uint32_t out = M.vout_base + first_local;

    const float near_plane = 0.1f;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3;
        const float4 cur = v[i], nxt = v[j];
        const float2 uvc = P.uv[gi[i]], uvn = P.uv[gi[j]];
        const float *ncp = P.nrm + 3 * (size_t)gi[i], *nnp = P.nrm + 3 * (size_t)gi[j];
        const float nc[3] = {ncp[0], ncp[1], ncp[2]}, nn[3] = {nnp[0], nnp[1], nnp[2]};
        if (cur.z < -near_plane) {  // :640-646
            float4 s = to_screen(P, cur);
            P.pv[out + k] = s;
            P.uv[out + k] = uvc;
            P.nrm[3 * (size_t)(out + k) + 0] = nc[0];
            P.nrm[3 * (size_t)(out + k) + 1] = nc[1];
            P.nrm[3 * (size_t)(out + k) + 2] = nc[2];
            bbox_add(&P.bbox[b], s.x, s.y);
            ++k;
        }
        if ((cur.z < -near_plane) != (nxt.z < -near_plane)) {  // :648-668
            float tt = (-near_plane - cur.z) / (nxt.z - cur.z);
            float4 isect;
            isect.x = cur.x + tt * (nxt.x - cur.x);
            isect.y = cur.y + tt * (nxt.y - cur.y);
            isect.z = cur.z + tt * (nxt.z - cur.z);
            isect.w = cur.w + tt * (nxt.w - cur.w);
            float2 iuv;
            iuv.x = uvc.x + tt * (uvn.x - uvc.x);
            iuv.y = uvc.y + tt * (uvn.y - uvc.y);
            // (n_current * (1.0 - t) + n_next * t).normalized()
            float omt = 1.0f - tt;
            float nx = nc[0] * omt + nn[0] * tt, ny = nc[1] * omt + nn[1] * tt, nz = nc[2] * omt + nn[2] * tt;
            float mag = sqrtf((nx * nx + ny * ny) + nz * nz);
            float4 s = to_screen(P, isect);
            P.pv[out + k] = s;
            P.uv[out + k] = iuv;
            P.nrm[3 * (size_t)(out + k) + 0] = nx / mag;
            P.nrm[3 * (size_t)(out + k) + 1] = ny / mag;
            P.nrm[3 * (size_t)(out + k) + 2] = nz / mag;
            bbox_add(&P.bbox[b], s.x, s.y);
            ++k;
        }
    }
    // fan, :672-678 (mesh-local indices, as the reference stores them)
    uint32_t ts = M.tout_base + M.n_tris + toff;
    for (int i = 1; i + 1 < k; ++i) {
        P.idx[3 * (size_t)ts + 0] = first_local;
        P.idx[3 * (size_t)ts + 1] = first_local + (uint32_t)i;
        P.idx[3 * (size_t)ts + 2] = first_local + (uint32_t)i + 1u;
        ++ts;
    }
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_clip_emit(ProjectParams P) { clip_emit_item(P, blockIdx.x * blockDim.x + threadIdx.x); }

// per frame: Edges for every triangle slot (:706-739 + edge.rs:12-24); unused slots become invisible
// one slot's record (all zero = unused / invisible)
__device__ __forceinline__ rxr_edges edges_of_slot(const ProjectParams &P, uint32_t s) {
    uint32_t b = find_mesh_wave(P.tout_prefix, P.n_meshes, s);
    const DevMesh &M = P.meshes[b];
    uint32_t local = s - M.tout_base;
    rxr_edges E;
#pragma unroll
    for (int i = 0; i < 3; ++i) E.a[i] = E.b[i] = E.c[i] = 0.0f;
    E.visible = 0;
    bool used = false, evis = true;
    if (!M.rejected) {
        if (local < M.n_tris) {
            used = true;
            evis = P.edge_vis[M.tin_base + local] != 0;
        } else {
            AppendCount tot = prefix_at(P, M.tin_base + M.n_tris) - prefix_at(P, M.tin_base);
            used = (local - M.n_tris) < (uint32_t)(tot >> 32);  // appended fans: edge_visibility defaults to true (:731-732)
        }
    }
    if (!used) return E;
    const uint32_t *ix = P.idx + 3 * (size_t)s;
    const float4 v0 = P.pv[M.vout_base + ix[0]], v1 = P.pv[M.vout_base + ix[1]], v2 = P.pv[M.vout_base + ix[2]];
    return edges_from_vertices(M.cull_mode, evis, v0, v1, v2);
}

// Small frames (RXR_PROJ_SMALL_MAX original vertices and triangles): every step above in ONE workgroup -- the six dependent launches
// of rxr_launch_project cost a map or a teapot ~5 us each, several times the work in them.  The same item functions in the same
// order; a workgroup barrier (which orders the workgroup's global-memory traffic) stands where a kernel boundary stood; the scan is
// the workgroup's own (every item lies in chunk 0 of the prefix scheme: chunk_base[0] = 0).
#ifndef RXR_PROJ_SMALL_MAX
#define RXR_PROJ_SMALL_MAX 1024u
#endif
static_assert(RXR_PROJ_SMALL_MAX < RXR_PROJ_SCAN_CHUNK, "k_proj_small scans chunk 0 only");
extern "C" __global__ void __launch_bounds__(256) k_proj_small(ProjectParams P) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ AppendCount wave_tot[4];
    for (uint32_t b = tid; b < P.n_meshes; b += 256u) proj_init_item(P, b);
    __syncthreads();
    // (whole waves: the loop bounds are rounded up to the workgroup, the items check their range themselves)
    for (uint32_t i = tid; i < ((P.n_verts_in + 255u) & ~255u); i += 256u) proj_vertices_item(P, i);
    __syncthreads();
    for (uint32_t t = tid; t < ((P.n_tris_in + 1u + 255u) & ~255u); t += 256u) clip_count_item(P, t);
    __syncthreads();
    // exclusive scan of P.append[0 .. n_tris_in] in place
    AppendCount carry = 0;
    for (uint32_t base = 0; base < P.n_tris_in + 1u; base += 256u) {
        const uint32_t i = base + tid;
        const AppendCount v = i <= P.n_tris_in ? P.append[i] : 0ull;
        AppendCount inc = v;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const AppendCount o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63u) wave_tot[wave] = inc;
        __syncthreads();
        AppendCount off = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            const AppendCount wt = wave_tot[w];
            if (w < wave) off += wt;
            total += wt;
        }
        if (i <= P.n_tris_in) P.append[i] = carry + off + inc - v;
        carry += total;
        __syncthreads();  // wave_tot is rewritten by the next round
    }
    if (tid == 0) P.chunk_base[0] = 0ull;
    __syncthreads();
    for (uint32_t t = tid; t < ((P.n_tris_in + 255u) & ~255u); t += 256u) clip_emit_item(P, t);   // (+ the meshes' live slot counts)
}

// The 40-byte records leave through LDS as the workgroup's contiguous 10 KB block (see k_setup3d in rxr_kernels.hip).
extern "C" __global__ void __launch_bounds__(256) k_proj_edges(ProjectParams P) {
    __shared__ uint2 xpose[256 * 5];
    const uint32_t t0 = blockIdx.x * blockDim.x, tid = threadIdx.x, s = t0 + tid;
    const uint32_t n_here = min(256u, P.n_tris_out - t0);   // the grid covers n_tris_out
    {
        // a workgroup whose slots all lie behind the live triangles of ONE mesh writes nothing: its consumers (k_setup3d, k_fill)
        // skip the same slots by the same rule, so stale records there are never read.  (wave-uniform: t0 comes from blockIdx)
        const uint32_t b0 = find_mesh(P.tout_prefix, P.n_meshes, t0), b1 = find_mesh(P.tout_prefix, P.n_meshes, t0 + n_here - 1u);
        if (b0 == b1 && t0 - P.tout_prefix[b0] >= P.mesh_live[b0]) return;
    }
    rxr_edges E;
#pragma unroll
    for (int i = 0; i < 3; ++i) E.a[i] = E.b[i] = E.c[i] = 0.0f;
    E.visible = 0;
    if (s < P.n_tris_out) E = edges_of_slot(P, s);
    uint2 rec[5];
    __builtin_memcpy(rec, &E, sizeof(E));
#pragma unroll
    for (int i = 0; i < 5; ++i) xpose[tid * 5u + i] = rec[i];
    __syncthreads();
    uint2 *dst = reinterpret_cast<uint2 *>(P.edges + t0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint32_t k = (uint32_t)i * 256u + tid;
        if (k < n_here * 5u) dst[k] = xpose[k];
    }
}

// ---- host-callable launchers ------------------------------------------------------------------------
extern "C" void rxr_launch_proj_static(const ProjectParams *P, hipStream_t s) {
    uint32_t n = P->n_verts_in > P->n_tris_in ? P->n_verts_in : P->n_tris_in;
    if (n == 0) return;
    RXR_LAUNCH(k_proj_static, dim3((n + 255u) / 256u), dim3(256), s, *P);
}
extern "C" void rxr_launch_project(const ProjectParams *P, hipStream_t s) {
    if (P->n_meshes == 0) return;
    static const bool small_ok = !(getenv("RXR_PROJ_SMALL") && atoi(getenv("RXR_PROJ_SMALL")) == 0);
    if (small_ok && P->n_verts_in <= RXR_PROJ_SMALL_MAX && P->n_tris_in < RXR_PROJ_SMALL_MAX && P->n_meshes <= RXR_PROJ_SMALL_MAX) {
        RXR_LAUNCH(k_proj_small, dim3(1), dim3(256), s, *P);
        if (P->n_tris_out && !P->edges_in_setup) RXR_LAUNCH(k_proj_edges, dim3((P->n_tris_out + 255u) / 256u), dim3(256), s, *P);
        return;
    }
    RXR_LAUNCH(k_proj_init, dim3((P->n_meshes + 255u) / 256u), dim3(256), s, *P);
    if (P->n_verts_in) RXR_LAUNCH(k_proj_vertices, dim3((P->n_verts_in + 255u) / 256u), dim3(256), s, *P);
    uint32_t nt1 = P->n_tris_in + 1u;
    RXR_LAUNCH(k_clip_count, dim3((nt1 + 255u) / 256u), dim3(256), s, *P);
    RXR_LAUNCH(k_proj_scan, dim3((nt1 + RXR_PROJ_SCAN_CHUNK - 1u) / RXR_PROJ_SCAN_CHUNK), dim3(256), s, *P);
    if (P->n_tris_in) RXR_LAUNCH(k_clip_emit, dim3((P->n_tris_in + 255u) / 256u), dim3(256), s, *P);
    if (P->n_tris_out && !P->edges_in_setup) RXR_LAUNCH(k_proj_edges, dim3((P->n_tris_out + 255u) / 256u), dim3(256), s, *P);
}
// the Edges pool alone (rxr_read_projected_mesh on a frame whose set-up built the records itself)
extern "C" void rxr_launch_proj_edges(const ProjectParams *P, hipStream_t s) {
    if (P->n_tris_out) RXR_LAUNCH(k_proj_edges, dim3((P->n_tris_out + 255u) / 256u), dim3(256), s, *P);
}

// =================================================================================================
// The 2D half (row N1): Batch2D::project on the device, and the Prim2D records of the raster kernels' 2D pass built from it
// (what rxr_upload_frame builds on the host for host-projected batches: same expressions, same order of operations).
// =================================================================================================
#include "rxr_device.h"
namespace {
// vek Mat3 * Vec3(x, y, 1) (include/rusterix_vek.hpp): first column multiplied, the others accumulated (fused like Mat4 * Vec4)
__device__ __forceinline__ float2 project2d(const Project2DParams &P, float2 v) {
    if (!P.has_matrix) return v;
    float2 o;
    o.x = madd(P.m[6], 1.0f, madd(P.m[3], v.y, P.m[0] * v.x));
    o.y = madd(P.m[7], 1.0f, madd(P.m[4], v.y, P.m[1] * v.x));
    return o;
}
__device__ __forceinline__ float dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e); }
// `x as usize` after the clamp against the screen (rasterizer.rs:631-634)
__device__ __forceinline__ uint32_t sat_px(float x, uint32_t hi) {
    if (!(x > 0.0f)) return 0u;
    if (x >= (float)hi) return hi;
    return (uint32_t)x;
}
// Rust `x as isize` narrowed to i32 for the Bresenham end points (:1785-1788); false beyond +-2^30
__device__ __forceinline__ bool to_isize32(float x, int32_t &out) {
    if (!(x == x)) {  // (NaN: refused like a coordinate out of range -- rxr_api.hip to_isize32 says why)
        out = 0;
        return false;
    }
    if (x <= -1073741824.0f || x >= 1073741824.0f) return false;
    out = (int32_t)x;
    return true;
}
}  // namespace

namespace {
__device__ __forceinline__ void proj2d_init_item(const Project2DParams &P, uint32_t i) {
    if (i < P.n_meshes) {
        P.bbox[i].min_x = P.bbox[i].min_y = 0xFFFFFFFFu;
        P.bbox[i].max_x = P.bbox[i].max_y = 0u;
    }
    if (i == 0u) {
        P.d2_box[0] = 0xFFFFu; P.d2_box[1] = 0u; P.d2_box[2] = 0xFFFFu; P.d2_box[3] = 0u;
    }
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_proj2d_init(Project2DParams P) { proj2d_init_item(P, blockIdx.x * 256u + threadIdx.x); }

// bounding box per batch (batch2d.rs:377-403): min / max of the projected vertices with NaN dropped
namespace {
__device__ __forceinline__ void proj2d_bbox_item(const Project2DParams &P, uint32_t i) {  // (whole waves, consecutive items)
    const bool active = i < P.n_verts;
    uint32_t m = 0;
    float2 p = make_float2(0.0f, 0.0f);
    if (active) {
        m = find_mesh_wave(P.vin_prefix, P.n_meshes, i);
        p = project2d(P, P.obj_verts[i]);
    }
    if (__ballot(active)) bbox_add_wave(P.bbox, m, active, p.x, p.y);
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_proj2d_bbox(Project2DParams P) { proj2d_bbox_item(P, blockIdx.x * 256u + threadIdx.x); }

namespace {
__device__ __forceinline__ void proj2d_prims_item(const Project2DParams &P, uint32_t i) {  // (whole waves: the union box is reduced across the wave)
    uint32_t min_x = 0, max_x = 0, min_y = 0, max_y = 0;
    if (i < P.n_prims) {
        const Prim2DSrc s = P.src[i];
        const DevMesh2D M = P.meshes[s.mesh];
        // the batch-level box reject with pad 0.5 against the whole screen (rasterizer.rs:594-600): Rect {x, y, width = max - min, ..}
        const DevBBox bb = P.bbox[s.mesh];
        const float bx = dec(bb.min_x), by = dec(bb.min_y), bw = dec(bb.max_x) - bx, bh = dec(bb.max_y) - by;
        const float pad = 0.5f;
        const bool keep = bx < P.width + pad && (bx + bw) > 0.0f - pad && by < P.height + pad && (by + bh) > 0.0f - pad;
        Prim2D T;
        __builtin_memset(&T, 0, sizeof(T));
        if (M.mode == RXR_MODE_TRIANGLES) {
            const float2 v0 = project2d(P, P.obj_verts[M.vin_base + s.ia]), v1 = project2d(P, P.obj_verts[M.vin_base + s.ib]),
                         v2 = project2d(P, P.obj_verts[M.vin_base + s.ic]);
            // Edges::new([v0, v1, v2], [v1, v2, v0]) (batch2d.rs:413-421, edge.rs:17-21)
            const float px[3] = {v0.x, v1.x, v2.x}, py[3] = {v0.y, v1.y, v2.y}, qx[3] = {v1.x, v2.x, v0.x}, qy[3] = {v1.y, v2.y, v0.y};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                T.ea[k] = qy[k] - py[k];
                T.eb[k] = px[k] - qx[k];
                T.ec[k] = qx[k] * py[k] - qy[k] * px[k];
            }
            T.v0x = v0.x; T.v0y = v0.y; T.v1x = v1.x; T.v1y = v1.y; T.v2x = v2.x; T.v2y = v2.y;
            const float2 u0 = P.obj_uvs[M.vin_base + s.ia], u1 = P.obj_uvs[M.vin_base + s.ib], u2 = P.obj_uvs[M.vin_base + s.ic];
            T.u0 = u0.x; T.v0 = u0.y; T.u1 = u1.x; T.v1 = u1.y; T.u2 = u2.x; T.v2 = u2.y;
            T.batch_kind = (s.mesh << 2) | 1u;
            if (keep) {
                const float min_xf = fminf(v0.x, fminf(v1.x, v2.x)), max_xf = fmaxf(v0.x, fmaxf(v1.x, v2.x));
                const float min_yf = fminf(v0.y, fminf(v1.y, v2.y)), max_yf = fmaxf(v0.y, fmaxf(v1.y, v2.y));
                min_x = sat_px(fmaxf(floorf(min_xf), 0.0f), 0xFFFFu);
                max_x = sat_px(fminf(ceilf(max_xf), P.width), 0xFFFFu);
                min_y = sat_px(fmaxf(floorf(min_yf), 0.0f), 0xFFFFu);
                max_y = sat_px(fminf(ceilf(max_yf), P.height), 0xFFFFu);
            }
        } else {
            const float2 a = project2d(P, P.obj_verts[M.vin_base + s.ia]), b = project2d(P, P.obj_verts[M.vin_base + s.ib]);
            int32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
            const bool ok = to_isize32(a.x, x0) && to_isize32(a.y, y0) && to_isize32(b.x, x1) && to_isize32(b.y, y1);
            if (keep && !ok) *P.bad_line = 1u;  // (the host builder refuses such a frame: RXR_ERR_UNSUPPORTED at rxr_synchronize)
            T.v0x = __int_as_float(x0); T.v0y = __int_as_float(y0); T.v1x = __int_as_float(x1); T.v1y = __int_as_float(y1);
            T.v2x = __uint_as_float(M.line_color);
            T.batch_kind = (s.mesh << 2) | 2u | 1u;
            if (keep && ok) {
                // the walk never leaves the end-point box (the last point is not plotted, :1800)
                const long long lx0 = min(x0, x1), lx1 = (long long)max(x0, x1) + 1, ly0 = min(y0, y1), ly1 = (long long)max(y0, y1) + 1;
                const long long W = (long long)P.width, H = (long long)P.height;  // (whole numbers: the frame size)
                min_x = (uint32_t)min(max(lx0, 0ll), W); max_x = (uint32_t)min(max(lx1, 0ll), W);
                min_y = (uint32_t)min(max(ly0, 0ll), H); max_y = (uint32_t)min(max(ly1, 0ll), H);
            }
        }
        if (keep && rxr_box_is_risky(bx, by, bw, bh)) {   // only inside the reference's tiles that pass its batch box test (rxr_device.h)
            uint32_t x0, x1, y0, y1;
            rxr_ref_tile_span(bx, bw, (uint32_t)P.width, P.ref_tile, pad, x0, x1);
            rxr_ref_tile_span(by, bh, (uint32_t)P.height, P.ref_tile, pad, y0, y1);
            min_x = max(min_x, x0); max_x = min(max_x, x1); min_y = max(min_y, y0); max_y = min(max_y, y1);
        }
        if (!(min_x < max_x && min_y < max_y)) min_x = max_x = min_y = max_y = 0u;
        T.bx = min_x | (max_x << 16);
        T.by = min_y | (max_y << 16);
        P.out[i] = T;
    }
    // union of the non-empty boxes: tiles outside it skip the 2D pass (RasterParams.d2_box_dev)
    const bool some = min_x < max_x && min_y < max_y;
    uint32_t a = some ? min_x : 0xFFFFu, b = some ? max_x : 0u, c = some ? min_y : 0xFFFFu, d = some ? max_y : 0u;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        a = min(a, (uint32_t)__shfl_xor((int)a, k, 64));
        b = max(b, (uint32_t)__shfl_xor((int)b, k, 64));
        c = min(c, (uint32_t)__shfl_xor((int)c, k, 64));
        d = max(d, (uint32_t)__shfl_xor((int)d, k, 64));
    }
    if ((threadIdx.x & 63u) == 0u && a < b && c < d) {
        atomicMin(&P.d2_box[0], a);
        atomicMax(&P.d2_box[1], b);
        atomicMin(&P.d2_box[2], c);
        atomicMax(&P.d2_box[3], d);
    }
}
}  // namespace
extern "C" __global__ void __launch_bounds__(256) k_proj2d_prims(Project2DParams P) { proj2d_prims_item(P, blockIdx.x * 256u + threadIdx.x); }

// few 2D batches (a HUD, a logo): the three steps in one workgroup, as k_proj_small
extern "C" __global__ void __launch_bounds__(256) k_proj2d_small(Project2DParams P) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < ((P.n_meshes + 1u + 255u) & ~255u); i += 256u) proj2d_init_item(P, i);   // (item 0 also resets the union box)
    __syncthreads();
    for (uint32_t i = tid; i < ((P.n_verts + 255u) & ~255u); i += 256u) proj2d_bbox_item(P, i);
    __syncthreads();
    for (uint32_t i = tid; i < ((P.n_prims + 255u) & ~255u); i += 256u) proj2d_prims_item(P, i);
}

extern "C" void rxr_launch_project2d(const Project2DParams *P, hipStream_t s) {
    static const bool small_ok = !(getenv("RXR_PROJ_SMALL") && atoi(getenv("RXR_PROJ_SMALL")) == 0);
    if (small_ok && P->n_meshes <= RXR_PROJ_SMALL_MAX && P->n_verts <= RXR_PROJ_SMALL_MAX && P->n_prims <= RXR_PROJ_SMALL_MAX) {
        RXR_LAUNCH(k_proj2d_small, dim3(1), dim3(256), s, *P);
        return;
    }
    RXR_LAUNCH(k_proj2d_init, dim3((P->n_meshes + 255u) / 256u + 1u), dim3(256), s, *P);
    if (P->n_verts) RXR_LAUNCH(k_proj2d_bbox, dim3((P->n_verts + 255u) / 256u), dim3(256), s, *P);
    if (P->n_prims) RXR_LAUNCH(k_proj2d_prims, dim3((P->n_prims + 255u) / 256u), dim3(256), s, *P);
}
