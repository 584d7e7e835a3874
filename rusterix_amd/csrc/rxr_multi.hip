// rxr_multi.hip -- multi-device contexts behind the C ABI (rxr_create_multi, include/rxr.h).
//
// The reference treats tiles as independent units and only concatenates them at the end (src/rasterizer.rs:273-275,
// :559-579).  A multi-device context does the same across the GPUs of one node, inside the library, so that a caller
// of rasterize() gets N GPUs without knowing about them:
//
//   * one plain context per device (its own streams, resident textures / meshes / programs, frame blob, scratch), all
//     driven from ONE process: a host worker thread per member issues that member's launches, so the N devices are fed
//     in parallel and every thread keeps its own current HIP device;
//   * the frame is sharded by interleaved 16-row stripes: member i of N renders stripes i, i+N, ... into a compact
//     buffer (rxr_render_stripes_to) -- the interleave spreads cheap sky rows and expensive floor rows evenly;
//   * host consumers (rxr_rasterize / rxr_render_download): EVERY device copies its own stripes straight into the
//     caller's `pixels` -- N PCIe links in parallel instead of one gather over xGMI followed by one 33 MB download;
//   * device consumers (rxr_render_gather): the members push their stripes to the root device's frame with peer copies
//     over xGMI (one link per source, all concurrent; the stripes land at their final place, no de-interleave pass) and
//     the root renders its own stripes directly into the frame.  A process-per-GPU host does the same exchange with
//     RCCL (rusterix_amd/distributed.py); inside one process the DMA engines do it without occupying compute units.
//
// The same device may be listed more than once (N logical members on one GPU): that is how the single-GPU tests check
// that an N-member frame is byte-identical to the single launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "rxr_ctx.h"

namespace {

// one host thread per member: launches for different devices are issued concurrently and each thread keeps its device current
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = RXR_OK;

    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> j = std::move(job);
            has_job = false;
            lk.unlock();
            const int r = j();
            lk.lock();
            rc = r;
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};

}  // namespace

struct rxr_group {
    std::vector<rxr_ctx *> members;
    std::vector<Worker *> workers;
    bool bands = false;        // RXR_MULTI_SHARD=bands: contiguous row bands instead of interleaved stripes (A-B runs)
    bool copy_1d = false;      // RXR_MULTI_COPY=1d: one copy per stripe instead of one strided 2-D copy per member (A-B runs)
    bool rendered = false;
    // the destination of the last rxr_render_gather: a member whose share is rendered again by rxr_synchronize (a bin list or a block of
    // k_blockscan overflowed) must ship it again, or the root's frame keeps the incomplete stripes
    struct {
        bool active = false;
        int root = 0;
        uint8_t *frame = nullptr;
        std::vector<uint32_t> rerenders;  // per member, when its share was shipped
    } gather;
};

namespace {

// runs fn(i) for every member on that member's worker thread; the first failure (lowest index) is reported on the handle
int run_all(rxr_ctx *ctx, const std::function<int(uint32_t)> &fn) {
    rxr_group *g = ctx->group;
    const uint32_t n = (uint32_t)g->members.size();
    if (n == 1u) {
        const int rc = fn(0);
        if (rc != RXR_OK) ctx->err = g->members[0]->err;
        return rc;
    }
    for (uint32_t i = 0; i < n; ++i) g->workers[i]->post([&fn, i] { return fn(i); });
    int first = RXR_OK;
    for (uint32_t i = 0; i < n; ++i) {
        const int rc = g->workers[i]->wait();
        if (rc != RXR_OK && first == RXR_OK) {
            first = rc;
            ctx->err = "member " + std::to_string(i) + " (device " + std::to_string(g->members[i]->device) + "): " + g->members[i]->err;
        }
    }
    return first;
}

struct Shard {
    uint32_t width, height, n_stripes;
    size_t stripe_bytes;
};
bool shard_of(const rxr_group *g, Shard &sh) {
    const rxr_ctx *m0 = g->members[0];
    if (!m0->has_frame) return false;
    sh.width = m0->P.width;
    sh.height = m0->P.height;
    sh.n_stripes = (sh.height + RXR_TILE_H - 1u) / RXR_TILE_H;
    sh.stripe_bytes = (size_t)RXR_TILE_H * sh.width * 4u;
    return true;
}
// stripes of member i: i, i + n, ...
uint32_t local_stripes(const Shard &sh, uint32_t i, uint32_t n) { return i < sh.n_stripes ? (sh.n_stripes - i + n - 1u) / n : 0u; }
// band mode: tile rows [t0, t1) of member i
void band_of(const Shard &sh, uint32_t i, uint32_t n, uint32_t &row0, uint32_t &row1) {
    const uint32_t t0 = (uint32_t)((uint64_t)sh.n_stripes * i / n), t1 = (uint32_t)((uint64_t)sh.n_stripes * (i + 1u) / n);
    row0 = std::min(sh.height, t0 * RXR_TILE_H);
    row1 = std::min(sh.height, t1 * RXR_TILE_H);
}

// member i renders its share into its own compact buffer, on its own stream (asynchronous)
int member_render(rxr_group *g, const Shard &sh, uint32_t i) {
    rxr_ctx *m = g->members[i];
    const uint32_t n = (uint32_t)g->members.size();
    HIPCHK(m, hipSetDevice(m->device));
    RenderSpec spec{};
    size_t bytes;
    if (g->bands) {
        uint32_t r0, r1;
        band_of(sh, i, n, r0, r1);
        spec.row0 = r0;
        spec.row1 = r1;
        spec.tile_y0 = r0 / RXR_TILE_H;
        spec.tile_stride = 1;
        spec.tiles_y = r1 > r0 ? (r1 + RXR_TILE_H - 1u) / RXR_TILE_H - spec.tile_y0 : 0u;
        spec.compact = false;
        spec.external = true;  // the buffer starts at row r0
        bytes = (size_t)(r1 - r0) * sh.width * 4u;
    } else {
        spec.row0 = 0;
        spec.row1 = sh.height;
        spec.tile_y0 = i;
        spec.tile_stride = n;
        spec.tiles_y = local_stripes(sh, i, n);
        spec.compact = true;
        spec.external = true;
        bytes = (size_t)spec.tiles_y * sh.stripe_bytes;
    }
    int rc = rxr_ensure(m, m->d_stripes, bytes ? bytes : 16);
    if (rc != RXR_OK) return rc;
    return rxr_render_spec(m, spec, m->d_stripes.p, m->stream);
}

// member i's rendered share -> its place in `dst` (a whole frame, row pitch width*4; host memory or, with `peer`, the
// root device's frame), queued on the member's stream behind its render
int member_ship(rxr_group *g, const Shard &sh, uint32_t i, uint8_t *dst, hipMemcpyKind kind) {
    rxr_ctx *m = g->members[i];
    const uint32_t n = (uint32_t)g->members.size();
    HIPCHK(m, hipSetDevice(m->device));
    if (g->bands) {
        uint32_t r0, r1;
        band_of(sh, i, n, r0, r1);
        if (r1 > r0)
            HIPCHK(m, hipMemcpyAsync(dst + (size_t)r0 * sh.width * 4u, m->d_stripes.p, (size_t)(r1 - r0) * sh.width * 4u, kind, m->stream));
        return RXR_OK;
    }
    const uint32_t cnt = local_stripes(sh, i, n);
    if (cnt == 0u) return RXR_OK;
    // the frame's last stripe may be short (height not a multiple of 16): it is the last local stripe of its owner
    const uint32_t last_frame_stripe = i + (cnt - 1u) * n;
    const uint32_t last_rows = std::min(sh.height - last_frame_stripe * RXR_TILE_H, (uint32_t)RXR_TILE_H);
    const uint32_t full = last_rows == RXR_TILE_H ? cnt : cnt - 1u;
    const uint8_t *src = (const uint8_t *)m->d_stripes.p;
    if (g->copy_1d) {
        for (uint32_t j = 0; j < full; ++j)
            HIPCHK(m, hipMemcpyAsync(dst + (size_t)(i + j * n) * sh.stripe_bytes, src + (size_t)j * sh.stripe_bytes, sh.stripe_bytes, kind, m->stream));
    } else if (full) {
        // local stripe j is one contiguous run of stripe_bytes in both buffers: a 2-D copy whose "rows" are whole stripes
        HIPCHK(m, hipMemcpy2DAsync(dst + (size_t)i * sh.stripe_bytes, (size_t)n * sh.stripe_bytes, src, sh.stripe_bytes, sh.stripe_bytes, full, kind, m->stream));
    }
    if (full < cnt)
        HIPCHK(m, hipMemcpyAsync(dst + (size_t)last_frame_stripe * sh.stripe_bytes, src + (size_t)full * sh.stripe_bytes, (size_t)last_rows * sh.width * 4u, kind,
                                 m->stream));
    return RXR_OK;
}

}  // namespace

extern "C" {

int rxr_create_multi(rxr_ctx **out, const int *device_ids, int n_devices) {
    if (!out) return rxr_fail(nullptr, RXR_ERR_INVALID, "rxr_create_multi: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices <= 0 || n_devices > 64) return rxr_fail(nullptr, RXR_ERR_INVALID, "rxr_create_multi: need 1..64 device ids");
    rxr_ctx *h = new rxr_ctx();
    h->group = new rxr_group();
    h->device = device_ids[0];
    rxr_group *g = h->group;
    for (int i = 0; i < n_devices; ++i) {
        rxr_ctx *m = nullptr;
        const int rc = rxr_create(&m, device_ids[i]);
        if (rc != RXR_OK) {
            const std::string msg = std::string("rxr_create_multi: device ") + std::to_string(device_ids[i]) + ": " + rxr_last_error(nullptr);
            rxr_group_destroy(h);
            return rxr_fail(nullptr, rc, msg);
        }
        g->members.push_back(m);
    }
    // peer access between every pair of distinct devices, so that rxr_render_gather can push stripes over xGMI.  Where
    // the platform refuses it the peer copies still work (staged by the runtime), only slower.
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < n_devices; ++j) {
            if (device_ids[i] == device_ids[j]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) != hipSuccess || !can) continue;
            if (hipSetDevice(device_ids[i]) != hipSuccess) continue;
            const hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
            if (e != hipSuccess) (void)hipGetLastError();  // hipErrorPeerAccessAlreadyEnabled included
        }
    if (const char *sh = getenv("RXR_MULTI_SHARD")) g->bands = strcmp(sh, "bands") == 0;
    if (const char *cp = getenv("RXR_MULTI_COPY")) g->copy_1d = strcmp(cp, "1d") == 0;
    if (n_devices > 1)
        for (int i = 0; i < n_devices; ++i) {
            Worker *w = new Worker();
            w->th = std::thread([w] { w->loop(); });
            g->workers.push_back(w);
        }
    *out = h;
    return RXR_OK;
}

int rxr_member_count(const rxr_ctx *ctx) { return !ctx ? 0 : (ctx->group ? (int)ctx->group->members.size() : 1); }

rxr_ctx *rxr_member(rxr_ctx *ctx, int index) {
    if (!ctx || index < 0) return nullptr;
    if (!ctx->group) return index == 0 ? ctx : nullptr;
    return index < (int)ctx->group->members.size() ? ctx->group->members[index] : nullptr;
}

int rxr_pin_host_buffer(rxr_ctx *ctx, void *ptr, size_t bytes) {
    if (!ctx || !ptr || bytes == 0) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_pin_host_buffer: NULL / empty buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(ptr, bytes, hipHostRegisterPortable));  // portable: pinned for every device of the process
    return RXR_OK;
}

int rxr_unpin_host_buffer(rxr_ctx *ctx, void *ptr) {
    if (!ctx || !ptr) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_unpin_host_buffer: NULL buffer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostUnregister(ptr));
    return RXR_OK;
}

int rxr_render_gather(rxr_ctx *ctx, int root, void *dev_pixels, void *hip_stream) {
    if (!ctx) return RXR_ERR_INVALID;
    if (!ctx->group) {
        // a plain context is its own root: the whole frame in one launch
        if (root != 0) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_gather: root out of range");
        if (!ctx->has_frame) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_gather: no frame uploaded");
        if (!dev_pixels) return rxr_render_rows(ctx, 0, ctx->P.height);
        return rxr_render_rows_to(ctx, 0, ctx->P.height, dev_pixels, hip_stream);
    }
    rxr_group *g = ctx->group;
    const uint32_t n = (uint32_t)g->members.size();
    if (root < 0 || (uint32_t)root >= n) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_gather: root out of range");
    Shard sh;
    if (!shard_of(g, sh)) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_gather: no frame uploaded");
    rxr_ctx *R = g->members[root];
    uint8_t *frame = (uint8_t *)(dev_pixels ? dev_pixels : R->d_fb.p);
    hipStream_t rs = hip_stream ? (hipStream_t)hip_stream : R->stream;
    // whatever the consumer stream has queued so far (readers of the PREVIOUS frame in this buffer) comes before the members' writes
    HIPCHK(ctx, hipSetDevice(R->device));
    HIPCHK(ctx, hipEventRecord(R->ev_band[5], rs));
    g->gather.active = false;
    g->gather.rerenders.assign(n, 0u);
    int rc = run_all(ctx, [&](uint32_t i) -> int {
        rxr_ctx *m = g->members[i];
        HIPCHK(m, hipSetDevice(m->device));
        if ((int)i == root && !g->bands) {
            // the root's own stripes go straight to their place in the frame (frame addressing with a tile-row stride)
            RenderSpec spec{};
            spec.row0 = 0;
            spec.row1 = sh.height;
            spec.tile_y0 = i;
            spec.tile_stride = n;
            spec.tiles_y = local_stripes(sh, i, n);
            spec.compact = false;
            spec.external = true;
            return rxr_render_spec(m, spec, frame, rs);
        }
        int r = member_render(g, sh, i);
        if (r != RXR_OK) return r;
        g->gather.rerenders[i] = m->rerenders;
        // push over xGMI on the SOURCE device's stream (one link per source, all concurrent); same-device members copy locally
        HIPCHK(m, hipStreamWaitEvent(m->stream, R->ev_band[5], 0));
        r = member_ship(g, sh, i, frame, hipMemcpyDeviceToDevice);
        if (r != RXR_OK) return r;
        HIPCHK(m, hipEventRecord(m->ev_band[4], m->stream));
        return RXR_OK;
    });
    if (rc != RXR_OK) return rc;
    // the frame is complete on the root's stream once every other member's stripes have landed
    HIPCHK(ctx, hipSetDevice(R->device));
    for (uint32_t i = 0; i < n; ++i) {
        if ((int)i == root && !g->bands) continue;
        hipError_t e = hipStreamWaitEvent(rs, g->members[i]->ev_band[4], 0);
        if (e != hipSuccess) return rxr_fail(ctx, RXR_ERR_HIP, std::string("rxr_render_gather: hipStreamWaitEvent: ") + hipGetErrorString(e));
    }
    g->gather.active = true;
    g->gather.root = root;
    g->gather.frame = frame;
    g->rendered = true;
    return RXR_OK;
}

}  // extern "C"

void rxr_group_destroy(rxr_ctx *ctx) {
    rxr_group *g = ctx->group;
    if (g) {
        for (Worker *w : g->workers) {
            {
                std::lock_guard<std::mutex> lk(w->m);
                w->quit = true;
                w->cv.notify_all();
            }
            if (w->th.joinable()) w->th.join();
            delete w;
        }
        for (rxr_ctx *m : g->members) rxr_destroy(m);
        delete g;
    }
    ctx->group = nullptr;
    delete ctx;
}

int rxr_group_set_textures(rxr_ctx *ctx, const rxr_tile *static_tiles, uint32_t n_static, const rxr_tile *dynamic_tiles, uint32_t n_dynamic) {
    return run_all(ctx, [&](uint32_t i) { return rxr_set_textures(ctx->group->members[i], static_tiles, n_static, dynamic_tiles, n_dynamic); });
}

int rxr_group_set_meshes(rxr_ctx *ctx, const rxr_mesh3d *meshes, uint32_t n_meshes) {
    return run_all(ctx, [&](uint32_t i) { return rxr_set_meshes(ctx->group->members[i], meshes, n_meshes); });
}

int rxr_group_set_meshes2d(rxr_ctx *ctx, const rxr_mesh2d *meshes, uint32_t n_meshes) {
    return run_all(ctx, [&](uint32_t i) { return rxr_set_meshes2d(ctx->group->members[i], meshes, n_meshes); });
}

int rxr_group_set_projection2d(rxr_ctx *ctx, const float *mat3) {
    return run_all(ctx, [&](uint32_t i) { return rxr_set_projection2d(ctx->group->members[i], mat3); });
}

int rxr_group_set_shaders(rxr_ctx *ctx, const rxr_shader_set *set) {
    return run_all(ctx, [&](uint32_t i) { return rxr_set_shaders(ctx->group->members[i], set); });
}

// the (small or replicated) scene goes to every device: N uploads over N PCIe links, flattened by N host threads
int rxr_group_upload_frame(rxr_ctx *ctx, const rxr_frame *frame) {
    ctx->group->rendered = false;
    ctx->group->gather.active = false;
    return run_all(ctx, [&](uint32_t i) { return rxr_upload_frame(ctx->group->members[i], frame); });
}

int rxr_group_render(rxr_ctx *ctx) {
    rxr_group *g = ctx->group;
    Shard sh;
    if (!shard_of(g, sh)) return rxr_fail(ctx, RXR_ERR_INVALID, "render: no frame uploaded");
    g->gather.active = false;
    const int rc = run_all(ctx, [&](uint32_t i) { return member_render(g, sh, i); });
    if (rc == RXR_OK) g->rendered = true;
    return rc;
}

int rxr_group_download(rxr_ctx *ctx, uint8_t *pixels) {
    rxr_group *g = ctx->group;
    Shard sh;
    if (!shard_of(g, sh) || !g->rendered) return rxr_fail(ctx, RXR_ERR_INVALID, "download: nothing rendered");
    return run_all(ctx, [&](uint32_t i) -> int {
        rxr_ctx *m = g->members[i];
        int rc = rxr_synchronize(m);
        if (rc != RXR_OK) return rc;
        if ((rc = member_ship(g, sh, i, pixels, hipMemcpyDeviceToHost)) != RXR_OK) return rc;
        HIPCHK(m, hipStreamSynchronize(m->stream));
        return RXR_OK;
    });
}

// render + download in one pass per member: the copy is queued right behind the render; only a bin-list overflow (which
// rxr_synchronize answers by rendering the member's share again) makes it ship twice
int rxr_group_render_download(rxr_ctx *ctx, uint8_t *pixels) {
    rxr_group *g = ctx->group;
    Shard sh;
    if (!shard_of(g, sh)) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_download: no frame uploaded");
    g->gather.active = false;
    const int rc = run_all(ctx, [&](uint32_t i) -> int {
        rxr_ctx *m = g->members[i];
        int r = member_render(g, sh, i);
        if (r != RXR_OK) return r;
        const uint32_t before = m->rerenders;
        if ((r = member_ship(g, sh, i, pixels, hipMemcpyDeviceToHost)) != RXR_OK) return r;
        if ((r = rxr_synchronize(m)) != RXR_OK) return r;
        if (m->rerenders != before) {
            if ((r = member_ship(g, sh, i, pixels, hipMemcpyDeviceToHost)) != RXR_OK) return r;
            HIPCHK(m, hipStreamSynchronize(m->stream));
        }
        return RXR_OK;
    });
    if (rc == RXR_OK) g->rendered = true;
    return rc;
}

// Frames in flight on ONE device ("lanes"): a group whose members all name the same device.  Frame k of the batch is rendered by
// member k mod L on that member's own stream -- every member has its own resident frame, records and bins, so consecutive frames
// share nothing and overlap on the GPU: the tail of one launch sequence (few workgroups left, the chip mostly idle) runs under the
// head of the next.  Measured on one MI355X (profiles/r03/share8_*.json): a 1/8 share of the 4K bench frame costs 31 us per frame
// back to back on one stream (16 us of it fixed: the set-up launch and the raster kernel's ramp and tail) and 17.5 us with two
// lanes.  The call forks the lanes' streams from the caller's stream and joins them back with events; everything is issued from
// the calling thread (one device: no worker threads).
int rxr_group_render_stripes_batch(rxr_ctx *ctx, uint32_t first, uint32_t stride, uint32_t n_frames, void *dev_pixels, size_t frame_stride_bytes,
                                   void *hip_stream) {
    rxr_group *g = ctx->group;
    const uint32_t L = (uint32_t)g->members.size();
    for (rxr_ctx *m : g->members)
        if (m->device != g->members[0]->device)
            return rxr_fail(ctx, RXR_ERR_UNSUPPORTED, "rxr_render_stripes_batch on a context over several devices: device pointers and streams belong to ONE device "
                                                      "(the batch call serves a plain context or members that all sit on one device)");
    Shard sh;
    if (!shard_of(g, sh)) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_batch: no frame uploaded");
    if (stride == 0) return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_batch: stride 0");
    if (n_frames == 0) return RXR_OK;
    const uint32_t cnt = first < sh.n_stripes ? (sh.n_stripes - first + stride - 1u) / stride : 0u;
    if (n_frames > 1u && frame_stride_bytes < (size_t)cnt * sh.stripe_bytes)
        return rxr_fail(ctx, RXR_ERR_INVALID, "rxr_render_stripes_batch: frame_stride_bytes is smaller than one compact stripe buffer");
    rxr_ctx *m0 = g->members[0];
    HIPCHK(ctx, hipSetDevice(m0->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m0->stream;
    const uint32_t used = std::min(L, n_frames);
    // fork: the lanes start behind what the caller has queued (e.g. the consumer of the buffers they are about to overwrite)
    HIPCHK(ctx, hipEventRecord(m0->ev_band[6], s));
    for (uint32_t l = 0; l < used; ++l)
        if (g->members[l]->stream != s) HIPCHK(ctx, hipStreamWaitEvent(g->members[l]->stream, m0->ev_band[6], 0));
    for (uint32_t k = 0; k < n_frames; ++k) {
        rxr_ctx *m = g->members[k % L];
        const int rc = rxr_render_stripes_to(m, first, stride, (uint8_t *)dev_pixels + (size_t)k * frame_stride_bytes, m->stream);
        if (rc != RXR_OK) {
            ctx->err = "lane " + std::to_string(k % L) + ": " + m->err;
            return rc;
        }
    }
    // join: the batch is complete on the caller's stream
    for (uint32_t l = 0; l < used; ++l) {
        rxr_ctx *m = g->members[l];
        if (m->stream == s) continue;
        HIPCHK(ctx, hipEventRecord(m->ev_band[7], m->stream));
        HIPCHK(ctx, hipStreamWaitEvent(s, m->ev_band[7], 0));
    }
    g->rendered = true;
    return RXR_OK;
}

int rxr_group_synchronize(rxr_ctx *ctx) {
    rxr_group *g = ctx->group;
    Shard sh{};
    const bool gathered = g->gather.active && shard_of(g, sh);
    return run_all(ctx, [&](uint32_t i) -> int {
        rxr_ctx *m = g->members[i];
        int rc = rxr_synchronize(m);
        if (rc != RXR_OK) return rc;
        // rxr_render_gather: a share that was rendered again (its lists had overflowed) reaches the root's frame only now
        if (gathered && !((int)i == g->gather.root && !g->bands) && m->rerenders != g->gather.rerenders[i]) {
            g->gather.rerenders[i] = m->rerenders;
            if ((rc = member_ship(g, sh, i, g->gather.frame, hipMemcpyDeviceToDevice)) != RXR_OK) return rc;
            HIPCHK(m, hipStreamSynchronize(m->stream));
        }
        return RXR_OK;
    });
}

int rxr_group_get_stats(rxr_ctx *ctx, rxr_stats *out) {
    rxr_group *g = ctx->group;
    *out = g->members[0]->stats;
    out->n_bin_entries = 0;
    out->tiles_y = 0;
    for (rxr_ctx *m : g->members) {
        out->n_bin_entries += m->stats.n_bin_entries;
        out->tiles_y += m->stats.tiles_y;
        out->setup_us = std::max(out->setup_us, m->stats.setup_us);
        out->raster_us = std::max(out->raster_us, m->stats.raster_us);
        out->total_us = std::max(out->total_us, m->stats.total_us);
    }
    return RXR_OK;
}
