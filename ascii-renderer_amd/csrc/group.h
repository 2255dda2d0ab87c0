// group.h - several GPUs in one process: a context and a host thread per device, row strips, RCCL all-gathers of the ASCII strips
// Part of libascii_rast.so; included by ascii_rast.hip (one translation unit), not compiled on its own.
//
// The frame is cut into one strip of rows per device (ar_strip_rows: heights differ by at most one row, so any number of
// rows works with any number of devices).  Every device holds the whole mesh and renders its strip with the ordinary
// per-context calls (ar_set_strip, ar_set_ascii_target, ar_render_async), each from a host thread of its own - five kernel
// launches per frame and device are more than one thread can issue for eight devices at the rate the GPUs render.
// A member hands the frames posted to it to its GPU up to sixteen at a time, as one chain of kernel launches
// (ar_render_batch_async): a strip of a frame is bound by launches and hand-overs, not by work.
// A strip lands in a slot of a ring (slots as tall as the tallest strip); every `batch` frames ONE all-gather per device
// combines the rings, so that afterwards every device holds every strip of those frames.  There are two rings: while one is
// gathered the next frames fill the other; a frame waits (on the GPU) only for the gather that last used its ring, the gather
// only for the frames that filled its ring, and the host threads wait for neither.
// RCCL is loaded with dlopen when the first group is created: a process that never makes one does not load it.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <thread>

namespace {

struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;

// Loads RCCL once per process; false (message in g_create_error) when it is not there.
bool load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return true;
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) { fail(nullptr, AR_E_STATE, "RCCL not found: %s", dlerror()); return false; }
    RcclApi a;
    a.lib = lib;
    a.comm_init_all = reinterpret_cast<decltype(a.comm_init_all)>(dlsym(lib, "ncclCommInitAll"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
    a.all_gather = reinterpret_cast<decltype(a.all_gather)>(dlsym(lib, "ncclAllGather"));
    a.group_start = reinterpret_cast<decltype(a.group_start)>(dlsym(lib, "ncclGroupStart"));
    a.group_end = reinterpret_cast<decltype(a.group_end)>(dlsym(lib, "ncclGroupEnd"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(lib, "ncclGetErrorString"));
    if (!a.comm_init_all || !a.comm_destroy || !a.all_gather || !a.group_start || !a.group_end || !a.error_string) {
        fail(nullptr, AR_E_STATE, "RCCL library lacks an entry point this library needs");
        dlclose(lib);
        return false;
    }
    g_rccl = a;
    return true;
}

constexpr int kGroupRings = 2;

struct GroupCommand {
    enum Kind { FRAME, FLUSH, STOP } kind = FRAME;
    int cols = 0, rows = 0;
    float proj[16], view[16];
};

}  // namespace

struct ar_group {
    struct Member {
        ar_group *group = nullptr;
        int index = 0, device = 0;
        ar_ctx *ctx = nullptr;
        ncclComm_t comm = nullptr;
        hipStream_t gather_stream = nullptr;
        uint8_t *send[kGroupRings] = {}, *recv[kGroupRings] = {};  // this device's strips of a batch / everybody's
        std::thread thread;
        std::mutex mutex;
        std::condition_variable wake, idle;
        std::deque<GroupCommand> queue;
        bool busy = false;
        int rc = 0;            // first failure of this member's thread
        char err[256] = "";
        // the ring bookkeeping, identical in every member (each advances it by the same commands); what the last gather
        // held may be asked for by the caller's thread while the member's thread is at work
        int cur = 0, filled = 0;
        std::atomic<int> last{-1}, gathered{0};
        // frames posted but not yet enqueued on the GPU: they go out together, as one chain of launches (ar_render_batch_async)
        std::vector<float> pending;  // 32 floats (projection, view) per frame
        int pending_cols = 0, pending_rows = 0;
    };
    int n = 0, batch = 1;
    int chain = 16;                // frames a member hands to its GPU per call (one chain of launches): at most this many
    int cols = 0, rows = 0;        // grid the rings are sized for (0: none yet)
    size_t slot_bytes = 0;         // one strip slot: tallest strip x cols
    std::vector<Member *> members;
    char err[256] = "";
};

namespace {

int group_fail(ar_group *g, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g ? g->err : g_create_error, 256, fmt, ap);
    va_end(ap);
    return code;
}

void member_fail(ar_group::Member *m, int code, const char *what, const char *detail) {
    if (m->rc) return;  // keep the first
    m->rc = code;
    snprintf(m->err, sizeof m->err, "device %d: %s: %s", m->device, what, detail ? detail : "");
}

// One all-gather of the ring just filled (cur), then on to the other ring.  The frames that follow reuse the ring of an
// EARLIER gather: they are made to wait for the gathers enqueued so far before this one is enqueued.
void member_gather(ar_group::Member *m) {
    ar_group *g = m->group;
    if (m->filled == 0) return;
    if (ar_frame_wait_stream(m->ctx, m->gather_stream)) { member_fail(m, AR_E_HIP, "ar_frame_wait_stream", ar_last_error(m->ctx)); return; }
    if (ar_stream_wait_frame(m->ctx, m->gather_stream)) { member_fail(m, AR_E_HIP, "ar_stream_wait_frame", ar_last_error(m->ctx)); return; }
    // (every member sends whole batches - slots a short last batch left unfilled travel as they are - so that all counts agree)
    const ncclResult_t r = g_rccl.all_gather(m->send[m->cur], m->recv[m->cur], size_t(g->batch) * g->slot_bytes, ncclUint8, m->comm, m->gather_stream);
    if (r != ncclSuccess) { member_fail(m, AR_E_HIP, "ncclAllGather", g_rccl.error_string(r)); return; }
    m->gathered = m->filled;
    m->last = m->cur;
    m->filled = 0;
    m->cur = (m->cur + 1) % kGroupRings;
}

// Enqueues the frames posted so far: one batch call, their strips into consecutive slots of the ring being filled.
void member_issue(ar_group::Member *m) {
    ar_group *g = m->group;
    const int n = int(m->pending.size() / 32);
    if (n == 0) return;
    int row0 = 0, row1 = 0;
    ar_strip_rows(m->pending_rows, g->n, m->index, &row0, &row1);
    if (row1 > row0) {  // (more devices than rows leave some without a strip: they only take part in the gathers)
        uint8_t *slot = m->send[m->cur] + size_t(m->filled) * g->slot_bytes;
        int rc = ar_set_strip(m->ctx, row0, row1);
        if (!rc) rc = ar_set_ascii_target(m->ctx, slot, row0);
        if (!rc) rc = ar_render_batch_async(m->ctx, m->pending_cols, m->pending_rows, n, m->pending.data(), g->slot_bytes);
        if (rc) member_fail(m, rc, "frames", ar_last_error(m->ctx));
    }
    m->pending.clear();
    m->filled += n;
    if (m->filled == g->batch) member_gather(m);
}

void member_frame(ar_group::Member *m, const GroupCommand &c) {
    ar_group *g = m->group;
    m->pending_cols = c.cols; m->pending_rows = c.rows;
    m->pending.insert(m->pending.end(), c.proj, c.proj + 16);
    m->pending.insert(m->pending.end(), c.view, c.view + 16);
    const int n = int(m->pending.size() / 32);
    if (m->filled + n == g->batch || n == g->chain) member_issue(m);  // the ring is full, or a chain's worth has come together
}

void member_loop(ar_group::Member *m) {
    (void)hipSetDevice(m->device);
    for (;;) {
        GroupCommand c;
        {
            std::unique_lock<std::mutex> lock(m->mutex);
            m->busy = false;
            if (m->queue.empty()) m->idle.notify_all();
            m->wake.wait(lock, [&] { return !m->queue.empty(); });
            c = m->queue.front();
            m->queue.pop_front();
            m->busy = true;
        }
        if (c.kind == GroupCommand::STOP) return;
        if (m->rc) continue;  // after a failure the rest of the queue is dropped
        if (c.kind == GroupCommand::FLUSH) { member_issue(m); member_gather(m); }
        else member_frame(m, c);
    }
}

void member_post(ar_group::Member *m, const GroupCommand &c) {
    {
        std::lock_guard<std::mutex> lock(m->mutex);
        m->queue.push_back(c);
    }
    m->wake.notify_one();
}

void member_drain(ar_group::Member *m) {
    std::unique_lock<std::mutex> lock(m->mutex);
    m->idle.wait(lock, [&] { return m->queue.empty() && !m->busy; });
}

// Waits for every member's queue, then for its GPU work; reports the first failure.
int group_quiesce(ar_group *g) {
    for (auto *m : g->members) member_drain(m);
    int rc = 0;
    for (auto *m : g->members) {
        if (m->rc && !rc) { rc = m->rc; snprintf(g->err, sizeof g->err, "%s", m->err); }
        if (hipSetDevice(m->device) != hipSuccess || ar_sync(m->ctx) != 0 || hipStreamSynchronize(m->gather_stream) != hipSuccess) {
            if (!rc) rc = group_fail(g, AR_E_HIP, "device %d: synchronisation failed: %s", m->device, ar_last_error(m->ctx));
        }
    }
    return rc;
}

void group_free_rings(ar_group *g) {
    for (auto *m : g->members) {
        (void)hipSetDevice(m->device);
        for (int r = 0; r < kGroupRings; r++) {
            (void)hipFree(m->send[r]); (void)hipFree(m->recv[r]);
            m->send[r] = m->recv[r] = nullptr;
        }
        m->cur = 0; m->filled = 0; m->last = -1; m->gathered = 0;
    }
    g->cols = g->rows = 0;
    g->slot_bytes = 0;
}

}  // namespace

extern "C" {

int ar_strip_rows(int rows, int n_strips, int strip, int *row0, int *row1) {
    if (rows < 0 || n_strips <= 0 || strip < 0 || strip >= n_strips || !row0 || !row1) return AR_E_ARG;
    const int base = rows / n_strips, rem = rows % n_strips;
    *row0 = strip * base + (strip < rem ? strip : rem);
    *row1 = *row0 + base + (strip < rem ? 1 : 0);
    return 0;
}

ar_group *ar_group_create(int n, const int *device_ids, int batch) {
    if (n <= 0 || n > 64 || !device_ids || batch <= 0 || batch > 4096) {
        group_fail(nullptr, AR_E_ARG, "ar_group_create: bad arguments (%d devices, batch %d)", n, batch);
        return nullptr;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (device_ids[i] == device_ids[j]) { group_fail(nullptr, AR_E_ARG, "ar_group_create: device %d named twice", device_ids[i]); return nullptr; }
    if (!load_rccl()) return nullptr;
    ar_group *g = new (std::nothrow) ar_group();
    if (!g) { group_fail(nullptr, AR_E_NOMEM, "out of host memory"); return nullptr; }
    g->n = n;
    g->batch = batch;
    g->chain = batch < 16 ? batch : 16;
    bool ok = true;
    for (int i = 0; i < n && ok; i++) {
        auto *m = new (std::nothrow) ar_group::Member();
        if (!m) { ok = false; group_fail(nullptr, AR_E_NOMEM, "out of host memory"); break; }
        g->members.push_back(m);
        m->group = g; m->index = i; m->device = device_ids[i];
        m->ctx = ar_create(device_ids[i]);  // (its message stays in place for the caller when it fails)
        ok = m->ctx != nullptr;
        if (ok && (hipSetDevice(m->device) != hipSuccess || hipStreamCreateWithFlags(&m->gather_stream, hipStreamNonBlocking) != hipSuccess)) {
            ok = false;
            group_fail(nullptr, AR_E_HIP, "ar_group_create: no stream on device %d", m->device);
        }
    }
    if (ok) {
        std::vector<ncclComm_t> comms(size_t(n), nullptr);
        const ncclResult_t r = g_rccl.comm_init_all(comms.data(), n, device_ids);
        if (r != ncclSuccess) { ok = false; group_fail(nullptr, AR_E_HIP, "ncclCommInitAll: %s", g_rccl.error_string(r)); }
        else for (int i = 0; i < n; i++) g->members[size_t(i)]->comm = comms[size_t(i)];
    }
    if (ok)
        for (auto *m : g->members) m->thread = std::thread(member_loop, m);
    if (!ok) { ar_group_destroy(g); return nullptr; }
    return g;
}

void ar_group_destroy(ar_group *g) {
    if (!g) return;
    for (auto *m : g->members)
        if (m->thread.joinable()) {
            GroupCommand stop;
            stop.kind = GroupCommand::STOP;
            member_post(m, stop);
            m->thread.join();
        }
    for (auto *m : g->members) {
        (void)hipSetDevice(m->device);
        if (m->ctx) (void)ar_sync(m->ctx);
        if (m->gather_stream) (void)hipStreamSynchronize(m->gather_stream);
    }
    for (auto *m : g->members)
        if (m->comm) (void)g_rccl.comm_destroy(m->comm);
    group_free_rings(g);
    for (auto *m : g->members) {
        (void)hipSetDevice(m->device);
        if (m->gather_stream) (void)hipStreamDestroy(m->gather_stream);
        if (m->ctx) ar_destroy(m->ctx);
        delete m;
    }
    delete g;
}

const char *ar_group_last_error(ar_group *g) { return g ? g->err : g_create_error; }

int ar_group_size(ar_group *g) { return g ? g->n : 0; }

ar_ctx *ar_group_ctx(ar_group *g, int member) {
    if (!g || member < 0 || member >= g->n) return nullptr;
    return g->members[size_t(member)]->ctx;
}

int ar_group_render_async(ar_group *g, int cols, int rows, const float proj[16], const float view[16]) {
    if (!g) return AR_E_ARG;
    if (cols <= 0 || rows <= 0 || !proj || !view || size_t(cols) * size_t(rows) > (size_t(1) << 31))
        return group_fail(g, AR_E_ARG, "ar_group_render_async: bad arguments (%d x %d)", cols, rows);
    if (cols != g->cols || rows != g->rows) {
        // a new grid: finish what was posted for the old one, then rings for the new strips (slots as tall as the tallest strip)
        int rc = g->cols ? ar_group_sync(g) : group_quiesce(g);
        if (rc) return rc;
        group_free_rings(g);
        const size_t tallest = size_t((rows + g->n - 1) / g->n);
        g->slot_bytes = tallest * size_t(cols);
        for (auto *m : g->members) {
            if (hipSetDevice(m->device) != hipSuccess) return group_fail(g, AR_E_HIP, "hipSetDevice(%d) failed", m->device);
            for (int r = 0; r < kGroupRings; r++) {
                if (hipMalloc(&m->send[r], size_t(g->batch) * g->slot_bytes) != hipSuccess ||
                    hipMalloc(&m->recv[r], size_t(g->n) * size_t(g->batch) * g->slot_bytes) != hipSuccess ||
                    hipMemset(m->send[r], 0, size_t(g->batch) * g->slot_bytes) != hipSuccess)
                    return group_fail(g, AR_E_NOMEM, "device %d: cannot allocate the gather rings", m->device);
            }
        }
        g->cols = cols; g->rows = rows;
    }
    GroupCommand c;
    c.kind = GroupCommand::FRAME;
    c.cols = cols; c.rows = rows;
    memcpy(c.proj, proj, sizeof c.proj);
    memcpy(c.view, view, sizeof c.view);
    for (auto *m : g->members) member_post(m, c);
    return 0;
}

int ar_group_sync(ar_group *g) {
    if (!g) return AR_E_ARG;
    GroupCommand c;
    c.kind = GroupCommand::FLUSH;
    for (auto *m : g->members) member_post(m, c);
    return group_quiesce(g);
}

int ar_group_frames_gathered(ar_group *g) {
    if (!g || g->members.empty()) return 0;
    return g->members[0]->last.load() < 0 ? 0 : g->members[0]->gathered.load();
}

int ar_group_read(ar_group *g, int member, int frame, uint8_t *ascii) {
    if (!g) return AR_E_ARG;
    if (!ascii || member < 0 || member >= g->n) return group_fail(g, AR_E_ARG, "ar_group_read: bad arguments");
    int rc = group_quiesce(g);
    if (rc) return rc;
    ar_group::Member *m = g->members[size_t(member)];
    if (m->last < 0 || frame < 0 || frame >= m->gathered)
        return group_fail(g, AR_E_STATE, "ar_group_read: frame %d of a last batch of %d", frame, m->last < 0 ? 0 : m->gathered.load());
    if (hipSetDevice(m->device) != hipSuccess) return group_fail(g, AR_E_HIP, "hipSetDevice(%d) failed", m->device);
    // recv is [device][frame of the batch][slot]: strip s of the frame sits at ((s * batch) + frame) * slot_bytes
    for (int s = 0; s < g->n; s++) {
        int row0 = 0, row1 = 0;
        ar_strip_rows(g->rows, g->n, s, &row0, &row1);
        if (row1 <= row0) continue;
        const uint8_t *src = m->recv[m->last] + (size_t(s) * size_t(g->batch) + size_t(frame)) * g->slot_bytes;
        if (hipMemcpy(ascii + size_t(row0) * size_t(g->cols), src, size_t(row1 - row0) * size_t(g->cols), hipMemcpyDeviceToHost) != hipSuccess)
            return group_fail(g, AR_E_HIP, "ar_group_read: copy from device %d failed", m->device);
    }
    return 0;
}

}  // extern "C"
