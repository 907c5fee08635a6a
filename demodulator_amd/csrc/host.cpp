// host.cpp -- the host-buffer paths of the C ABI (include/demod_hip.h): what the
// reference's main thread and consumer thread do around the per-chunk pipeline
// (/root/reference/src/main.c:58-98, src/matrix.c:236-273: fread into the hand-off
// buffer, memcpy out of it, fwrite of the results), restated for a device behind PCIe.
//
// Three forms, one mechanism -- a three-deep pipeline whose stages run on three HIP
// streams (upload, hot path, download) chained by events, so that the upload of piece
// i+1, the kernels of piece i and the download of piece i-1 overlap:
//
//   demod_process_host     a whole buffer.  Pinned caller memory (demod_alloc_pinned) is
//                          DMA'd in place; pageable memory is staged through pinned
//                          buffers by a small pool of copy threads (copy-in of piece i+1
//                          and copy-out of piece i-2 run together).
//   demod_stream_submit /  hand-off at a time, results in order (the seamless backend
//   _collect               and chunks below 1 KiB use it).
//   demod_ring_*           the producer/consumer form: hand-offs are laid one after the
//                          other in a pinned ring (the producer freads straight into it,
//                          no copy), and whole runs of them travel as ONE upload + ONE
//                          hot-path call + ONE download -- as many as have arrived while
//                          the GPU was busy, at most 8 MiB, and at once when it is idle.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.h"

using namespace demod;

namespace {

constexpr uint64_t kStageBytes = 32ull << 20;  // input bytes a pipeline stage holds (also the largest hand-off)
constexpr int kDepth = 3;                      // pieces in flight
constexpr uint64_t kBatchBytes = 8ull << 20;   // ring: most input bytes one batch carries

// ---------------------------------------------------------------------------
// copy workers (pageable <-> pinned staging)
// ---------------------------------------------------------------------------
// One thread moves ~10 GB/s, a PCIe Gen5 x16 link five times that in each direction.
class CopyPool {
public:
    explicit CopyPool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    struct Job {
        void *dst;
        const void *src;
        size_t n;
    };
    // Runs all jobs (each cut into parts of about 2 MiB) on the workers and the calling thread; returns when done.
    void run(const std::vector<Job> &jobs) {
        std::vector<Job> parts;
        constexpr size_t kPart = 2u << 20;
        for (const Job &j : jobs)
            for (size_t off = 0; off < j.n; off += kPart)
                parts.push_back({static_cast<char *>(j.dst) + off, static_cast<const char *>(j.src) + off,
                                 std::min(kPart, j.n - off)});
        if (parts.empty()) return;
        if (parts.size() == 1 || workers_.empty()) {
            for (const Job &p : parts) memcpy(p.dst, p.src, p.n);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            parts_ = &parts;
            next_ = 0;
            left_ = parts.size();
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        parts_ = nullptr;
    }

private:
    void work() {
        for (;;) {
            Job p;
            {
                std::lock_guard<std::mutex> g(m_);
                if (!parts_ || next_ >= parts_->size()) return;
                p = (*parts_)[next_++];
            }
            memcpy(p.dst, p.src, p.n);
            std::lock_guard<std::mutex> g(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return stop_ || (parts_ && next_ < parts_->size()); });
                if (stop_) return;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::vector<Job> *parts_ = nullptr;
    size_t next_ = 0, left_ = 0;
    bool stop_ = false;
};

bool is_pinned_host(const void *p, uint64_t n) {
    if (!p || !n) return false;
    for (const char *q : {static_cast<const char *>(p), static_cast<const char *>(p) + n - 1}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void) hipGetLastError();  // not registered with the runtime: plain pageable memory
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------
// the pipeline's streams and stages
// ---------------------------------------------------------------------------
struct HostPipe {
    hipStream_t up = nullptr, run = nullptr, down = nullptr;
    struct Stage {
        void *h_in = nullptr, *h_out = nullptr, *h_halo = nullptr, *d_in = nullptr, *d_out = nullptr, *d_halo = nullptr;
        hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
        uint64_t out_count = 0;  // results of the piece in this stage (count, in the output format); 0 = stage free
        char *dst = nullptr;     // pageable destination the results are copied to when the stage retires (or null)
    } st[kDepth];
    bool stages_ready = false;
    CopyPool *pool = nullptr;
    // demod_stream_submit / _collect
    int submit_at = 0, collect_at = 0, in_flight = 0;
};

static int pipe_get(demod_ctx *ctx, bool want_stages, HostPipe **out) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->pipe) {
        HostPipe *p = new (std::nothrow) HostPipe;
        if (!p) {
            ctx->fail(DEMOD_E_NOMEM, "out of memory");
            return DEMOD_E_NOMEM;
        }
        ctx->pipe = p;
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->run, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking));
    }
    HostPipe *p = ctx->pipe;
    if (want_stages && !p->stages_ready) {
        for (auto &s : p->st) {
            HIP_TRY(ctx, hipHostMalloc(&s.h_in, kStageBytes, hipHostMallocDefault));
            HIP_TRY(ctx, hipHostMalloc(&s.h_out, kStageBytes, hipHostMallocDefault));
            HIP_TRY(ctx, hipHostMalloc(&s.h_halo, DEMOD_HALO_BYTES, hipHostMallocDefault));
            HIP_TRY(ctx, hipMalloc(&s.d_in, kStageBytes));
            HIP_TRY(ctx, hipMalloc(&s.d_out, kStageBytes));
            HIP_TRY(ctx, hipMalloc(&s.d_halo, DEMOD_HALO_BYTES));
            HIP_TRY(ctx, hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&s.computed, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming));
        }
        p->stages_ready = true;
    }
    *out = p;
    return DEMOD_OK;
}

void host_pipe_destroy(demod_ctx *ctx) {
    HostPipe *p = ctx->pipe;
    if (!p) return;
    for (hipStream_t s : {p->up, p->run, p->down})
        if (s) (void) hipStreamSynchronize(s);
    for (auto &s : p->st) {
        if (s.h_in) (void) hipHostFree(s.h_in);
        if (s.h_out) (void) hipHostFree(s.h_out);
        if (s.h_halo) (void) hipHostFree(s.h_halo);
        if (s.d_in) (void) hipFree(s.d_in);
        if (s.d_out) (void) hipFree(s.d_out);
        if (s.d_halo) (void) hipFree(s.d_halo);
        for (hipEvent_t e : {s.uploaded, s.computed, s.downloaded})
            if (e) (void) hipEventDestroy(e);
    }
    for (hipStream_t s : {p->up, p->run, p->down})
        if (s) (void) hipStreamDestroy(s);
    delete p->pool;
    delete p;
    ctx->pipe = nullptr;
}

bool host_pipe_busy(const demod_ctx *ctx) { return ctx->pipe && ctx->pipe->in_flight > 0; }

// upload (already enqueued on p->up) -> hot path -> download, chained by the stage's events
static int chain(demod_ctx *ctx, HostPipe *p, HostPipe::Stage &s, const void *d_in, uint64_t n_bytes, void *d_out,
                 const demod_span *sp, void *h_dst) {
    HIP_TRY(ctx, hipEventRecord(s.uploaded, p->up));
    HIP_TRY(ctx, hipStreamWaitEvent(p->run, s.uploaded, 0));
    const int rc = demod_process_device(ctx, d_in, n_bytes, d_out, sp, p->run);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(s.computed, p->run));
    HIP_TRY(ctx, hipStreamWaitEvent(p->down, s.computed, 0));
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_out, n_bytes / 4 * ctx->out_bytes(), hipMemcpyDeviceToHost, p->down));
    HIP_TRY(ctx, hipEventRecord(s.downloaded, p->down));
    return DEMOD_OK;
}

extern "C" {

// ---------------------------------------------------------------------------
// demod_process_host
// ---------------------------------------------------------------------------
static int process_host_pieces(demod_ctx *ctx, HostPipe *p, const void *h_in_v, uint64_t n_bytes, void *h_out_v, const demod_span *span) {
    char *h_out = static_cast<char *>(h_out_v);
    int rc = DEMOD_OK;

    const uint8_t *h_in = static_cast<const uint8_t *>(h_in_v);
    const uint64_t ob = ctx->out_bytes();
    const uint64_t B = ctx->seamless ? (1ull << 41) : ctx->plan.chunk_bytes;
    const uint64_t offset0 = span ? span->chunk_offset : 0;
    const bool eos = span ? span->end_of_stream != 0 : true;
    const uint8_t *tail_halo = span ? static_cast<const uint8_t *>(span->halo) : nullptr;
    // pinned caller memory goes over the link in place; pageable memory through the stages' pinned buffers
    const bool in_place = is_pinned_host(h_in, n_bytes), out_place = is_pinned_host(h_out, n_bytes / 4 * ob);
    if ((!in_place || !out_place) && !p->pool) {
        const unsigned hw = std::thread::hardware_concurrency();
        p->pool = new CopyPool(std::max(1u, std::min(7u, hw ? hw / 2 : 1u)));
    }

    // pieces are whole chunks when a chunk fits a stage, otherwise row-aligned cuts
    // inside a chunk with the look-ahead taken from the host buffer
    uint64_t piece = kStageBytes;
    if (B <= kStageBytes) piece = std::max<uint64_t>(B, std::min<uint64_t>(kStageBytes, 16ull << 20) / B * B);

    auto retire = [&](HostPipe::Stage &s, std::vector<CopyPool::Job> &jobs) -> int {
        if (!s.out_count) return DEMOD_OK;
        HIP_TRY(ctx, hipEventSynchronize(s.downloaded));
        if (s.dst) jobs.push_back({s.dst, s.h_out, (size_t) (s.out_count * ob)});
        s.out_count = 0;
        s.dst = nullptr;
        return DEMOD_OK;
    };

    uint64_t done = 0;
    int at = 0;
    while (done < n_bytes) {
        uint64_t len = std::min(piece, n_bytes - done);
        if (B <= kStageBytes && done == 0 && offset0) len = std::min(len, B - offset0);  // realign to chunks
        HostPipe::Stage &s = p->st[at];
        // the stage's previous piece leaves (its results are copied out, if they were staged)
        // while this piece is copied in: one batch of copy jobs
        std::vector<CopyPool::Job> jobs;
        if ((rc = retire(s, jobs))) return rc;

        demod_span sp{};
        sp.chunk_offset = B < kRowBytes ? 0 : (offset0 + done) % B;
        const bool last = done + len == n_bytes;
        sp.end_of_stream = last ? (eos ? 1u : 0u) : 0u;
        const bool ends_inside = B >= kRowBytes && ((offset0 + done + len) % B) != 0;
        if (ends_inside && !sp.end_of_stream) {
            if (last && !tail_halo) {
                ctx->fail(DEMOD_E_INVAL, "buffer ends inside a chunk of a continuing stream: span.halo is required");
                return DEMOD_E_INVAL;
            }
            // look-ahead bytes: what follows this piece in the host buffer, topped
            // up from the caller's halo when the buffer itself runs out
            uint8_t *hh = static_cast<uint8_t *>(s.h_halo);
            memset(hh, 0, DEMOD_HALO_BYTES);
            uint64_t have = 0;
            if (!last) {
                have = std::min<uint64_t>(DEMOD_HALO_BYTES, n_bytes - done - len);
                memcpy(hh, h_in + done + len, have);
            }
            if (have < DEMOD_HALO_BYTES && tail_halo) memcpy(hh + have, tail_halo, DEMOD_HALO_BYTES - have);
            HIP_TRY(ctx, hipMemcpyAsync(s.d_halo, hh, DEMOD_HALO_BYTES, hipMemcpyHostToDevice, p->up));
            sp.halo = s.d_halo;
        }
        const void *src = h_in + done;
        if (!in_place) {
            jobs.push_back({s.h_in, h_in + done, (size_t) len});
            src = s.h_in;
        }
        if (!jobs.empty()) p->pool->run(jobs);
        HIP_TRY(ctx, hipMemcpyAsync(s.d_in, src, len, hipMemcpyHostToDevice, p->up));
        char *dst = h_out + done / 4 * ob;
        if ((rc = chain(ctx, p, s, s.d_in, len, s.d_out, &sp, out_place ? (void *) dst : s.h_out))) return rc;
        s.out_count = len / 4;
        s.dst = out_place ? nullptr : dst;
        done += len;
        at = (at + 1) % kDepth;
    }
    for (int k = 0; k < kDepth; ++k) {  // oldest first
        std::vector<CopyPool::Job> jobs;
        if ((rc = retire(p->st[(at + k) % kDepth], jobs))) return rc;
        if (!jobs.empty()) p->pool->run(jobs);
    }
    return DEMOD_OK;
}


int demod_process_host(demod_ctx *ctx, const void *h_in_v, uint64_t n_bytes, void *h_out_v, const demod_span *span) {
    if (!ctx) return DEMOD_E_INVAL;
    if (n_bytes == 0) return DEMOD_OK;
    if (!h_in_v || !h_out_v || (n_bytes & 3)) {
        ctx->fail(DEMOD_E_INVAL, "null buffer or n_bytes not a multiple of 4");
        return DEMOD_E_INVAL;
    }
    HostPipe *p = nullptr;
    int rc = pipe_get(ctx, true, &p);
    if (rc) return rc;
    if (p->in_flight) {
        ctx->fail(DEMOD_E_INVAL, "demod_process_host while submitted pieces are uncollected");
        return DEMOD_E_INVAL;
    }
    rc = process_host_pieces(ctx, p, h_in_v, n_bytes, h_out_v, span);
    if (rc) {
        // A failure part-way leaves pieces on the three streams and stages that still name the caller's
        // output buffer: drain the streams and forget the pieces, so that the next call neither retires
        // stale results nor copies them to a buffer that may be gone.  (The error message stays.)
        const std::string msg = ctx->err;
        const int code = ctx->err_code;
        for (hipStream_t s : {p->up, p->run, p->down})
            if (s) (void) hipStreamSynchronize(s);
        for (int k = 0; k < kDepth; ++k) {
            p->st[k].out_count = 0;
            p->st[k].dst = nullptr;
        }
        ctx->err = msg;
        ctx->err_code = code;
    }
    return rc;
}

// ---------------------------------------------------------------------------
// demod_stream_submit / _collect
// ---------------------------------------------------------------------------
uint64_t demod_stream_capacity(void) { return kStageBytes; }
int demod_stream_depth(void) { return kDepth; }

int demod_stream_submit(demod_ctx *ctx, const void *h_in, uint64_t n_bytes, const demod_span *span) {
    if (!ctx) return DEMOD_E_INVAL;
    if (!h_in || n_bytes == 0 || (n_bytes & 3) || n_bytes > kStageBytes) {
        ctx->fail(DEMOD_E_INVAL, "stream_submit: need 0 < n_bytes <= capacity, multiple of 4");
        return DEMOD_E_INVAL;
    }
    HostPipe *p = nullptr;
    int rc = pipe_get(ctx, true, &p);
    if (rc) return rc;
    if (p->in_flight >= kDepth) {
        ctx->fail(DEMOD_E_INVAL, "stream_submit: every staging stage is in flight; collect first");
        return DEMOD_E_INVAL;
    }
    HostPipe::Stage &s = p->st[p->submit_at];
    demod_span sp{};
    if (span) sp = *span;
    else sp.end_of_stream = 1;
    if (sp.halo) {  // host pointer here: stage it
        memcpy(s.h_halo, sp.halo, DEMOD_HALO_BYTES);
        HIP_TRY(ctx, hipMemcpyAsync(s.d_halo, s.h_halo, DEMOD_HALO_BYTES, hipMemcpyHostToDevice, p->up));
        sp.halo = s.d_halo;
    }
    memcpy(s.h_in, h_in, n_bytes);  // the caller's buffer is free again when this returns
    HIP_TRY(ctx, hipMemcpyAsync(s.d_in, s.h_in, n_bytes, hipMemcpyHostToDevice, p->up));
    if ((rc = chain(ctx, p, s, s.d_in, n_bytes, s.d_out, &sp, s.h_out))) return rc;
    s.out_count = n_bytes / 4;
    p->submit_at = (p->submit_at + 1) % kDepth;
    p->in_flight++;
    return DEMOD_OK;
}

int demod_stream_collect(demod_ctx *ctx, const void **out, uint64_t *n_outputs) {
    if (!ctx || !out || !n_outputs) return DEMOD_E_INVAL;
    *out = nullptr;
    *n_outputs = 0;
    HostPipe *p = ctx->pipe;
    if (!p || p->in_flight == 0) return DEMOD_OK;  // nothing pending
    HostPipe::Stage &s = p->st[p->collect_at];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(s.downloaded));
    *out = s.h_out;
    *n_outputs = s.out_count;
    s.out_count = 0;
    p->collect_at = (p->collect_at + 1) % kDepth;
    p->in_flight--;
    return DEMOD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// demod_ring_*: hand-offs laid one after the other in a pinned ring, batched
// ---------------------------------------------------------------------------
struct demod_ring {
    demod_ctx *ctx = nullptr;
    HostPipe *pipe = nullptr;
    uint64_t slot = 0;    // bytes per slot = the chunk size B
    uint32_t n_slots = 0;
    uint32_t batch_max = 1;  // most slots one batch carries
    char *h_in = nullptr;    // pinned input arena (the caller's, or owned)
    bool owns_in = false;
    char *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;  // same slot layout: slot j at j * slot (outputs: j * slot / 4 * out_bytes)

    std::mutex m;
    std::condition_variable cv;
    // slots [tail, head) hold committed hand-offs (ring indices, monotonically increasing);
    // of those, [tail, sent) have been submitted; the producer writes into slot head
    uint64_t head = 0, sent = 0, tail = 0;
    uint64_t last_bytes = 0;  // bytes of the most recently committed slot (only a final one may be short)
    bool finished = false;    // no hand-off will follow
    bool failed = false;
    struct Batch {
        uint64_t first;       // ring index of its first slot
        uint32_t count;
        uint64_t bytes;
        hipEvent_t uploaded, computed, downloaded;
    };
    std::deque<Batch> flying;                                // submitted, oldest first
    std::vector<Batch> spare;                                // event triples for reuse
    bool lent = false;                                       // the front batch's results are with the caller (collect .. release)
    bool counted = false;                                    // in ctx->rings
    uint64_t n_batches = 0, n_handoffs = 0, biggest = 0;      // DEMOD_RING_STATS=1: printed when the ring goes
};

namespace {

// Submit committed slots [sent, sent + count) as one batch.  Called with r->m held.
int ring_submit(demod_ring *r, uint32_t count, uint64_t bytes) {
    demod_ctx *ctx = r->ctx;
    HostPipe *p = r->pipe;
    demod_ring::Batch b{};
    if (!r->spare.empty()) {
        b = r->spare.back();
        r->spare.pop_back();
    }
    // Whatever fails from here on, the batch's events go back to `spare`: demod_ring_destroy
    // destroys what is there (and in `flying`), nothing is left behind.  (A triple may be
    // partly created: destroy skips null events.)
    auto enqueue = [&]() -> int {
        if (!b.uploaded) HIP_TRY(ctx, hipEventCreateWithFlags(&b.uploaded, hipEventDisableTiming));
        if (!b.computed) HIP_TRY(ctx, hipEventCreateWithFlags(&b.computed, hipEventDisableTiming));
        if (!b.downloaded) HIP_TRY(ctx, hipEventCreateWithFlags(&b.downloaded, hipEventDisableTiming));
        b.first = r->sent;
        b.count = count;
        b.bytes = bytes;
        const uint64_t off = (b.first % r->n_slots) * r->slot;
        const uint64_t ooff = off / 4 * ctx->out_bytes();
        HIP_TRY(ctx, hipMemcpyAsync(r->d_in + off, r->h_in + off, bytes, hipMemcpyHostToDevice, p->up));
        HIP_TRY(ctx, hipEventRecord(b.uploaded, p->up));
        HIP_TRY(ctx, hipStreamWaitEvent(p->run, b.uploaded, 0));
        // whole chunks, a short one only at the very end: every batch is a stream of its own
        demod_span sp{};
        sp.end_of_stream = 1;
        const int rc = demod_process_device(ctx, r->d_in + off, bytes, r->d_out + ooff, &sp, p->run);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(b.computed, p->run));
        HIP_TRY(ctx, hipStreamWaitEvent(p->down, b.computed, 0));
        HIP_TRY(ctx, hipMemcpyAsync(r->h_out + ooff, r->d_out + ooff, bytes / 4 * ctx->out_bytes(), hipMemcpyDeviceToHost, p->down));
        HIP_TRY(ctx, hipEventRecord(b.downloaded, p->down));
        return DEMOD_OK;
    };
    const int rc = enqueue();
    if (rc) {
        r->spare.push_back(b);
        return rc;
    }
    r->flying.push_back(b);
    r->sent += count;
    r->n_batches++;
    r->n_handoffs += count;
    r->biggest = std::max<uint64_t>(r->biggest, count);
    return DEMOD_OK;
}

// Submit what has been committed, in batches that do not wrap around the ring's end.
// force: everything; otherwise only full batches, plus whatever there is while the GPU has
// nothing to do (a slow producer's hand-offs leave one at a time, a fast one's in runs).
int ring_pump(demod_ring *r, bool force) {
    while (r->sent < r->head) {
        const uint64_t pending = r->head - r->sent;
        const uint64_t to_end = r->n_slots - (r->sent % r->n_slots);
        const uint32_t count = (uint32_t) std::min<uint64_t>({pending, to_end, r->batch_max});
        const bool full = count == r->batch_max || count == to_end;
        if (!(force || full || r->flying.empty())) break;
        const bool has_last = r->sent + count == r->head;
        const uint64_t bytes = (uint64_t) (count - 1) * r->slot + (has_last ? r->last_bytes : r->slot);
        const int rc = ring_submit(r, count, bytes);
        if (rc) {
            r->failed = true;
            return rc;
        }
    }
    return DEMOD_OK;
}

}  // namespace

extern "C" {

uint32_t demod_ring_slots_for(uint64_t chunk_bytes) {
    if (chunk_bytes < kRowBytes || chunk_bytes > kStageBytes || (chunk_bytes & (chunk_bytes - 1))) return 0;
    const uint64_t per_batch = std::max<uint64_t>(1, kBatchBytes / chunk_bytes);
    return (uint32_t) (kDepth * per_batch + 1);
}

int demod_ring_create(demod_ctx *ctx, void *pinned_in, uint32_t n_slots, demod_ring **out) {
    if (!ctx || !out) return DEMOD_E_INVAL;
    *out = nullptr;
    const uint64_t B = ctx->plan.chunk_bytes;
    const uint32_t want = demod_ring_slots_for(B);
    if (!want || ctx->seamless) {
        ctx->fail(DEMOD_E_INVAL, "a ring needs chunks of 1 KiB .. 32 MiB and the reference's per-chunk filtering (not seamless mode)");
        return DEMOD_E_INVAL;
    }
    if (pinned_in && n_slots < 3) {
        ctx->fail(DEMOD_E_INVAL, "a caller-provided ring arena needs at least 3 slots of one chunk each");
        return DEMOD_E_INVAL;
    }
    HostPipe *p = nullptr;
    int rc = pipe_get(ctx, false, &p);
    if (rc) return rc;
    demod_ring *r = new (std::nothrow) demod_ring;
    if (!r) {
        ctx->fail(DEMOD_E_NOMEM, "out of memory");
        return DEMOD_E_NOMEM;
    }
    r->ctx = ctx;
    r->pipe = p;
    r->slot = B;
    r->n_slots = pinned_in ? n_slots : want;
    r->batch_max = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(kBatchBytes / B, (r->n_slots - 1) / 2));
    const uint64_t arena = (uint64_t) r->n_slots * B;
    hipError_t e = hipSuccess;
    if (pinned_in) {
        r->h_in = static_cast<char *>(pinned_in);
    } else {
        e = hipHostMalloc((void **) &r->h_in, arena, hipHostMallocDefault);
        r->owns_in = e == hipSuccess;
    }
    if (e == hipSuccess) e = hipHostMalloc((void **) &r->h_out, arena, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **) &r->d_in, arena);
    if (e == hipSuccess) e = hipMalloc((void **) &r->d_out, arena);
    if (e != hipSuccess) {
        ctx->fail(DEMOD_E_HIP, "ring buffers", e);
        demod_ring_destroy(r);
        return DEMOD_E_HIP;
    }
    ctx->rings++;
    r->counted = true;
    *out = r;
    return DEMOD_OK;
}

void demod_ring_destroy(demod_ring *r) {
    if (!r) return;
    if (const char *e = getenv("DEMOD_RING_STATS"); e && e[0] == '1' && r->n_batches)
        fprintf(stderr, "demod ring: %llu hand-offs of %llu bytes in %llu batches (%.1f per batch, largest %llu), %u slots\n",
                (unsigned long long) r->n_handoffs, (unsigned long long) r->slot, (unsigned long long) r->n_batches,
                (double) r->n_handoffs / (double) r->n_batches, (unsigned long long) r->biggest, r->n_slots);
    (void) hipSetDevice(r->ctx->device);
    for (hipStream_t s : {r->pipe->up, r->pipe->run, r->pipe->down})
        if (s) (void) hipStreamSynchronize(s);
    for (auto &b : r->flying) r->spare.push_back(b);
    for (auto &b : r->spare)
        for (hipEvent_t e : {b.uploaded, b.computed, b.downloaded})
            if (e) (void) hipEventDestroy(e);
    if (r->owns_in && r->h_in) (void) hipHostFree(r->h_in);
    if (r->h_out) (void) hipHostFree(r->h_out);
    if (r->d_in) (void) hipFree(r->d_in);
    if (r->d_out) (void) hipFree(r->d_out);
    if (r->counted) r->ctx->rings--;
    delete r;
}

void *demod_ring_input(demod_ring *r) {
    if (!r) return nullptr;
    std::unique_lock<std::mutex> g(r->m);
    // the slot after the committed ones must not be one whose results are still to be written out
    r->cv.wait(g, [r] { return r->failed || r->head - r->tail < r->n_slots; });
    return r->h_in + (r->head % r->n_slots) * r->slot;
}

int demod_ring_commit(demod_ring *r, uint64_t n_bytes, int final) {
    if (!r) return DEMOD_E_INVAL;
    std::unique_lock<std::mutex> g(r->m);
    demod_ctx *ctx = r->ctx;
    if (r->failed) return DEMOD_E_HIP;
    if (r->finished || n_bytes > r->slot || (n_bytes & 3) || (n_bytes < r->slot && !final)) {
        ctx->fail(DEMOD_E_INVAL, "ring_commit: a hand-off is one whole chunk; only the last one may be shorter (multiple of 4)");
        return DEMOD_E_INVAL;
    }
    if (r->head - r->tail >= r->n_slots) {
        ctx->fail(DEMOD_E_INVAL, "ring_commit without demod_ring_input: the ring is full");
        return DEMOD_E_INVAL;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return DEMOD_E_HIP;
    if (n_bytes) {
        r->head++;
        r->last_bytes = n_bytes;
    }
    if (final) r->finished = true;
    const int rc = ring_pump(r, final != 0);
    r->cv.notify_all();
    return rc;
}

int demod_ring_collect(demod_ring *r, const void **out, uint64_t *n_outputs) {
    if (!r || !out || !n_outputs) return DEMOD_E_INVAL;
    *out = nullptr;
    *n_outputs = 0;
    demod_ring::Batch b{};
    {
        std::unique_lock<std::mutex> g(r->m);
        if (r->lent) {
            r->ctx->fail(DEMOD_E_INVAL, "ring_collect: release the previous batch first");
            return DEMOD_E_INVAL;
        }
        r->cv.wait(g, [r] { return r->failed || !r->flying.empty() || (r->finished && r->sent == r->head); });
        if (r->failed) return DEMOD_E_HIP;
        if (r->flying.empty()) return DEMOD_OK;  // the stream has ended and everything was handed out
        b = r->flying.front();
        r->lent = true;
    }
    if (hipSetDevice(r->ctx->device) != hipSuccess || hipEventSynchronize(b.downloaded) != hipSuccess) {
        std::lock_guard<std::mutex> g(r->m);
        r->failed = true;
        r->ctx->fail(DEMOD_E_HIP, "ring_collect: waiting for a batch", hipGetLastError());
        r->cv.notify_all();
        return DEMOD_E_HIP;
    }
    *out = r->h_out + (b.first % r->n_slots) * r->slot / 4 * r->ctx->out_bytes();
    *n_outputs = b.bytes / 4;
    return DEMOD_OK;
}

int demod_ring_release(demod_ring *r) {
    if (!r) return DEMOD_E_INVAL;
    std::unique_lock<std::mutex> g(r->m);
    if (!r->lent) return DEMOD_OK;
    demod_ring::Batch b = r->flying.front();
    r->flying.pop_front();
    r->spare.push_back(b);
    r->tail = b.first + b.count;
    r->lent = false;
    // the GPU may have run dry while hand-offs trickled in: send what has arrived
    int rc = DEMOD_OK;
    if (r->flying.empty() && !r->failed && hipSetDevice(r->ctx->device) == hipSuccess) rc = ring_pump(r, r->finished);
    r->cv.notify_all();
    return rc;
}

}  // extern "C"
