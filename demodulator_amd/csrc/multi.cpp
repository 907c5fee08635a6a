// multi.cpp -- the hot path over several GPUs from ONE process (include/demod_hip.h,
// demod_multi_*): SURVEY.md section 8(b) "multi-GPU init" and 8(e).
//
// The reference has no parallelism beyond one producer and one consumer thread
// (/root/reference/src/main.c:67-70); its chunks, however, are filtered independently
// (src/matrix.c:272), so a stream splits into contiguous shards with NO exchange when the
// cuts are chunk boundaries, and independent channels (BASELINE config 5) never interact.
// When a cut falls inside a chunk the shard on its left needs the DEMOD_HALO_BYTES that
// follow it: one ncclSend / ncclRecv pair per cut, all of a step's pairs in one group
// (RCCL over xGMI), enqueued on each device's stream ahead of its kernels.  librccl is
// loaded on first need (dlopen), so that the CLI and single-GPU users do not pay for it; if
// it is not there, or refuses the device list, the 64 bytes travel with
// hipMemcpyPeerAsync instead and demod_multi_halo_transport() says so.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ctx.h"

using namespace demod;

namespace {

// the few RCCL entry points used, resolved at run time (prototypes: rccl/rccl.h:236-933)
struct Rccl {
    typedef void *comm_t;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *handle = nullptr;
    static constexpr int kUint8 = 1;  // ncclUint8, rccl.h:460
    bool load() {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) return false;
        auto sym = [this](const char *n) { return dlsym(handle, n); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return CommInitAll && CommDestroy && Send && Recv && GroupStart && GroupEnd;
    }
};

}  // namespace

struct demod_multi {
    std::vector<int> devices;
    std::vector<demod_ctx *> ctx;
    std::vector<hipStream_t> stream;
    std::vector<uint8_t *> d_halo;
    uint64_t chunk = 0;
    Rccl rccl;
    std::vector<Rccl::comm_t> comm;
    enum { kUnset, kNone, kRccl, kPeer } transport = kUnset;
    std::string err;
    int fail(int code, const std::string &what) {
        err = what;
        return code;
    }
};

extern "C" {

// ---------------------------------------------------------------------------
// shard plan (host only)
// ---------------------------------------------------------------------------
int demod_plan_shards(uint64_t total_bytes, int n, uint64_t chunk_bytes, uint64_t phase_bytes, uint64_t align,
                      demod_shard *out) {
    if (n < 1 || !out || chunk_bytes < 4 || (chunk_bytes & (chunk_bytes - 1))) return DEMOD_E_INVAL;
    if ((phase_bytes % kRowBytes) || phase_bytes >= chunk_bytes) return DEMOD_E_INVAL;
    const uint64_t share = total_bytes / (uint64_t) n;
    if (align == 0)
        align = (chunk_bytes >= kRowBytes && phase_bytes == 0 && share >= chunk_bytes) ? chunk_bytes : kRowBytes;
    if (chunk_bytes < kRowBytes) align = chunk_bytes;  // tiny chunks: whole chunks only
    if (chunk_bytes >= kRowBytes && (align % kRowBytes)) return DEMOD_E_INVAL;
    std::vector<uint64_t> cuts(n + 1, 0);
    for (int r = 1; r < n; ++r) cuts[r] = std::min(total_bytes, share * (uint64_t) r / align * align);
    cuts[n] = total_bytes;
    // the look-ahead of a shard comes from the next NON-EMPTY shard; an empty shard takes no part
    std::vector<int> next(n, -1);
    int later = -1;
    for (int r = n - 1; r >= 0; --r) {
        next[r] = later;
        if (cuts[r + 1] > cuts[r]) later = r;
    }
    for (int r = 0; r < n; ++r) {
        demod_shard &s = out[r];
        memset(&s, 0, sizeof s);
        s.start = cuts[r];
        s.length = cuts[r + 1] - cuts[r];
        s.chunk_offset = (phase_bytes + s.start) % chunk_bytes;
        s.end_of_stream = cuts[r + 1] == total_bytes;
        s.halo_src = s.halo_dst = -1;
    }
    for (int r = 0; r < n; ++r) {
        demod_shard &s = out[r];
        const uint64_t stop = s.start + s.length;
        if (s.length && chunk_bytes >= kRowBytes && (phase_bytes + stop) % chunk_bytes != 0 && stop < total_bytes &&
            next[r] >= 0) {
            s.needs_halo = 1;
            s.halo_src = next[r];
            out[next[r]].halo_dst = r;
        }
    }
    return DEMOD_OK;
}

int demod_plan_channels(uint64_t n_channels, uint64_t bytes_per_channel, int n, uint64_t chunk_bytes, demod_shard *out) {
    if (n < 1 || !out || chunk_bytes < 4 || (bytes_per_channel % chunk_bytes)) return DEMOD_E_INVAL;
    // whole channels per device, as even as they go; every channel is a stream of its own made
    // of whole chunks, so a device's channels laid back to back are one buffer for one call
    uint64_t c0 = 0;
    for (int r = 0; r < n; ++r) {
        const uint64_t cnt = n_channels / (uint64_t) n + ((uint64_t) r < n_channels % (uint64_t) n ? 1 : 0);
        demod_shard &s = out[r];
        memset(&s, 0, sizeof s);
        s.start = c0 * bytes_per_channel;
        s.length = cnt * bytes_per_channel;
        s.end_of_stream = 1;
        s.halo_src = s.halo_dst = -1;
        c0 += cnt;
    }
    return DEMOD_OK;
}

// ---------------------------------------------------------------------------
// contexts, streams
// ---------------------------------------------------------------------------
int demod_multi_create(const demod_params *p, const int *devices, int n_devices, demod_multi **out) {
    if (!p || !devices || n_devices < 1 || !out) return DEMOD_E_INVAL;
    *out = nullptr;
    demod_multi *m = new (std::nothrow) demod_multi;
    if (!m) return DEMOD_E_NOMEM;
    m->chunk = p->buf_size;
    for (int i = 0; i < n_devices; ++i) {
        demod_ctx *c = nullptr;
        const int rc = demod_create(p, devices[i], &c);  // message in demod_last_error(NULL)
        if (rc != DEMOD_OK) {
            demod_multi_destroy(m);
            return rc;
        }
        m->devices.push_back(devices[i]);
        m->ctx.push_back(c);
        if (c->plan.conditioning == kCorrectIq) {
            // -q1: correctIq's running offsets belong to ONE stream walked in order (src/matrix.c:125 static state).
            // Every shard but the first would start them from zero, and a device's channels, laid back to back in
            // one call, would hand them from one channel to the next: neither is what the reference computes.
            demod_multi_destroy(m);
            return demod_create_fail(DEMOD_E_INVAL, "-q1 (correctIq) carries its offsets through one stream in order: it cannot be "
                                                    "sharded or batched over devices (demod_multi_*); use one context per stream");
        }
        hipStream_t s = nullptr;
        uint8_t *h = nullptr;
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void **) &h, DEMOD_HALO_BYTES) != hipSuccess) {
            if (s) (void) hipStreamDestroy(s);
            demod_multi_destroy(m);
            return DEMOD_E_HIP;
        }
        m->stream.push_back(s);
        m->d_halo.push_back(h);
    }
    *out = m;
    return DEMOD_OK;
}

void demod_multi_destroy(demod_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void) hipSetDevice(m->devices[i]);
        if (i < m->stream.size() && m->stream[i]) {
            (void) hipStreamSynchronize(m->stream[i]);
        }
    }
    for (auto c : m->comm)
        if (c && m->rccl.CommDestroy) (void) m->rccl.CommDestroy(c);
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        (void) hipSetDevice(m->devices[i]);
        if (i < m->stream.size() && m->stream[i]) (void) hipStreamDestroy(m->stream[i]);
        if (i < m->d_halo.size() && m->d_halo[i]) (void) hipFree(m->d_halo[i]);
        demod_destroy(m->ctx[i]);
    }
    delete m;
}

int demod_multi_devices(const demod_multi *m) { return m ? (int) m->ctx.size() : DEMOD_E_INVAL; }
demod_ctx *demod_multi_context(demod_multi *m, int i) {
    return (m && i >= 0 && i < (int) m->ctx.size()) ? m->ctx[i] : nullptr;
}
void *demod_multi_stream(demod_multi *m, int i) {
    return (m && i >= 0 && i < (int) m->stream.size()) ? (void *) m->stream[i] : nullptr;
}
const char *demod_multi_last_error(const demod_multi *m) { return m ? m->err.c_str() : ""; }

const char *demod_multi_halo_transport(const demod_multi *m) {
    if (!m) return "";
    switch (m->transport) {
    case demod_multi::kRccl: return "rccl";
    case demod_multi::kPeer: return "peer-copy";
    case demod_multi::kNone: return "none";
    default: return "unset";
    }
}

int demod_multi_plan(const demod_multi *m, uint64_t total_bytes, uint64_t phase_bytes, demod_shard *out) {
    if (!m) return DEMOD_E_INVAL;
    return demod_plan_shards(total_bytes, (int) m->ctx.size(), m->chunk, phase_bytes, 0, out);
}

// the first exchange decides how halos travel: RCCL if it loads and takes the device list
// (DEMOD_MULTI_TRANSPORT=peer skips it), else peer copies
static void choose_transport(demod_multi *m) {
    const char *want = getenv("DEMOD_MULTI_TRANSPORT");
    m->transport = demod_multi::kPeer;
    if (want && !strcmp(want, "peer")) return;
    if (!m->rccl.load()) return;
    m->comm.assign(m->ctx.size(), nullptr);
    const int rc = m->rccl.CommInitAll(m->comm.data(), (int) m->devices.size(), m->devices.data());
    if (rc != 0) {
        m->err = std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "failed") +
                 " -- halos travel by peer copy";
        m->comm.clear();
        return;
    }
    m->transport = demod_multi::kRccl;
}

int demod_multi_process_device(demod_multi *m, void *const *d_in, void *const *d_out, const demod_shard *sh) {
    if (!m || !d_in || !d_out || !sh) return DEMOD_E_INVAL;
    const int n = (int) m->ctx.size();
    bool any_halo = false;
    for (int i = 0; i < n; ++i) {
        if (sh[i].needs_halo) {
            const int src = sh[i].halo_src;
            if (src < 0 || src >= n || sh[src].halo_dst != i || !sh[src].length)
                return m->fail(DEMOD_E_INVAL, "shard plan: a halo without its source");
            any_halo = true;
        }
    }
    // "none" only says that no step has needed an exchange so far
    if (any_halo && (m->transport == demod_multi::kUnset || m->transport == demod_multi::kNone)) choose_transport(m);
    if (!any_halo && m->transport == demod_multi::kUnset) m->transport = demod_multi::kNone;

    // ---- look-ahead bytes of every cut that falls inside a chunk --------------------------
    if (any_halo && m->transport == demod_multi::kRccl) {
        int rc = m->rccl.GroupStart();
        for (int i = 0; i < n && rc == 0; ++i) {
            if (sh[i].halo_dst >= 0)
                rc = m->rccl.Send(d_in[i], std::min<uint64_t>(DEMOD_HALO_BYTES, sh[i].length), Rccl::kUint8, sh[i].halo_dst,
                                  m->comm[i], m->stream[i]);
            if (rc == 0 && sh[i].needs_halo)
                rc = m->rccl.Recv(m->d_halo[i], std::min<uint64_t>(DEMOD_HALO_BYTES, sh[sh[i].halo_src].length), Rccl::kUint8,
                                  sh[i].halo_src, m->comm[i], m->stream[i]);
        }
        const int rc2 = m->rccl.GroupEnd();
        if (rc || rc2)
            return m->fail(DEMOD_E_HIP, std::string("RCCL halo exchange: ") +
                                            (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc ? rc : rc2) : "failed"));
    }
    for (int i = 0; i < n; ++i) {
        if (!sh[i].length) continue;
        if (hipSetDevice(m->devices[i]) != hipSuccess) return m->fail(DEMOD_E_HIP, "hipSetDevice");
        demod_span sp{};
        sp.chunk_offset = sh[i].chunk_offset;
        sp.end_of_stream = sh[i].end_of_stream ? 1u : 0u;
        if (sh[i].needs_halo) {
            const int src = sh[i].halo_src;
            const uint64_t have = std::min<uint64_t>(DEMOD_HALO_BYTES, sh[src].length);
            // a source shorter than the halo can only be the stream's last bytes: zeros follow
            if (have < DEMOD_HALO_BYTES &&
                hipMemsetAsync(m->d_halo[i] + have, 0, DEMOD_HALO_BYTES - have, m->stream[i]) != hipSuccess)
                return m->fail(DEMOD_E_HIP, "hipMemsetAsync");
            if (m->transport == demod_multi::kPeer &&
                hipMemcpyPeerAsync(m->d_halo[i], m->devices[i], d_in[src], m->devices[src], have, m->stream[i]) != hipSuccess)
                return m->fail(DEMOD_E_HIP, "hipMemcpyPeerAsync");
            sp.halo = m->d_halo[i];
        }
        const int rc = demod_process_device(m->ctx[i], d_in[i], sh[i].length, d_out[i], &sp, m->stream[i]);
        if (rc) return m->fail(rc, std::string("device ") + std::to_string(m->devices[i]) + ": " + demod_last_error(m->ctx[i]));
    }
    return DEMOD_OK;
}

int demod_multi_synchronize(demod_multi *m) {
    if (!m) return DEMOD_E_INVAL;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (hipSetDevice(m->devices[i]) != hipSuccess || hipStreamSynchronize(m->stream[i]) != hipSuccess)
            return m->fail(DEMOD_E_HIP, "hipStreamSynchronize");
    }
    return DEMOD_OK;
}

}  // extern "C"
