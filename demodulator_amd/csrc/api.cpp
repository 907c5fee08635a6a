// api.cpp -- the C ABI declared in include/demod_hip.h: contexts, the device
// hot-path entry point, helpers.  (The host-buffer paths: host.cpp.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ctx.h"

#ifndef DEMOD_Q1_THREADS
#define DEMOD_Q1_THREADS 16384   // threads of correct_iq_offsets that fill the chip for its purposes (see run_offsets)
#endif


using namespace demod;

namespace {

std::mutex g_err_mutex;
std::string g_create_error;

#ifndef DEMOD_TINY_THREADS
#define DEMOD_TINY_THREADS 65536
#endif
constexpr uint32_t kTinyThreads = DEMOD_TINY_THREADS;  // threads of demod_tiny (a multiple of 64); each owns tiny_stride floats of scratch
constexpr uint32_t kPieceRows = 1u << 21;       // rows per demod_rows launch (2 GiB of input)
// chunk heads inside the streaming kernel for chunks of at least this many rows (below, the
// demod_heads kernel is cheaper), see heads_are_fused()
constexpr uint64_t kFusedMinRows = 256;

uint32_t log2u(uint64_t v) {
    uint32_t l = 0;
    while ((1ull << (l + 1)) <= v) ++l;
    return l;
}

}  // namespace

extern "C" {

void demod_default_params(demod_params *p) {
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->sample_rate = 125000.;
    p->lowpass_in = 0.;
    p->lowpass_out = 12500.;
    p->epsilon = .3;
    p->in_degree = 0;
    p->out_degree = 3;
    p->mode = 0x10;
    p->buf_size = DEMOD_DEFAULT_BUF_SIZE;
}

int demod_abi_version(void) { return DEMOD_ABI_VERSION; }

const char *demod_last_error(const demod_ctx *ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> g(g_err_mutex);
    static thread_local std::string copy;
    copy = g_create_error;
    return copy.c_str();
}

int demod_last_error_code(const demod_ctx *ctx) { return ctx ? ctx->err_code : DEMOD_E_INVAL; }

static int create_fail(int code, const std::string &msg) {
    std::lock_guard<std::mutex> g(g_err_mutex);
    g_create_error = msg;
    return code;
}

int demod_create_fail(int code, const char *msg) { return create_fail(code, msg ? msg : ""); }

int demod_create(const demod_params *p, int device, demod_ctx **out) {
    if (!p || !out) return create_fail(DEMOD_E_INVAL, "null argument");
    *out = nullptr;
    demod_ctx *ctx = new (std::nothrow) demod_ctx;
    if (!ctx) return create_fail(DEMOD_E_NOMEM, "out of memory");
    ctx->piece_rows = kPieceRows;
    ctx->opt = options_from_params(*p);
    std::string err;
    if (!build_plan(ctx->opt, ctx->plan, err)) {
        delete ctx;
        return create_fail(DEMOD_E_INVAL, err);
    }
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        delete ctx;
        return create_fail(DEMOD_E_NODEVICE,
                           std::string("no HIP device: the demodulation path has no CPU fallback (") +
                               hipGetErrorString(e) + ")");
    }
    if (device < 0 || device >= n_dev) {
        delete ctx;
        return create_fail(DEMOD_E_INVAL, "device index out of range");
    }
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        std::string msg = std::string("HIP setup failed: ") + hipGetErrorString(e);
        demod_destroy(ctx);
        return create_fail(DEMOD_E_HIP, msg);
    }
    *out = ctx;
    return DEMOD_OK;
}

void demod_destroy(demod_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    host_pipe_destroy(ctx);
    for (auto &p : ctx->timed) {
        (void) hipEventDestroy(p.first);
        (void) hipEventDestroy(p.second);
    }
    for (auto &p : ctx->event_pool) {
        (void) hipEventDestroy(p.first);
        (void) hipEventDestroy(p.second);
    }
    if (ctx->q1_table) (void) hipFree(ctx->q1_table);
    if (ctx->q1_carry) (void) hipFree(ctx->q1_carry);
    if (ctx->tiny_work) (void) hipFree(ctx->tiny_work);
    if (ctx->tiny_done) (void) hipEventDestroy(ctx->tiny_done);
    delete ctx;
}

int demod_set_segment_rows(demod_ctx *ctx, uint32_t rows) {
    if (!ctx) return DEMOD_E_INVAL;
    if (rows & (rows - 1)) {
        ctx->fail(DEMOD_E_INVAL, "segment rows must be a power of two (or 0)");
        return DEMOD_E_INVAL;
    }
    ctx->seg_rows = rows;
    return DEMOD_OK;
}

int demod_set_output_format(demod_ctx *ctx, int format, float scale) {
    if (!ctx) return DEMOD_E_INVAL;
    if ((format != DEMOD_OUT_F32 && format != DEMOD_OUT_S16) || !(scale > 0.f) || !(scale < 3.0e38f)) {
        ctx->fail(DEMOD_E_INVAL, "output format must be DEMOD_OUT_F32 or DEMOD_OUT_S16 with a positive finite scale");
        return DEMOD_E_INVAL;
    }
    if (host_pipe_busy(ctx) || ctx->rings) {
        ctx->fail(DEMOD_E_INVAL, "output format cannot change while submitted pieces are uncollected or a ring is open");
        return DEMOD_E_INVAL;
    }
    ctx->out_s16 = format == DEMOD_OUT_S16;
    ctx->out_scale = scale;
    return DEMOD_OK;
}

int demod_output_bytes(const demod_ctx *ctx) { return ctx ? (int) ctx->out_bytes() : DEMOD_E_INVAL; }

int demod_set_seamless(demod_ctx *ctx, int on) {
    if (!ctx) return DEMOD_E_INVAL;
    if ((on != 0) != ctx->seamless && (host_pipe_busy(ctx) || ctx->rings)) {
        ctx->fail(DEMOD_E_INVAL, "seamless mode cannot change while submitted pieces are uncollected or a ring is open");
        return DEMOD_E_INVAL;
    }
    ctx->seamless = on != 0;
    return DEMOD_OK;
}

static int rows_conditioning(const demod_ctx *ctx);

int demod_set_stencil_form(demod_ctx *ctx, int form) {
    if (!ctx) return DEMOD_E_INVAL;
    if (form != DEMOD_STENCIL_AUTO && form != DEMOD_STENCIL_PLAIN && form != DEMOD_STENCIL_FOLDED && form != DEMOD_STENCIL_NORMALIZED) {
        ctx->fail(DEMOD_E_INVAL, "stencil form must be DEMOD_STENCIL_AUTO, _PLAIN, _FOLDED or _NORMALIZED");
        return DEMOD_E_INVAL;
    }
    if (form != ctx->stencil_form && (host_pipe_busy(ctx) || ctx->rings)) {
        ctx->fail(DEMOD_E_INVAL, "the stencil form cannot change while submitted pieces are uncollected or a ring is open");
        return DEMOD_E_INVAL;
    }
    ctx->stencil_form = form;
    // 1: asked for, not available for this design (the plain form runs)
    if (form == DEMOD_STENCIL_FOLDED) return ctx->plan.fold_exact ? DEMOD_OK : 1;
    if (form == DEMOD_STENCIL_NORMALIZED) return ctx->plan.norm_ok ? DEMOD_OK : 1;
    return DEMOD_OK;
}

int demod_stencil_form(const demod_ctx *ctx) {
    if (!ctx) return DEMOD_E_INVAL;
    if (ctx->plan.conditioning != kShiftOriginPlain || ctx->plan.stencil_in.taps == 0) return DEMOD_STENCIL_PLAIN;
    const int c = rows_conditioning(ctx);
    return c == kShiftOrigin ? DEMOD_STENCIL_FOLDED : c == kShiftOriginNorm ? DEMOD_STENCIL_NORMALIZED : DEMOD_STENCIL_PLAIN;
}

int demod_set_unpack_fold(demod_ctx *ctx, int on) {
    return demod_set_stencil_form(ctx, on ? DEMOD_STENCIL_FOLDED : DEMOD_STENCIL_AUTO);
}

int demod_set_head_mode(demod_ctx *ctx, int mode) {
    if (!ctx) return DEMOD_E_INVAL;
    if (mode < -1 || mode > 2) {
        ctx->fail(DEMOD_E_INVAL, "head mode must be -1 (automatic), 0 (head kernels), 1 (inside the streaming kernel) or 2 (thread-per-chunk head kernels only)");
        return DEMOD_E_INVAL;
    }
    ctx->head_mode = mode;
    return DEMOD_OK;
}

int demod_set_piece_rows(demod_ctx *ctx, uint32_t rows) {
    if (!ctx) return DEMOD_E_INVAL;
    if (rows == 0) rows = kPieceRows;
    if ((rows & (rows - 1)) || rows < max_segment_rows()) {
        ctx->fail(DEMOD_E_INVAL, "piece rows must be a power of two, at least one segment");
        return DEMOD_E_INVAL;
    }
    ctx->piece_rows = rows;
    return DEMOD_OK;
}

int demod_last_call_launches(const demod_ctx *ctx) { return ctx ? ctx->last_row_launches : DEMOD_E_INVAL; }

void demod_enable_timing(demod_ctx *ctx, int on) {
    if (ctx) ctx->timing = on != 0;
}

double demod_kernel_ms(demod_ctx *ctx, int *launches) {
    if (launches) *launches = 0;
    if (!ctx) return -1.0;
    if (!ctx->timing) {
        ctx->timing = true;
        return -1.0;
    }
    if (ctx->timed.empty()) return -1.0;
    double total = 0.0;
    int n = 0;
    for (auto &p : ctx->timed) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess &&
            hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            total += ms;
            ++n;
        }
        ctx->event_pool.push_back(p);
    }
    ctx->timed.clear();
    if (launches) *launches = n;
    return n ? total / n : -1.0;
}

// ---------------------------------------------------------------------------
// the hot path on device-resident data
// ---------------------------------------------------------------------------

// Where the chunk heads are replayed.  The streaming kernel can do it itself (chunk_head:
// one head per wavefront, one output per lane) when both stages have at most 4 sections and
// highpassDc does not run in front of -L; whether it should is a matter of chunk size:
// a head costs that wavefront about a third of a 4-row segment's arithmetic (~270 VALU
// instructions with a quarter of the lanes busy), against ONE THREAD of the demod_heads kernel
// (64 heads per wavefront, but a launch of its own).  Measured on MI355X (tools/chunk_sweep.py,
// tools/head_mode_probe.py, profiles/r2_head_modes.txt): level at the default 256 KiB
// chunks and above (one kernel per dispatch), the separate kernel ahead below that (128 KiB:
// box-dependent, 64 KiB: 0.743 against 0.731 of peak, 16 KiB: 0.730 against 0.672, 1 KiB:
// 0.37 against 0.27).
// which unpack and complex stencil the streaming kernel's builds use: the plan's; for the default conditioning in
// front of a complex stage (-q0 with -L) the stencil divided by its first tap where build_plan allows it, the folded
// form where it was asked for and its constant is exact
static int rows_conditioning(const demod_ctx *ctx) {
    const Plan &pl = ctx->plan;
    if (pl.conditioning != kShiftOriginPlain || pl.stencil_in.taps == 0) return pl.conditioning;
    if (ctx->stencil_form == DEMOD_STENCIL_FOLDED && pl.fold_exact) return (int) kShiftOrigin;
    if ((ctx->stencil_form == DEMOD_STENCIL_AUTO || ctx->stencil_form == DEMOD_STENCIL_NORMALIZED) && pl.norm_ok) return (int) kShiftOriginNorm;
    return pl.conditioning;
}

static bool heads_are_fused(const demod_ctx *ctx) {
    const Plan &pl = ctx->plan;
    if (pl.conditioning == kCorrectIq) return false;  // -q1 runs in the bounds-checked build
    if (pl.head_quads == 0) return true;
    const bool can = pl.in.sections <= kFusedHeadSections && pl.out.sections <= kFusedHeadSections && pl.pre.sections == 0;
    if (!can || ctx->head_mode == 0 || ctx->head_mode == 2) return false;
    if (ctx->head_mode == 1) return true;
    const uint64_t rows_per_chunk = ctx->seamless ? (1ull << 31) : pl.chunk_bytes / kRowBytes;
    // Round 3 (tools/head_pack_probe.py, profiles/r3_head_modes.txt): without a complex stage a head is a few dozen
    // operations and the streaming kernel wins at every chunk size (1 KiB: 0.70 -> 0.76 of peak, 4 KiB: 0.78 -> 0.81,
    // 256 KiB: 0.81 -> 0.82); with one it wins where a head is rare (256 rows and more) and at one-row chunks, whose
    // four heads per segment are replayed side by side in ONE pass (0.61 -> 0.62-0.65); in between the head kernel
    // stays ahead (2 KiB: 0.71 against 0.65, 16 KiB: 0.78 against 0.71).
    if (pl.in.sections == 0) return true;
    if (rows_per_chunk == 1 && 4 * pl.head_quads + pl.in.sections + pl.out.sections + 2 <= 16) return true;
    return rows_per_chunk >= kFusedMinRows;
}

static void fill_stencils(const demod_ctx *ctx, RowLaunch &r) {
    const Plan &pl = ctx->plan;
    for (int j = 0; j < kMaxTaps; ++j) {
        r.hin[j] = pl.stencil_in.h[j];
        r.hout[j] = pl.out.h[j];
    }
    r.bias_im = pl.stencil_in.bias_im;
    {
        double sum = 0.0;
        for (int j = 0; j < pl.stencil_in.taps; ++j) sum += (double) pl.stencil_in.h[j];
        r.fold_re = (float) (-128.0 * sum);
        r.fold_im = (float) ((double) pl.stencil_in.bias_im - 128.0 * sum);
    }
    for (int j = 0; j < kMaxTaps; ++j) r.rho[j] = pl.rho[j];
    r.alpha = pl.norm_alpha;
    r.nk_re = -pl.norm_k_re;
    r.nk_im = -pl.norm_k_im;
    // exact: what the kernel's own alpha * 128 - k_im gives for a 0.0f sample
    r.past_im = (float) (128.0 * (double) pl.norm_alpha - (double) pl.norm_k_im);
    r.scale = (float) (2. / 255.);
    // the chunk head is replayed inside the streaming kernel when both stages fit its
    // lane-parallel form; otherwise (more than 4 sections, highpassDc in front of -L)
    // demod_heads runs behind the streaming kernel and overwrites the head outputs
    r.head_quads = (uint32_t) pl.head_quads;
    r.fused_heads = heads_are_fused(ctx) ? 1u : 0u;
    r.heads_packed = (4 * pl.head_quads + pl.in.sections + pl.out.sections + 2 <= 16) ? 1u : 0u;
    r.sec_in = pl.in.sections;
    r.sec_out = pl.out.sections;
    r.in_separate = pl.in_separate;
    if (r.fused_heads) {
        memcpy(r.sos_in, pl.in.sos, sizeof r.sos_in);
        memcpy(r.sos_out, pl.out.sos, sizeof r.sos_out);
    }
}

static void fill_sections(const Plan &pl, int &sec_in, int &sec_out, float (*sos_in)[6],
                          float (*sos_out)[6], float &scale, int &sec_pre, float (*sos_pre)[6]) {
    sec_pre = pl.pre.sections;
    memcpy(sos_pre, pl.pre.sos, 2 * 6 * sizeof(float));
    sec_in = pl.in.sections;
    sec_out = pl.out.sections;
    memcpy(sos_in, pl.in.sos, sizeof pl.in.sos);
    memcpy(sos_out, pl.out.sos, sizeof pl.out.sos);
    scale = (float) (2. / 255.);
}

// -q1: replay correctIq's offsets for the n_bytes at d_in (whole chunks of B bytes and, at a
// stream end, a short last one) into ctx->q1_table, starting from the state the previous call
// left (src/matrix.c:125: static, zero at process start).
static int run_offsets(demod_ctx *ctx, const uint8_t *d_in, uint64_t n_bytes, uint64_t B, hipStream_t stream) {
    const Plan &pl = ctx->plan;
    const uint64_t pairs = n_bytes / 4 + 4;
    if (!ctx->q1_carry) {
        HIP_TRY(ctx, hipMalloc((void **) &ctx->q1_carry, 4 * sizeof(float)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->q1_carry, 0, 4 * sizeof(float), stream));
    }
    if (ctx->q1_pairs < pairs) {
        if (ctx->q1_table) {
            HIP_TRY(ctx, hipDeviceSynchronize());  // an earlier call's kernels may still read it
            HIP_TRY(ctx, hipFree(ctx->q1_table));
        }
        ctx->q1_table = nullptr;
        ctx->q1_pairs = 0;
        HIP_TRY(ctx, hipMalloc((void **) &ctx->q1_table, pairs * 2 * sizeof(float)));
        ctx->q1_pairs = pairs;
    }
    OffsetLaunch o{};
    o.in = d_in;
    o.table = ctx->q1_table;
    o.carry_in = ctx->q1_carry + 2 * ctx->q1_flip;
    o.carry_out = ctx->q1_carry + 2 * (ctx->q1_flip ^ 1);
    ctx->q1_flip ^= 1;
    o.chunk_bytes = B;
    o.n_chunks = n_bytes / B;
    o.tail_bytes = (uint32_t) (n_bytes - o.n_chunks * B);
    o.stride = (uint32_t) (B / 4);
    o.warm = pl.q1_warm;
    o.retry = pl.q1_retry;
    // Every thread repeats the run-in before what it writes, so more threads shorten the chain
    // only down to the run-in's own length, and they multiply the work -- 16-byte requests to 64
    // different cache lines per load instruction, paid for in the CU's address unit.  A budget
    // of DEMOD_Q1_THREADS threads: long chunks are shared by `parts` threads (pieces of at least
    // 4096 iterations), short ones walked through `group` at a time by one thread.
    const uint64_t chunks = o.n_chunks + (o.tail_bytes ? 1 : 0);
    o.parts = o.group = 1;
    if (chunks >= DEMOD_Q1_THREADS)
        o.group = (uint32_t) ((chunks + DEMOD_Q1_THREADS - 1) / DEMOD_Q1_THREADS);
    else
        o.parts = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>({256, (B / 4) / 4096, DEMOD_Q1_THREADS / std::max<uint64_t>(1, chunks)}));
    o.esr = pl.esr;
    o.widen = pl.esr * (1.0f / 2048.0f);    // 2 J, J <= esr 2^-12: what the roundings of the small terms can undo (kernels.hip)
    o.floor2 = pl.esr * 8192.0f;            // twice esr 2^12: above esr 2^12 one ulp of the offsets exceeds J
    if (launch_offsets(o, stream)) {
        ctx->fail(DEMOD_E_HIP, "correct_iq_offsets launch", hipGetLastError());
        return DEMOD_E_HIP;
    }
    return DEMOD_OK;
}

static int run_tiny(demod_ctx *ctx, const uint8_t *d_in, uint64_t chunk_bytes, uint64_t n_chunks,
                    uint32_t tail_bytes, void *d_out, hipStream_t stream, const float *q1_table = nullptr,
                    uint32_t q1_stride = 0) {
    const Plan &pl = ctx->plan;
    const uint64_t stride = 6 * chunk_bytes + 3 * (4 * kMaxSections + 8);
    if (!ctx->tiny_done) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->tiny_done, hipEventDisableTiming));
    // one scratch per thread, as many threads as the call has chunks (up to kTinyThreads, whole
    // wavefronts): a hand-off of a few chunks does not pay for a long stream's scratch
    const uint64_t total = n_chunks + (tail_bytes ? 1 : 0);
    if (total == 0) return DEMOD_OK;
    const uint32_t threads = (uint32_t) ((std::min<uint64_t>(total, kTinyThreads) + 63) / 64 * 64);
    if (!ctx->tiny_work || ctx->tiny_stride < stride || ctx->tiny_threads < threads) {
        if (ctx->tiny_work) {
            HIP_TRY(ctx, hipDeviceSynchronize());  // an earlier launch may still be using it
            HIP_TRY(ctx, hipFree(ctx->tiny_work));
        }
        ctx->tiny_work = nullptr;
        ctx->tiny_pending = false;
        ctx->tiny_threads = 0;
        HIP_TRY(ctx, hipMalloc((void **) &ctx->tiny_work, (size_t) threads * stride * sizeof(float)));
        ctx->tiny_stride = stride;
        ctx->tiny_threads = threads;
    }
    // the scratch is shared: wait for the previous literal launch, whatever stream it ran on
    if (ctx->tiny_pending) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->tiny_done, 0));
    TinyLaunch t{};
    t.in = d_in;
    t.out = d_out;
    t.work = ctx->tiny_work;
    t.work_stride = ctx->tiny_stride;
    t.chunk_bytes = chunk_bytes;
    t.n_chunks = n_chunks;
    t.tail_bytes = tail_bytes;
    fill_sections(pl, t.sec_in, t.sec_out, t.sos_in, t.sos_out, t.scale, t.sec_pre, t.sos_pre);
    t.in_separate = pl.in_separate;
    t.out_s16 = ctx->out_s16;
    t.out_scale = ctx->out_scale;
    t.q1_table = q1_table;
    t.q1_stride = q1_stride;
    if (launch_tiny(t, pl.conditioning, threads, stream)) {
        ctx->fail(DEMOD_E_HIP, "demod_tiny launch", hipGetLastError());
        return DEMOD_E_HIP;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->tiny_done, stream));
    ctx->tiny_pending = true;
    return DEMOD_OK;
}

int demod_process_device(demod_ctx *ctx, const void *d_in_v, uint64_t n_bytes, void *d_out_v,
                         const demod_span *span, void *stream_v) {
    if (!ctx) return DEMOD_E_INVAL;
    if (n_bytes == 0) return DEMOD_OK;
    const uint8_t *d_in = static_cast<const uint8_t *>(d_in_v);
    char *d_out = static_cast<char *>(d_out_v);
    const uint64_t ob = ctx->out_bytes();  // bytes per output
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const Plan &pl = ctx->plan;
    // seamless mode: one chunk as long as any stream (2 TiB); only whether the
    // buffer starts the stream matters, so a non-zero offset is normalised to
    // one segment's worth of rows
    const uint64_t B = ctx->seamless ? (1ull << 41) : pl.chunk_bytes;

    if (!d_in || !d_out || (n_bytes & 3) || ((uintptr_t) d_in & 15) || ((uintptr_t) d_out & 15)) {
        ctx->fail(DEMOD_E_INVAL, "buffers must be non-null and 16-byte aligned, n_bytes a multiple of 4");
        return DEMOD_E_INVAL;
    }
    uint64_t offset = span ? span->chunk_offset : 0;
    const bool eos = span ? span->end_of_stream != 0 : true;
    if (ctx->seamless) {
        if (offset & (kRowBytes - 1)) {
            ctx->fail(DEMOD_E_INVAL, "chunk_offset must be a multiple of 1024");
            return DEMOD_E_INVAL;
        }
        offset = offset ? (uint64_t) max_segment_rows() * kRowBytes : 0;
    }
    const uint8_t *halo = span ? static_cast<const uint8_t *>(span->halo) : nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // -q1: correctIq's offsets run through the whole stream (static state, src/matrix.c:125), chunk
    // after chunk, call after call: buffers are whole chunks from a chunk boundary (a short last
    // one ends the stream), handed over in stream order on one HIP stream
    const bool q1 = pl.conditioning == kCorrectIq;
    if (q1) {
        if (ctx->seamless || offset != 0 || ((n_bytes % B) != 0 && !eos)) {
            ctx->fail(DEMOD_E_INVAL, "-q1 (correctIq) carries its offsets from chunk to chunk: a buffer must start on a chunk "
                                     "boundary and hold whole chunks (a short last chunk ends the stream); no seamless mode");
            return DEMOD_E_INVAL;
        }
    }
    // (the offsets pass consumes the stream's carried state: it runs once every check below has passed, so that a
    // rejected call leaves the state where it was)
    const uint32_t q1_stride = (uint32_t) (B / 4);

    // ---- chunks shorter than one row: literal whole-chunk kernel ----------
    if (B < kRowBytes) {
        if (offset != 0) {
            ctx->fail(DEMOD_E_INVAL, "chunk_offset must be 0 when the chunk size is below 1024 bytes");
            return DEMOD_E_INVAL;
        }
        const uint64_t full = n_bytes / B;
        const uint32_t tail = (uint32_t) (n_bytes - full * B);
        if (tail && !eos) {
            ctx->fail(DEMOD_E_INVAL, "buffer must hold whole chunks unless it ends the stream");
            return DEMOD_E_INVAL;
        }
        if (q1) {
            const int rc = run_offsets(ctx, d_in, n_bytes, B, stream);
            if (rc) return rc;
        }
        return run_tiny(ctx, d_in, B, full, tail, d_out, stream, ctx->q1_table, q1_stride);
    }

    if (offset >= B || (offset & (kRowBytes - 1))) {
        ctx->fail(DEMOD_E_INVAL, "chunk_offset must be a multiple of 1024 below the chunk size");
        return DEMOD_E_INVAL;
    }
    const uint64_t end_in_chunk = (offset + n_bytes) % B;  // 0: buffer ends on a chunk boundary
    if (end_in_chunk != 0 && !eos && !halo) {
        ctx->fail(DEMOD_E_INVAL, "buffer ends inside a chunk of a continuing stream: span.halo is required");
        return DEMOD_E_INVAL;
    }
    if ((n_bytes & (kRowBytes - 1)) && !eos) {
        ctx->fail(DEMOD_E_INVAL, "a continuing stream must be cut at multiples of 1024 bytes");
        return DEMOD_E_INVAL;
    }
    if (offset + n_bytes > (uint64_t) 0xFFFFFFFFu * kRowBytes) {
        ctx->fail(DEMOD_E_INVAL, "buffer too large for one call (limit 4 TiB)");
        return DEMOD_E_INVAL;
    }
    if (q1) {
        const int rc = run_offsets(ctx, d_in, n_bytes, B, stream);
        if (rc) return rc;
    }

    // A final short chunk of less than one row (it then starts inside this
    // buffer, at a row boundary) goes to the literal kernel; the rest is rows.
    uint64_t row_bytes = n_bytes;
    uint32_t tiny_tail = 0;
    if (eos && end_in_chunk != 0 && end_in_chunk < kRowBytes && end_in_chunk <= n_bytes) {
        tiny_tail = (uint32_t) end_in_chunk;
        row_bytes = n_bytes - end_in_chunk;
    }

    const uint32_t rpc_log2 = log2u(B / kRowBytes);
    uint32_t seg_log2 = log2u(ctx->seg_rows ? std::min(ctx->seg_rows, max_segment_rows()) : max_segment_rows());
    // (a segment spans several chunks when chunks are shorter than it: the kernels restart the
    // look-ahead at every chunk boundary inside a segment)

    // chunk heads that start inside the buffer
    const uint64_t first_head = offset == 0 ? 0 : B - offset;
    uint32_t n_heads = 0;
    if (first_head < row_bytes) n_heads = (uint32_t) ((row_bytes - first_head + B - 1) / B);
    uint64_t heads_from = 0;  // demod_heads replays the heads that start at or behind this byte of the buffer

    // ---- the fused streaming kernel ---------------------------------------
    if (row_bytes) {
        RowLaunch r{};
        r.in = d_in;
        r.out = d_out;
        r.halo = (!eos && end_in_chunk != 0) ? halo : nullptr;
        r.phase_rows = (uint32_t) (offset / kRowBytes);
        r.rpc_log2 = rpc_log2;
        r.seg_log2 = seg_log2;
        fill_stencils(ctx, r);
        r.out_s16 = ctx->out_s16;
        r.out_scale = ctx->out_scale;
        if (q1) {
            r.q1_table = ctx->q1_table;
            r.q1_stride = q1_stride;
            const uint64_t tail = n_bytes % B;  // a short last chunk (of at least one row: smaller ones go to the literal kernel)
            r.q1_last_chunk = tail ? (uint32_t) (n_bytes / B) : 0xFFFFFFFFu;
            r.q1_last_k = (uint32_t) (tail / 2);
        }

        ctx->last_row_launches = 0;
        const uint32_t ragged = (uint32_t) (row_bytes & (kRowBytes - 1));
        const uint32_t total_rows = (uint32_t) ((row_bytes + kRowBytes - 1) / kRowBytes);
        // A ragged last row (stream end only) runs in the bounds-checked build,
        // together with at most two segments before it; everything ahead of
        // that split runs in the fast build and may look ahead into the first
        // row after the split, which is a full row.
        uint32_t split = total_rows;
        if (q1) {
            split = 0;  // every row in the bounds-checked build, which reads the offsets table
        } else if (ragged) {
            const uint32_t seg = 1u << seg_log2;
            const uint32_t phase = (uint32_t) (offset / kRowBytes);
            const uint32_t keep = total_rows >= 2 ? total_rows - 2 : 0;       // last full row index, at most
            const uint32_t g = (phase + keep) / seg * seg;                    // segment boundary at or below it
            split = g > phase ? g - phase : 0;
        }
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        if (ctx->timing) {
            if (!ctx->event_pool.empty()) {
                ev = ctx->event_pool.back();
                ctx->event_pool.pop_back();
            } else {
                HIP_TRY(ctx, hipEventCreate(&ev.first));
                HIP_TRY(ctx, hipEventCreate(&ev.second));
            }
            HIP_TRY(ctx, hipEventRecord(ev.first, stream));
        }
        if (split) {
            // The full rows go out in pieces of at most kPieceRows (2 GiB of input):
            // on MI355X one 8 GiB launch runs ~4 % slower per sample than four 2 GiB
            // launches back to back (tools/split_probe.py).  Pieces are piece_rows
            // long and start on segment boundaries (counted from the chunk origin);
            // a piece looks ahead into the next piece's first row through
            // rows_readable.
            const uint32_t readable = ragged ? split + 1 : split;
            const uint32_t phase = r.phase_rows;
            uint32_t p0 = 0;
            while (p0 < split) {
                const uint32_t seg_rows = 1u << seg_log2;
                uint32_t p1 = (uint32_t) std::min<uint64_t>(
                    split, (uint64_t) p0 + ctx->piece_rows - ((phase + p0) & (seg_rows - 1u)));
                RowLaunch a = r;
                a.in = d_in + (uint64_t) p0 * kRowBytes;
                a.out = d_out + (uint64_t) p0 * kRowOutputs * ob;
                a.n_rows = p1 - p0;
                a.rows_readable = readable - p0;
                a.phase_rows = (uint32_t) (((uint64_t) phase + p0) & ((1ull << rpc_log2) - 1));
                a.ragged_bytes = 0;
                if (ragged || p1 < split) a.halo = nullptr;  // only the buffer's last rows can need the shard halo
                if (launch_rows(a, pl.stencil_in.taps, pl.out.taps, rows_conditioning(ctx), stream)) {
                    ctx->fail(DEMOD_E_HIP, "demod_rows launch", hipGetLastError());
                    if (ev.first) ctx->event_pool.push_back(ev);   // (the timing pair goes back where it came from)
                    return DEMOD_E_HIP;
                }
                ++ctx->last_row_launches;
                p0 = p1;
            }
        }
        if (ragged || (q1 && total_rows)) {
            RowLaunch a = r;
            a.in = d_in + (uint64_t) split * kRowBytes;
            a.out = d_out + (uint64_t) split * kRowOutputs * ob;
            a.n_rows = total_rows - split;
            a.rows_readable = a.n_rows;
            a.phase_rows = (uint32_t) (((offset + (uint64_t) split * kRowBytes) % B) / kRowBytes);
            a.ragged_bytes = ragged;
            a.halo = nullptr;
            if (launch_rows(a, pl.stencil_in.taps, pl.out.taps, rows_conditioning(ctx), stream)) {
                ctx->fail(DEMOD_E_HIP, "demod_rows (ragged) launch", hipGetLastError());
                if (ev.first) ctx->event_pool.push_back(ev);
                return DEMOD_E_HIP;
            }
            ++ctx->last_row_launches;
        }
        if (ctx->timing) {
            HIP_TRY(ctx, hipEventRecord(ev.second, stream));
            ctx->timed.push_back(ev);
        }
        // the streaming kernel replays the heads of the chunks whose first row its fast build
        // computes; the bounds-checked build (ragged stream end) leaves its chunks' heads out
        if (r.fused_heads) heads_from = (uint64_t) split * kRowBytes;
    }

    // ---- chunk heads the streaming kernel did not replay (literal, one thread per chunk) ----
    if (n_heads && pl.head_quads && first_head + (uint64_t) (n_heads - 1) * B >= heads_from) {
        const uint32_t skip = heads_from > first_head ? (uint32_t) ((heads_from - first_head + B - 1) / B) : 0u;
        HeadLaunch h{};
        h.in = d_in + first_head + (uint64_t) skip * B;
        h.out = d_out + (first_head + (uint64_t) skip * B) / 4 * ob;
        h.out_s16 = ctx->out_s16;
        h.out_scale = ctx->out_scale;
        h.chunk_bytes = B;
        h.n_heads = n_heads - skip;
        h.head_outputs = 4u * (uint32_t) pl.head_quads;
        const uint64_t last_len = row_bytes - (first_head + (uint64_t) (n_heads - 1) * B);
        // a last chunk that continues past the buffer is a full chunk for the
        // head window; only a stream end makes it short
        h.last_chunk_bytes = (uint32_t) std::min<uint64_t>((eos && last_len < B) ? last_len : B, 1u << 30);
        fill_sections(pl, h.sec_in, h.sec_out, h.sos_in, h.sos_out, h.scale, h.sec_pre, h.sos_pre);
        h.in_separate = pl.in_separate;
        h.thread_per_head = ctx->head_mode == 2;
        if (q1) {
            h.q1_table = ctx->q1_table + 2ull * skip * q1_stride;  // (buffers start on a chunk boundary: head i is chunk i)
            h.q1_stride = q1_stride;
        }
        if (launch_heads(h, pl.conditioning, stream)) {
            ctx->fail(DEMOD_E_HIP, "demod_heads launch", hipGetLastError());
            return DEMOD_E_HIP;
        }
    }

    if (tiny_tail) {
        const int rc = run_tiny(ctx, d_in + row_bytes, kRowBytes, 0, tiny_tail, d_out + row_bytes / 4 * ob, stream,
                                q1 ? ctx->q1_table + 2ull * (row_bytes / B) * q1_stride : nullptr, q1_stride);
        if (rc) return rc;
    }
    return DEMOD_OK;
}

// ---------------------------------------------------------------------------
// memory helpers, introspection, synthetic input
// ---------------------------------------------------------------------------

int demod_alloc_pinned(void **p, size_t len) {
    if (!p) return DEMOD_E_INVAL;
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, len ? len : 1, hipHostMallocDefault);
    if (e != hipSuccess) return create_fail(DEMOD_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    memset(*p, 0, len);  // the reference's allocateBuffer callocs (src/matrix.c:282-285)
    return DEMOD_OK;
}

void demod_free_pinned(void *p) {
    if (p) (void) hipHostFree(p);
}

int demod_alloc_device(demod_ctx *ctx, void **p, size_t len) {
    if (!ctx || !p) return DEMOD_E_INVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(p, len ? len : 1));
    return DEMOD_OK;
}

void demod_free_device(demod_ctx *ctx, void *p) {
    if (ctx && p) {
        (void) hipSetDevice(ctx->device);
        (void) hipFree(p);
    }
}

int demod_copy_to_device(demod_ctx *ctx, void *d_dst, const void *h_src, size_t len) {
    if (!ctx) return DEMOD_E_INVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(d_dst, h_src, len, hipMemcpyHostToDevice));
    return DEMOD_OK;
}

int demod_copy_to_host(demod_ctx *ctx, void *h_dst, const void *d_src, size_t len) {
    if (!ctx) return DEMOD_E_INVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(h_dst, d_src, len, hipMemcpyDeviceToHost));
    return DEMOD_OK;
}

int demod_synchronize(demod_ctx *ctx) {
    if (!ctx) return DEMOD_E_INVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    return DEMOD_OK;
}

static const Stage &stage_of(const Plan &pl, int which) {
    return which == 2 ? pl.pre : which ? pl.in : pl.out;
}
// the stencil of a stage as the streaming kernel applies it: for the input stage the
// composition with highpassDc when both are on (Plan::stencil_in)
static const Stage &stencil_of(const Plan &pl, int which) {
    return which == 2 ? pl.pre : which ? pl.stencil_in : pl.out;
}

int demod_get_sos(const demod_ctx *ctx, int which, float *rows6, int max_rows) {
    if (!ctx) return DEMOD_E_INVAL;
    const Stage &st = stage_of(ctx->plan, which);
    const int n = std::min(st.sections, max_rows);
    if (rows6)
        for (int r = 0; r < n; ++r) memcpy(rows6 + 6 * r, st.sos[r], 6 * sizeof(float));
    return st.sections;
}

int demod_get_taps(const demod_ctx *ctx, int which, float *taps, int max_taps, float *bias_im) {
    if (!ctx) return DEMOD_E_INVAL;
    const Stage &st = stencil_of(ctx->plan, which);
    if (taps)
        for (int j = 0; j < std::min(st.taps, max_taps); ++j) taps[j] = st.h[j];
    if (bias_im) *bias_im = st.bias_im;
    return st.taps;
}

int demod_get_head_outputs(const demod_ctx *ctx) { return ctx ? ctx->plan.head_outputs : DEMOD_E_INVAL; }

int demod_design_query(const demod_params *p, int which, float *rows6, int max_rows, float *taps,
                       int max_taps, float *bias_im, int *n_taps, int *head_outputs) {
    if (!p) return create_fail(DEMOD_E_INVAL, "null argument");
    Plan plan;
    std::string err;
    if (!build_plan(options_from_params(*p), plan, err)) return create_fail(DEMOD_E_INVAL, err);
    const Stage &st = stage_of(plan, which);
    const Stage &sn = stencil_of(plan, which);
    if (rows6)
        for (int r = 0; r < std::min(st.sections, max_rows); ++r) memcpy(rows6 + 6 * r, st.sos[r], 6 * sizeof(float));
    if (taps)
        for (int j = 0; j < std::min(sn.taps, max_taps); ++j) taps[j] = sn.h[j];
    if (bias_im) *bias_im = sn.bias_im;
    if (n_taps) *n_taps = sn.taps;
    if (head_outputs) *head_outputs = plan.head_outputs;
    return st.sections;
}

double demod_design_magnitude_bound(const demod_params *p) {
    if (!p) return (double) create_fail(DEMOD_E_INVAL, "null argument");
    Plan plan;
    std::string err;
    if (!build_plan(options_from_params(*p), plan, err)) return (double) create_fail(DEMOD_E_INVAL, err);
    return plan.magnitude_bound;
}

int demod_design_unpack_folded(const demod_params *p) {
    if (!p) return create_fail(DEMOD_E_INVAL, "null argument");
    Plan plan;
    std::string err;
    if (!build_plan(options_from_params(*p), plan, err)) return create_fail(DEMOD_E_INVAL, err);
    return plan.fold_exact ? 1 : 0;
}

int demod_design_stencil_form(const demod_params *p, double *normalized_error, double *plain_error) {
    if (!p) return create_fail(DEMOD_E_INVAL, "null argument");
    Plan plan;
    std::string err;
    if (!build_plan(options_from_params(*p), plan, err)) return create_fail(DEMOD_E_INVAL, err);
    if (normalized_error) *normalized_error = plan.norm_error + plan.norm_bias_error;
    if (plain_error) *plain_error = plan.plain_error;
    return plan.norm_ok ? DEMOD_STENCIL_NORMALIZED : DEMOD_STENCIL_PLAIN;
}

int demod_design_normalized_constants(const demod_params *p, float *alpha, float *k_re, float *k_im, float *rho, int max_taps) {
    if (!p) return create_fail(DEMOD_E_INVAL, "null argument");
    Plan plan;
    std::string err;
    if (!build_plan(options_from_params(*p), plan, err)) return create_fail(DEMOD_E_INVAL, err);
    if (alpha) *alpha = plan.norm_alpha;
    if (k_re) *k_re = plan.norm_k_re;
    if (k_im) *k_im = plan.norm_k_im;
    const int n = plan.stencil_in.taps;
    for (int j = 0; rho && j < n && j < max_taps; ++j) rho[j] = plan.rho[j];
    return n;
}

int demod_synth_fill(demod_ctx *ctx, void *d_dst, uint64_t first_byte, uint64_t n_bytes, uint64_t seed,
                     void *stream) {
    if (!ctx || !d_dst || (first_byte & 15) || (n_bytes & 15) || ((uintptr_t) d_dst & 15)) {
        if (ctx) ctx->fail(DEMOD_E_INVAL, "synth_fill: 16-byte alignment required");
        return DEMOD_E_INVAL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (launch_synth_fill(d_dst, first_byte, n_bytes, seed, stream)) {
        ctx->fail(DEMOD_E_HIP, "synth_fill launch", hipGetLastError());
        return DEMOD_E_HIP;
    }
    return DEMOD_OK;
}

// Device self-tests used by the GPU test-suite (not part of the data path).
int demod_selftest_atan2(demod_ctx *ctx, const float *h_y, const float *h_x, float *h_out, uint32_t n) {
    if (!ctx || !n) return DEMOD_E_INVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **) &d, 3ull * n * sizeof(float)));
    int rc = DEMOD_OK;
    if (hipMemcpy(d, h_y, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, h_x, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        launch_atan2_probe(d, d + n, d + 2ull * n, n, nullptr) != 0 ||
        hipMemcpy(h_out, d + 2ull * n, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        ctx->fail(DEMOD_E_HIP, "atan2 self-test", hipGetLastError());
        rc = DEMOD_E_HIP;
    }
    (void) hipFree(d);
    return rc;
}

int demod_selftest_q1_offsets(demod_ctx *ctx, float *h_pairs, uint64_t n_pairs, float *h_state2) {
    if (!ctx) return DEMOD_E_INVAL;
    if (ctx->plan.conditioning != kCorrectIq) {
        ctx->fail(DEMOD_E_INVAL, "not a -q1 context");
        return DEMOD_E_INVAL;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    if (h_pairs && n_pairs) {
        if (!ctx->q1_table || n_pairs > ctx->q1_pairs) {
            ctx->fail(DEMOD_E_INVAL, "more offsets asked for than the last call's table holds");
            return DEMOD_E_INVAL;
        }
        HIP_TRY(ctx, hipMemcpy(h_pairs, ctx->q1_table, n_pairs * 2 * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (h_state2) {
        h_state2[0] = h_state2[1] = 0.f;   // no call yet: the reference's static offsets start at zero
        if (ctx->q1_carry) HIP_TRY(ctx, hipMemcpy(h_state2, ctx->q1_carry + 2 * ctx->q1_flip, 2 * sizeof(float), hipMemcpyDeviceToHost));
    }
    return DEMOD_OK;
}

int demod_selftest_lane_shift(demod_ctx *ctx, const float *h_in64, float halo, float *h_out64) {
    if (!ctx) return DEMOD_E_INVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **) &d, 128 * sizeof(float)));
    int rc = DEMOD_OK;
    if (hipMemcpy(d, h_in64, 64 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        launch_lane_probe(d, d + 64, halo, nullptr) != 0 ||
        hipMemcpy(h_out64, d + 64, 64 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        ctx->fail(DEMOD_E_HIP, "lane-shift self-test", hipGetLastError());
        rc = DEMOD_E_HIP;
    }
    (void) hipFree(d);
    return rc;
}

}  // extern "C"
