/*
 * demod_hip.h -- C ABI of the MI355X-native FM-demodulation hot path.
 *
 * This is the drop-in boundary: a shared library (libdemod_hip.so) with plain
 * pointers and sizes, no C++/torch types.  It replaces the reference's
 * per-chunk pipeline
 *
 *     shiftOrigin -> [applyComplexFilter] -> fmDemod -> applyFilter
 *     (/root/reference/src/matrix.c:236-273, src/filter.c:212-259)
 *
 * and the once-per-run filter design behind it (src/matrix.c:25-80,
 * src/filter.c:22-210).  The reference's own two-symbol backend boundary
 * (processMatrix / allocateBuffer, include/matrix.h:59-62) is provided as a
 * thin adapter over this ABI in demod_backend.h.
 *
 * Units: one IQ sample = 2 input bytes (uint8 I, uint8 Q).  4 input bytes
 * produce 1 float32 output (the discriminator consumes non-overlapping pairs
 * of complex samples, src/matrix.c:166-175).
 */
#ifndef DEMOD_HIP_H
#define DEMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEMOD_ABI_VERSION 4

/* Default chunk size, bytes (include/matrix.h:37-39). */
#define DEMOD_DEFAULT_BUF_SIZE 262144u

/* Bytes of look-ahead a shard needs from the stream that follows it when the
 * shard does not end on a chunk boundary (see demod_span). */
#define DEMOD_HALO_BYTES 64u

/* Highest supported filter degree (-d / -D). */
#define DEMOD_MAX_DEGREE 16u

/* Error codes (negative). */
enum {
    DEMOD_OK = 0,
    DEMOD_E_INVAL = -1,     /* bad argument / unsupported option set */
    DEMOD_E_NODEVICE = -2,  /* no usable HIP device */
    DEMOD_E_HIP = -3,       /* a HIP runtime call failed (see demod_last_error) */
    DEMOD_E_NOMEM = -4
};

/* Options that reach the hot path.  Same meaning and defaults as the fields of
 * the reference's consumerArgs that processMatrix reads
 * (include/matrix.h:43-57; defaults src/main.c:102-113). */
typedef struct demod_params {
    double sample_rate;   /* -S  default 125000 */
    double lowpass_in;    /* -L  default 0 = no input (complex) lowpass */
    double lowpass_out;   /* -l  default 12500 */
    double epsilon;       /* -e x is stored as x/10; default .3 (Chebyshev only) */
    uint32_t in_degree;   /* -D  default 0 (= use out_degree when -L is given, src/matrix.c:190-192) */
    uint32_t out_degree;  /* -d  default 3 */
    uint32_t mode;        /* ww|dd|qq|ff byte, default 0x10 (src/main.c:112):
                             bit0 output filter Chebyshev-I, bit1 input filter
                             Chebyshev-I, bits2-3 IQ conditioning (-q: 0
                             shiftOrigin, 1 correctIq, 2 highpassDc, 3
                             normalizeInput), bits4-5 demod mode */
    uint32_t reserved;
    uint64_t buf_size;    /* chunk bytes B (DEFAULT_BUF_SIZE shifted by -b); power of two >= 4 */
} demod_params;

/* Where a device buffer sits in the stream.  Chunks are the reference's unit
 * of filtering: every chunk is filtered on its own with zero padding
 * (src/matrix.c:272), so chunk boundaries are part of the numerical result. */
typedef struct demod_span {
    uint64_t chunk_offset;   /* bytes between the start of the enclosing chunk and
                                the first byte of the buffer; multiple of 1024
                                (0 = buffer starts on a chunk boundary) */
    const void *halo;        /* DEVICE pointer to the DEMOD_HALO_BYTES bytes that
                                follow the buffer in the stream, or NULL.  Only
                                read when the buffer ends inside a chunk. */
    uint32_t end_of_stream;  /* 1: the stream ends with this buffer; a trailing
                                partial chunk of r bytes is filtered as a short
                                chunk of (r & ~3) bytes.  0: the stream continues
                                (halo must be given if the end is inside a chunk) */
    uint32_t reserved;
} demod_span;

/* A context belongs to one HIP device and is NOT thread-safe: use it from one
 * host thread at a time (several contexts may be used concurrently). */
typedef struct demod_ctx demod_ctx;

/* Fill p with the reference's defaults. */
void demod_default_params(demod_params *p);

/* ABI version of the loaded library. */
int demod_abi_version(void);

/* Create a context on HIP device `device`: designs the filters on the host,
 * derives the stencil taps and uploads nothing until the first process call.
 * Fails with DEMOD_E_INVAL for option sets the reference leaves undefined
 * (-D without -L, src/matrix.c:227-231) or that are out of scope (demod mode 0),
 * with a message in demod_last_error(NULL).
 * -q1 (correctIq, src/matrix.c:120-140) keeps running offsets through the whole
 * stream, as the reference's static state does: the context carries them from call
 * to call, so its buffers must be whole chunks from a chunk boundary (a short last
 * chunk ends the stream), handed over in stream order on one HIP stream. */
int demod_create(const demod_params *p, int device, demod_ctx **out);
void demod_destroy(demod_ctx *ctx);

/* Last error text of a context (or of demod_create when ctx == NULL), and the
 * DEMOD_E_* code of the failure it describes. */
const char *demod_last_error(const demod_ctx *ctx);
int demod_last_error_code(const demod_ctx *ctx);

/* HOT PATH.  Demodulate n_bytes of uint8 IQ resident in device memory at d_in
 * into n_bytes/4 float32 (or int16, see demod_set_output_format) at d_out, asynchronously on `stream` (a hipStream_t
 * passed as void*, NULL = default stream).  d_in must be 16-byte aligned,
 * d_out 16-byte aligned, n_bytes a multiple of 4.  span == NULL means: buffer
 * starts on a chunk boundary and the stream ends with it.
 * One call = one launch of the fused kernel per 2 GiB of input (the chunk heads are
 * replayed inside it; for chunks below 256 KiB, more than 4 filter sections or a ragged
 * stream end, by a small demod_heads launch behind it on the same stream). */
int demod_process_device(demod_ctx *ctx, const void *d_in, uint64_t n_bytes, void *d_out,
                         const demod_span *span, void *stream);

/* HOST-BUFFER PATHS (the reference's main.c:58-98 producer loop and the consumer's
 * memcpy/fwrite around the per-chunk pipeline, for a device behind PCIe).  All three are
 * a three-deep pipeline on three HIP streams -- upload, hot path, download -- chained by
 * events: the upload of piece i+1, the kernels of piece i and the download of piece i-1
 * overlap.
 *
 * demod_process_host: a whole host buffer; returns when h_out is complete.  Caller memory
 * from demod_alloc_pinned is DMA'd in place; pageable memory is staged through pinned
 * buffers by a few copy threads.  Same span semantics as above (halo is a HOST pointer). */
int demod_process_host(demod_ctx *ctx, const void *h_in, uint64_t n_bytes, void *h_out,
                       const demod_span *span);

/* One hand-off at a time: submit copies n_bytes (<= demod_stream_capacity()) into a pinned
 * staging stage and enqueues upload + hot path + download, then returns (the caller's
 * buffer is free again); collect waits for the OLDEST submitted piece and hands out its
 * results (pinned memory owned by the context, valid until demod_stream_depth() further
 * submits).  demod_stream_depth() pieces may be in flight; submit returns DEMOD_E_INVAL
 * when all are. */
uint64_t demod_stream_capacity(void);
int demod_stream_depth(void);
int demod_stream_submit(demod_ctx *ctx, const void *h_in, uint64_t n_bytes, const demod_span *span);
int demod_stream_collect(demod_ctx *ctx, const void **out, uint64_t *n_outputs);

/* Streaming ring, the producer/consumer form (CLI, backend adapter): hand-offs of one chunk
 * each are laid one after the other in a pinned ring -- the producer reads its input
 * straight into the slot demod_ring_input() names, nothing is copied -- and runs of them
 * travel as ONE upload + ONE hot-path call + ONE download: whatever has arrived while the
 * GPU was busy (at most 8 MiB), at once when it is idle, so a slow producer's chunks leave
 * one by one and a fast producer's in batches.
 *   demod_ring_slots_for(B)  slots a ring for chunk size B wants (0: no ring for this size --
 *                            below 1 KiB or above the stage capacity; use submit/collect)
 *   demod_ring_create        pinned_in = NULL: the ring allocates its input arena; else a
 *                            pinned arena of n_slots * B bytes the caller owns (how
 *                            allocateBuffer hands the producer its buffer before a context exists)
 *   demod_ring_input         where the next hand-off goes (blocks while the ring is full)
 *   demod_ring_commit        n_bytes there are valid: one whole chunk, or -- only with
 *                            final = 1 -- fewer (multiple of 4; 0 = nothing more); never copies
 *   demod_ring_collect       results of the oldest batch, in stream order; blocks until they
 *                            are on the host; *n_outputs = 0 once the stream has ended and
 *                            everything was handed out
 *   demod_ring_release       the collected results have been consumed
 * commit (one thread) and collect/release (another thread) may run concurrently. */
typedef struct demod_ring demod_ring;
uint32_t demod_ring_slots_for(uint64_t chunk_bytes);
int demod_ring_create(demod_ctx *ctx, void *pinned_in, uint32_t n_slots, demod_ring **ring);
void demod_ring_destroy(demod_ring *ring);
void *demod_ring_input(demod_ring *ring);
int demod_ring_commit(demod_ring *ring, uint64_t n_bytes, int final);
int demod_ring_collect(demod_ring *ring, const void **out, uint64_t *n_outputs);
int demod_ring_release(demod_ring *ring);

/* SEVERAL GPUS FROM ONE PROCESS.  The reference has one producer and one consumer thread
 * and nothing else (src/main.c:67-70); its chunks are filtered independently
 * (src/matrix.c:272), so a stream splits into contiguous shards, one per device:
 *   - cuts on chunk boundaries need no exchange at all (BASELINE config 4 as specified);
 *   - a cut inside a chunk makes the shard on its left need the DEMOD_HALO_BYTES that
 *     follow it: one ncclSend / ncclRecv pair per cut, grouped, on the devices' streams
 *     (RCCL over xGMI; librccl is dlopen'ed on first need; without it, or if it refuses the
 *     device list, the bytes travel with hipMemcpyPeerAsync -- demod_multi_halo_transport()
 *     returns "rccl", "peer-copy" or "none"; DEMOD_MULTI_TRANSPORT=peer forces the copy);
 *   - independent channels (BASELINE config 5) are whole-chunk streams of their own: a
 *     device's channels laid back to back go through one call, no communication.
 * demod_plan_shards / demod_plan_channels are host-only (no device needed).
 * Measured on one GPU only (n = 1, and n = 2 on the same device with peer copies): a node
 * with several MI355X has not been available to this repository's tests. */
typedef struct demod_shard {
    uint64_t start;         /* first byte of the shard in the global stream */
    uint64_t length;        /* bytes (may be 0: fewer rows than devices) */
    uint64_t chunk_offset;  /* bytes between the enclosing chunk's start and `start` */
    int32_t needs_halo;     /* the shard ends inside a chunk and bytes follow it in the stream */
    int32_t end_of_stream;  /* nothing follows this shard */
    int32_t halo_src;       /* shard that holds the bytes following this one (needs_halo), else -1 */
    int32_t halo_dst;       /* shard whose look-ahead this shard's first bytes are, else -1 */
} demod_shard;

/* total_bytes cut into n contiguous shards of (nearly) equal size; the stream itself starts
 * phase_bytes (multiple of 1024, below the chunk size) into a chunk; cuts are multiples of
 * `align` bytes (0 = the chunk size when it divides the even share and phase is 0, else
 * 1 KiB rows).  Empty shards take no part in any exchange. */
int demod_plan_shards(uint64_t total_bytes, int n, uint64_t chunk_bytes, uint64_t phase_bytes, uint64_t align,
                      demod_shard *shards);
/* n_channels independent channels of bytes_per_channel (whole chunks) each, whole channels
 * per device: shard r = the bytes of device r's channels laid back to back (start counts
 * from the first channel's first byte). */
int demod_plan_channels(uint64_t n_channels, uint64_t bytes_per_channel, int n, uint64_t chunk_bytes, demod_shard *shards);

typedef struct demod_multi demod_multi;
/* One context and one stream per device of the list (a device may not appear twice when
 * halos travel over RCCL). */
int demod_multi_create(const demod_params *p, const int *devices, int n_devices, demod_multi **out);
void demod_multi_destroy(demod_multi *m);
int demod_multi_devices(const demod_multi *m);
demod_ctx *demod_multi_context(demod_multi *m, int i);  /* device i's context: allocation, synth_fill, knobs */
void *demod_multi_stream(demod_multi *m, int i);        /* device i's hipStream_t */
int demod_multi_plan(const demod_multi *m, uint64_t total_bytes, uint64_t phase_bytes, demod_shard *shards);
/* One step: d_in[i] / d_out[i] are shard i's buffers in device i's memory.  Exchanges the
 * halos the plan asks for and enqueues the hot path on every device's stream; returns
 * without waiting. */
int demod_multi_process_device(demod_multi *m, void *const *d_in, void *const *d_out, const demod_shard *shards);
int demod_multi_synchronize(demod_multi *m);
const char *demod_multi_halo_transport(const demod_multi *m);
const char *demod_multi_last_error(const demod_multi *m);

/* Pinned host memory (the reference's allocateBuffer hook,
 * src/matrix.c:282-285 / cuda/matrix.cu:106-109). */
int demod_alloc_pinned(void **p, size_t len);
void demod_free_pinned(void *p);

/* Device memory helpers so a C host needs no HIP headers. */
int demod_alloc_device(demod_ctx *ctx, void **p, size_t len);
void demod_free_device(demod_ctx *ctx, void *p);
int demod_copy_to_device(demod_ctx *ctx, void *d_dst, const void *h_src, size_t len);
int demod_copy_to_host(demod_ctx *ctx, void *h_dst, const void *d_src, size_t len);
int demod_synchronize(demod_ctx *ctx);

/* Introspection (tests): the designed second-order sections
 * {b0,b1,b2,a0,a1,a2} as float rows, the stencil taps derived from them and
 * the per-chunk head length that is replayed literally (chunk_head / demod_heads).
 * which: 0 = output filter, 1 = input filter (-L, or the -q2 highpass when alone), 2 = the -q2
 * highpass when it runs in front of -L.  Taps of the input filter are those of the stencil
 * the streaming kernel applies: with -q2 and -L both on, the composition of the two.
 * Return the row / tap count. */
int demod_get_sos(const demod_ctx *ctx, int which, float *rows6, int max_rows);
int demod_get_taps(const demod_ctx *ctx, int which, float *taps, int max_taps, float *bias_im);
int demod_get_head_outputs(const demod_ctx *ctx);

/* The same design without a context or a device (host only): fills what is
 * non-NULL and returns the number of sections of the selected filter, or a
 * negative error code with the text in demod_last_error(NULL). */
int demod_design_query(const demod_params *p, int which, float *rows6, int max_rows, float *taps,
                       int max_taps, float *bias_im, int *n_taps, int *head_outputs);

/* Largest magnitude any float of the path can take for an option set (samples,
 * filtered samples, overflow cells, discriminator arguments, outputs), proven
 * from the design alone (host only).  The reference turns a NaN discriminator
 * value into 0 (src/matrix.c:174); NaN needs non-finite arithmetic, and context
 * creation refuses every design whose bound reaches 1e30 -- for an accepted
 * design the rule never fires, here or in the reference.  Returns the bound, or
 * a negative error code (as a double) with the text in demod_last_error(NULL). */
double demod_design_magnitude_bound(const demod_params *p);

/* The streaming kernel has three forms of the complex stencil behind the default (shiftOrigin) unpack
 * (-q0 with -L; DESIGN.md sections 4.1 and 5).  All compute z[k] = sum_j h[j] x[k+j] + i bias.
 *   PLAIN       x = byte - 128, then one fused multiply-add per tap from (0, bias).
 *   NORMALIZED  z / h[0]: X[k] + sum_{j>=1} (h[j]/h[0]) X[k+j] with X = alpha (x, x + s), s = bias / sum(h), straight
 *               from the unpack (one packed multiply-add, exact, where PLAIN has a packed subtraction; alpha is
 *               chosen at context creation so that its constants are floats).  The discriminator takes the
 *               ARGUMENT of z[2n] conj(z[2n+1]), which a common positive factor does not turn: one fused
 *               multiply-add per sample less, one rounding per filtered sample instead of one per tap.  The
 *               taps h[j]/h[0] are floats; context creation prices what their rounding can move a full-scale
 *               filtered sample by against what the plain form's intermediate rounding can, and offers the
 *               form where that is at most an eighth of it (the BASELINE designs: 4e-8 against 7.6e-6; about
 *               one design in six).
 *   FOLDED      byte + 128-complement handed to the stencil, the -128 absorbed in its constant: at samples
 *               that filter to next to nothing it differs from the reference by more than the reference's
 *               own builds differ from each other; available where that constant is exactly a float.
 * DEMOD_STENCIL_AUTO (the default) = NORMALIZED where the design allows it, PLAIN otherwise.
 * demod_set_stencil_form returns 0, or 1 when the form asked for is not available for the context's design
 * (the plain form runs), negative on error; demod_stencil_form says which form a context runs;
 * demod_design_stencil_form which one AUTO gives an option set (and the two error bounds, in the
 * reference's units of a filtered sample; host only).  demod_design_unpack_folded: 1 when FOLDED is
 * available.  demod_set_unpack_fold(ctx, on) = demod_set_stencil_form(ctx, on ? FOLDED : AUTO).
 * There is no environment switch: the arithmetic of a context changes only through these calls -- and not in the
 * middle of a stream: a change of form (like one of the output format or of seamless mode) is refused with
 * DEMOD_E_INVAL while submitted pieces are uncollected or a ring is open. */
#define DEMOD_STENCIL_AUTO 0
#define DEMOD_STENCIL_PLAIN 1
#define DEMOD_STENCIL_FOLDED 2
#define DEMOD_STENCIL_NORMALIZED 3
int demod_set_stencil_form(demod_ctx *ctx, int form);
int demod_stencil_form(const demod_ctx *ctx);
int demod_design_stencil_form(const demod_params *p, double *normalized_error, double *plain_error);
int demod_design_unpack_folded(const demod_params *p);
int demod_set_unpack_fold(demod_ctx *ctx, int on);
/* The NORMALIZED form's constants for an option set (host only; what the tests check the exactness claims on):
 * the unpack X = alpha * (sample + 128) - (k_re, k_im) and the taps rho[j] = h[j] / h[0] (rho[0] = 1).  Returns
 * the number of taps of the complex stencil (0: no complex stage), negative on error; whether the form is OFFERED for
 * the design is demod_design_stencil_form's answer (the constants are what the search found either way). */
int demod_design_normalized_constants(const demod_params *p, float *alpha, float *k_re, float *k_im, float *rho, int max_taps);

/* Seamless mode (not reference behaviour; off by default): the whole stream is
 * filtered as ONE chunk, so the look-ahead stencils run across what would be
 * chunk boundaries instead of restarting with zero padding every buf_size bytes
 * (no edge outputs inside the stream; the result equals the reference run with
 * a chunk as long as the stream).  span.chunk_offset is then the buffer's
 * offset from the START OF THE STREAM (0 or any multiple of 1024) and every
 * buffer but the last needs span.halo. */
int demod_set_seamless(demod_ctx *ctx, int on);

/* Output format (not reference behaviour; float32 by default).  DEMOD_OUT_S16
 * fuses the float32 -> signed 16-bit hop that follows the reference in its
 * README pipes (`sox -ef -b32 ... -es -b16`, README.md:22,29) into the kernel's
 * store: every result leaves as clamp(y * scale, -1, 1) * 32767 rounded to
 * nearest, 2 bytes per output instead of 4 (d_out / h_out / collect then hold
 * int16_t; 3 instead of 4 bytes of HBM traffic per IQ sample).  scale = 1
 * clips at +-1 rad as sox does with the reference's output; 1 / (pi * sum |h_out|)
 * maps the discriminator's full range onto the full int16 range.
 * demod_output_bytes: size of one output in the current format (4 or 2). */
#define DEMOD_OUT_F32 0
#define DEMOD_OUT_S16 1
int demod_set_output_format(demod_ctx *ctx, int format, float scale);
int demod_output_bytes(const demod_ctx *ctx);

/* Tuning knob: rows (1 KiB of input each) a wavefront streams through before
 * it needs a fresh look-ahead; power of two, 0 = automatic. */
int demod_set_segment_rows(demod_ctx *ctx, uint32_t rows);

/* Tuning knob: where the first outputs of every chunk (partial sections, overflow
 * cells; not interior) are replayed: 1 = by the wavefront of the streaming kernel that
 * owns the chunk's first row (one kernel per dispatch), 0 = by a head kernel behind it
 * (a thread per chunk for the default designs, a wavefront per chunk for designs with more
 * than four sections per stage), 2 = as 0 but thread-per-chunk kernels only, -1 = automatic
 * by chunk size (default).  Results are bit-identical. */
int demod_set_head_mode(demod_ctx *ctx, int mode);

/* Tuning knob: rows per dispatch of the fused kernel (a call covering more is
 * cut into that many back-to-back dispatches); power of two, 0 = default
 * (2^21 rows = 2 GiB of input). */
int demod_set_piece_rows(demod_ctx *ctx, uint32_t rows);

/* Counter-based synthetic IQ bytes written straight into device memory
 * (benchmark input): byte i of stream `seed` is byte
 * (i & 7) of splitmix64 word (i >> 3).  first_byte and n_bytes multiples of 16. */
int demod_synth_fill(demod_ctx *ctx, void *d_dst, uint64_t first_byte, uint64_t n_bytes,
                     uint64_t seed, void *stream);

/* Device self-tests for the GPU test-suite (not on the data path): the
 * kernel's atan2 on n (y, x) pairs from host arrays, and the wavefront
 * neighbour exchange (out[i] = in[i + 1], out[63] = halo). */
int demod_selftest_atan2(demod_ctx *ctx, const float *h_y, const float *h_x, float *h_out,
                         uint32_t n);
int demod_selftest_lane_shift(demod_ctx *ctx, const float *h_in64, float halo, float *h_out64);
/* -q1 only (test-suite): the running offsets the most recent process call replayed -- n_pairs (re, im) pairs, one
 * per iteration of correctIq in stream order (iteration t of chunk c at pair c * chunk_bytes / 4 + t: the offsets
 * complex samples t and K-1-t of that chunk were corrected with, src/matrix.c:131-135) -- and the state the call
 * left for the next one (h_state2, two floats).  Waits for the device.  Either pointer may be null. */
int demod_selftest_q1_offsets(demod_ctx *ctx, float *h_pairs, uint64_t n_pairs, float *h_state2);

/* Average time in milliseconds the fused kernel took per process call, over the
 * calls issued since the previous call to this function, measured with HIP
 * events on the stream the kernel was launched on (a call that covers more than
 * 2 GiB of input dispatches the kernel once per 2 GiB piece, back to back; the
 * events bracket all of them).  *calls = number of calls averaged.  Enables
 * event recording on first use; returns < 0 if nothing was recorded.
 * demod_last_call_launches: dispatches of the fused kernel in the most recent call. */
double demod_kernel_ms(demod_ctx *ctx, int *calls);
int demod_last_call_launches(const demod_ctx *ctx);
void demod_enable_timing(demod_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* DEMOD_HIP_H */
