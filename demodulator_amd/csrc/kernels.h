// kernels.h -- launch descriptors shared by the C-ABI layer (api.cpp) and the
// HIP kernels (kernels.hip).  Plain structs passed to kernels by value.
#pragma once

#include <cstdint>

#include "plan.h"

namespace demod {

constexpr uint32_t kRowBytes = 1024;   // one wavefront row: 64 lanes x 16 B of IQ
constexpr uint32_t kRowOutputs = 256;  // float32 results per row: 64 lanes x 4

// Largest segment (rows per wavefront) the streaming kernel was built for.
uint32_t max_segment_rows();
// Sections per stage the streaming kernel's own chunk-head replay handles.
constexpr int kFusedHeadSections = 4;

// Main streaming kernel: rows [0, n_rows) of a device buffer.
struct RowLaunch {
    const uint8_t *in;
    void *out;             // float32, or int16 when out_s16
    const uint8_t *halo;   // DEMOD_HALO_BYTES that follow the buffer in the stream (device), or null
    uint32_t n_rows;       // rows this launch computes; the last one may be ragged
    uint32_t rows_readable; // rows that exist at `in` (>= n_rows): a launch may look ahead into the row after its last
    uint32_t phase_rows;   // chunk-local row index of buffer row 0
    uint32_t rpc_log2;     // log2(rows per chunk)
    uint32_t seg_log2;     // log2(rows per wavefront segment); a segment may span chunks shorter than it
    uint32_t head_quads;   // lanes of a chunk's row 0 whose outputs are chunk-head outputs (ceil(H / 4))
    // chunk head replayed by the wavefront that owns the chunk's row 0 (kernels.hip chunk_head); 0: left to demod_heads
    uint32_t fused_heads;
    // chunks shorter than a segment: the heads of a segment's chunks side by side, sixteen lanes each, in one pass
    // (a head's reach 4 head_quads + sec_in + sec_out + 2 has to fit sixteen lanes); 0: one pass per chunk
    uint32_t heads_packed;
    int32_t sec_in, sec_out;  // sections of the complex stage (0 = none) and of the output stage
    int32_t in_separate;      // Plan::in_separate
    float sos_in[4][6], sos_out[4][6];
    // -q1 (kCorrectIq, bounds-checked build only): the running offsets of every iteration, one
    // (re, im) pair per iteration, chunk c's at q1_table[c * q1_stride ..]; chunk 0 = the chunk
    // that holds buffer row 0.  The stream's last chunk may be short: q1_last_chunk / q1_last_k.
    const float *q1_table;
    uint32_t q1_stride;      // entries (pairs) per chunk = chunk bytes / 4
    uint32_t q1_last_chunk;  // index of the short last chunk, or 0xFFFFFFFF
    uint32_t q1_last_k;      // its complex samples
    uint32_t ragged_bytes; // valid bytes of the last row (multiple of 4, < 1024), 0 = full
    float hin[kMaxTaps];
    float bias_im;
    float fold_re, fold_im;  // constant terms of the complex stencil with the unpack's -128 folded in
    // kShiftOriginNorm: taps over the first one; the unpack alpha * (sample + 128) + (nk_re, nk_im), nk = -(128 alpha,
    // alpha (128 - s)); a sample past a chunk end (0.0f) is (0, past_im) there
    float rho[kMaxTaps];
    float alpha;
    float nk_re, nk_im;
    float past_im;
    float hout[kMaxTaps];
    float scale;           // normalizeInput: (float)(2./255.)
    uint32_t out_s16;      // results leave as int16: clamp(y * out_scale, -1, 1) * 32767 rounded (host side: picks the build)
    float out_scale;
};

// Chunk-head kernel: literal replay of the sectioned recurrences over the first
// few samples of every chunk that starts inside the buffer.
struct HeadLaunch {
    const uint8_t *in;         // first byte of the first chunk head
    void *out;                 // its first output (float32 or int16, see RowLaunch)
    int out_s16;
    float out_scale;
    uint64_t chunk_bytes;      // distance between heads
    uint32_t n_heads;
    uint32_t head_outputs;     // outputs to produce per head (multiple of 4)
    uint32_t last_chunk_bytes; // bytes of the last head's chunk (== chunk_bytes unless ragged tail)
    int sec_in, sec_out;
    int in_separate;           // Plan::in_separate
    int sec_pre;               // 2 when highpassDc runs in front of the -L stage (Plan::pre), else 0
    float sos_pre[2][6];
    float sos_in[kMaxSections][6];
    float sos_out[kMaxSections][6];
    float scale;
    const float *q1_table;     // -q1: offsets of head 0's chunk onwards (see RowLaunch), q1_stride pairs per chunk
    uint32_t q1_stride;
    int thread_per_head;       // keep to the thread-per-chunk kernels (demod_set_head_mode 2: what the tests compare demod_heads_wave with)
};

// Whole-chunk literal kernel for chunks smaller than one row (B < 1024).
struct TinyLaunch {
    const uint8_t *in;
    void *out;
    int out_s16;
    float out_scale;
    float *work;               // per-thread scratch, work_stride floats each
    uint64_t work_stride;
    uint64_t chunk_bytes;
    uint64_t n_chunks;         // full chunks
    uint32_t tail_bytes;       // trailing short chunk (multiple of 4), 0 = none
    int sec_in, sec_out;
    int in_separate;           // Plan::in_separate
    int sec_pre;               // Plan::pre
    float sos_pre[2][6];
    float sos_in[kMaxSections][6];
    float sos_out[kMaxSections][6];
    float scale;
    const float *q1_table;     // -q1: offsets of chunk 0 onwards, q1_stride pairs per chunk
    uint32_t q1_stride;
};

// -q1: replay of correctIq's offset recurrence (src/matrix.c:131-138), a fixed budget of threads.
struct OffsetLaunch {
    const uint8_t *in;         // first chunk
    float *table;              // out: (re, im) offsets BEFORE every iteration, stride pairs per chunk
    const float *carry_in;     // the two offsets the first chunk starts from
    float *carry_out;          // ... and what the last chunk leaves
    uint64_t chunk_bytes;
    uint64_t n_chunks;         // full chunks
    uint32_t tail_bytes;       // trailing short chunk (multiple of 4), 0 = none
    uint32_t stride;           // table pairs per chunk = chunk_bytes / 4
    uint32_t warm;             // iterations of the first run-in before a piece (Plan::q1_warm) ...
    uint32_t retry;            // ... and how much further back the second one starts (Plan::q1_retry); then twice as far each time
    uint32_t parts;            // threads that share one chunk (>= 1) ...
    uint32_t group;            // ... or chunks one thread walks through (>= 1; one of the two is 1)
    float esr;
    float widen;               // the bracket's widening per iteration during its first span (>= 2 J, kernels.hip) ...
    float floor2;              // ... and twice the offset above which the rounded step is monotone (the lower copy's floor)
};

// Host-callable launchers (defined in kernels.hip).  `stream` is a hipStream_t.
int launch_rows(const RowLaunch &a, int taps_in, int taps_out, int conditioning, void *stream);
int launch_heads(const HeadLaunch &a, int conditioning, void *stream);
int launch_tiny(const TinyLaunch &a, int conditioning, uint32_t threads, void *stream);
int launch_offsets(const OffsetLaunch &a, void *stream);
int launch_synth_fill(void *dst, uint64_t first_byte, uint64_t n_bytes, uint64_t seed, void *stream);
// Device self-test helper: atan2 of n (y, x) pairs with the kernel's approximation.
int launch_atan2_probe(const float *y, const float *x, float *out, uint32_t n, void *stream);
// Device self-test helper: out[i] = value of lane i+1 (lane 63: halo) -- checks the DPP shift direction.
int launch_lane_probe(const float *in, float *out, float halo, void *stream);

}  // namespace demod
