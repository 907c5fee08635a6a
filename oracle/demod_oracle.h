/*
 * oracle/demod_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference FM-demodulation hot path
 * (/root/reference/src/matrix.c + src/filter.c), written from the algorithm,
 * used only as the checker in tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing under demodulator_amd/ may include,
 * link or call this.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the
 * reference's own functions compiled from /root/reference (oracle/_ref, see
 * oracle/Makefile) and against the golden vectors committed in tests/golden/
 * which were produced by that compiled reference (tests/golden/make_golden.py).
 */
#ifndef DEMOD_ORACLE_H
#define DEMOD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Run-time options of the path; same meaning as the fields of the reference's
 * consumerArgs (include/matrix.h:43-57) that processMatrix reads. */
typedef struct {
    long double sample_rate;   /* -S, default 125000 (src/main.c:105) */
    long double lowpass_in;    /* -L, 0 = no input filter (src/main.c:106) */
    long double lowpass_out;   /* -l, default 12500 (src/main.c:107) */
    long double epsilon;       /* -e x stores x/10, default .3 (src/main.c:110,158-160) */
    size_t in_degree;          /* -D, default 0 (src/main.c:108) */
    size_t out_degree;         /* -d, default 3 (src/main.c:109) */
    unsigned mode;             /* ww|dd|qq|ff byte, default 0x10 (src/main.c:112) */
    size_t buf_size;           /* chunk bytes B, default 262144 (include/matrix.h:37-39) */
} orc_params;

void orc_default_params(orc_params *p);

/* Number of second-order sections for a degree: ceil(degree/2)
 * (src/matrix.c:32-33,195-200). */
size_t orc_sections(size_t degree);

/* Filter design -> float SOS rows {b0,b1,b2,a0,a1,a2}.
 * Restates processFilterOption (src/matrix.c:25-80) with transformBilinear
 * (src/filter.c:142-210), the warpButter family (src/filter.c:22-58), the
 * warpCheby1 family (src/filter.c:60-101) and zp2Sos
 * (src/filter.c:104-140).  kind: 0 Butterworth LP, 1 Chebyshev-I LP,
 * 2 Butterworth HP, 3 Chebyshev-I HP. */
void orc_design_sos(unsigned kind, size_t degree, long double fc, long double fs,
                    long double epsilon, float (*sos)[6]);

/* uint8 IQ -> float, (int8_t)(byte-127); shiftOrigin, src/matrix.c:82-98. */
void orc_unpack_shift(const uint8_t *bytes, size_t len, float *out);
/* uint8 IQ -> float, byte*(2/255)-1; normalizeInput, src/matrix.c:100-118. */
void orc_unpack_normalize(const uint8_t *bytes, size_t len, float *out);

/* uint8 IQ -> float with running DC offsets, correctIq (-q1), src/matrix.c:120-140;
 * esr = (float)(50 / sample_rate) (src/matrix.c:210); off[2] is the reference's static
 * state, carried from call to call. */
void orc_unpack_correct_iq(const uint8_t *bytes, size_t len, float *out, float esr, float off[2]);

/* Quadrature discriminator, fmDemod, src/matrix.c:159-176. len = floats in. */
void orc_discriminate(const float *in, size_t len, float *out);

/* Sectioned in-place recurrences, src/filter.c:212-231 and :233-259.
 * y must be pre-zeroed (or hold what the reference's buffer holds). */
void orc_filter_real(const float *x, float *y, size_t len, size_t sections,
                     const float (*sos)[6]);
void orc_filter_complex(const float *x, float *y, size_t len, size_t sections,
                        const float (*sos)[6]);

/* Whole-stream replay of the processMatrix chunk loop (src/matrix.c:236-273),
 * single-threaded, no semaphores: every full chunk of p->buf_size bytes of
 * `in` yields buf_size/4 floats in `out`.  A trailing partial chunk of r bytes
 * is replayed as a short chunk of (r & ~3) bytes yielding r/4 floats (the
 * build's deterministic tail policy; the reference's
 * own tail is racy).  Returns the number of floats written, or -1 on an
 * unsupported option set (-D without -L, demod mode 0). */
long long orc_process_stream(const orc_params *p, const uint8_t *in, size_t n_bytes,
                             float *out);

/* Same, chunk-parallel over `threads` worker threads (chunks are independent,
 * src/matrix.c:272).  threads <= 1 runs the serial loop. */
long long orc_process_stream_mt(const orc_params *p, const uint8_t *in, size_t n_bytes,
                                float *out, int threads);

/* Counter-based synthetic IQ bytes: byte i of the
 * stream with seed s is byte (i&7) of splitmix64 word (i>>3). */
void orc_synth_bytes(uint64_t seed, uint64_t first_byte, size_t n, uint8_t *dst);

#ifdef __cplusplus
}
#endif
#endif
