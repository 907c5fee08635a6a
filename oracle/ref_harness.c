/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thread-free harness around the REAL reference sources.  It pulls in
 * /root/reference/src/matrix.c by path (to reach its static functions) and is
 * linked with /root/reference/src/filter.c where they lie; nothing from the
 * reference is copied into this repository.  Built only where /root/reference
 * exists (this container) into oracle/_ref/libdemod_ref.so, which is
 * git-ignored and travels to the GPU box as a prebuilt file.
 *
 * Used to (a) pin oracle/demod_oracle.c bit-for-bit and (b) generate
 * tests/golden/ (tests/golden/make_golden.py).
 */
#include "/root/reference/src/matrix.c"

#include "demod_oracle.h"

void ref_design_sos(unsigned kind, size_t degree, long double fc, long double fs,
                    long double epsilon, float (*sos)[6]) {
    processFilterOption((uint8_t) kind, degree, sos, fc, fs, epsilon);
}

void ref_unpack_shift(const uint8_t *bytes, size_t len, float *out) {
    shiftOrigin((void *) bytes, len, out);
}

void ref_unpack_normalize(const uint8_t *bytes, size_t len, float *out) {
    normalizeInput((void *) bytes, len, out);
}

void ref_discriminate(const float *in, size_t len, float *out) {
    fmDemod(in, len, out);
}

void ref_filter_real(float *x, float *y, size_t len, size_t sections, const float (*sos)[6]) {
    applyFilter(x, y, len, sections, sos);
}

void ref_filter_complex(float *x, float *y, size_t len, size_t sections,
                        const float (*sos)[6]) {
    applyComplexFilter(x, y, len, sections, sos);
}

/* Replays the consumer loop of processMatrix (src/matrix.c:181-273) without the
 * semaphore hand-off: same allocations, same buffer offsets, same call
 * sequence, the reference's own functions.  Full chunks only (the reference's
 * tail handling is racy).  Returns floats written or
 * -1 for option sets the harness does not drive. */
/* -q1 and -q2 in THIS function are only meaningful in a fresh process (-q1: correctIq's
 * static offsets start at zero once per process, src/matrix.c:125).  -q2: highpassDc keeps
 * a static buffer sized by its first call and static coefficients
 * (src/matrix.c:147-153), and what its filter reads behind that buffer depends
 * on the allocator's state.  oracle/__init__.py therefore runs -q2 streams
 * through oracle/_ref/ref_stream (ref_stream_main.c), one process per stream. */
long long ref_process_stream(const orc_params *p, const uint8_t *in, size_t n_bytes,
                             float *out) {
    const size_t B = p->buf_size;
    const size_t chunks = B ? n_bytes / B : 0;
    const size_t outputLen = B >> 2;
    const unsigned demodMode = (p->mode >> 4) & 3;
    const unsigned iqMode = (p->mode >> 2) & 3;
    size_t inDegree = p->lowpass_in != 0 && !p->in_degree ? p->out_degree : p->in_degree;
    const size_t sosLenOut = (p->out_degree >> 1) + (p->out_degree & 1);
    const size_t sosLenIn = (inDegree >> 1) + (inDegree & 1);
    size_t workLen = B << 1;
    iqCorrection_t unpack;
    float *work, *dem;
    size_t c;

    if (!demodMode) return -1;
    if (inDegree && p->lowpass_in == 0) return -1;
    unpack = iqMode == 3 ? normalizeInput : iqMode == 2 ? highpassDc : iqMode == 1 ? correctIq : shiftOrigin;
    samplingRate = p->sample_rate;  /* src/matrix.c:206 */
    esr = (REAL) (50. / p->sample_rate); /* src/matrix.c:210 (correctIq keeps static offsets: one process per stream) */

    {
        REAL sosIn[sosLenIn ? sosLenIn : 1][6];
        REAL sosOut[sosLenOut][6];

        processFilterOption(p->mode & 1, p->out_degree, sosOut, p->lowpass_out,
                            p->sample_rate, p->epsilon);
        if (p->lowpass_in != 0) {
            workLen <<= 1;
            processFilterOption((p->mode >> 1) & 1, inDegree, sosIn, p->lowpass_in,
                                p->sample_rate, p->epsilon);
        }
        work = calloc(workLen, sizeof(REAL));
        dem = calloc(B, sizeof(REAL));

        for (c = 0; c < chunks; ++c) {
            unpack((void *) (in + c * B), B, work);
            if (!inDegree) {
                fmDemod(work, B, dem);
                applyFilter(dem, work + B, outputLen, sosLenOut, sosOut);
                memcpy(out + c * outputLen, work + B, outputLen * sizeof(REAL));
            } else {
                applyComplexFilter(work, work + B, B, sosLenIn, sosIn);
                fmDemod(work + B, B, dem);
                applyFilter(dem, work + (B << 1), outputLen, sosLenOut, sosOut);
                memcpy(out + c * outputLen, work + (B << 1), outputLen * sizeof(REAL));
            }
            memset(work, 0, workLen * sizeof(REAL));
        }
        free(dem);
        free(work);
    }
    return (long long) (chunks * outputLen);
}
