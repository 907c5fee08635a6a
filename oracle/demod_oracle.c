/*
 * oracle/demod_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference hot path; see demod_oracle.h for scope and
 * parity status (pinned against oracle/_ref and tests/golden/).
 *
 * Floating-point order matters: every expression below keeps the operand
 * association of the reference expression it cites so that, compiled without
 * FMA contraction (-ffp-contract=off, as the reference's generic x86-64 build
 * behaves), results are bit-identical to the reference's functions.
 */
#define _GNU_SOURCE
#include "demod_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* options                                                                   */
/* ------------------------------------------------------------------------ */

void orc_default_params(orc_params *p) {
    /* src/main.c:102-113 */
    p->sample_rate = 125000.L;
    p->lowpass_in = 0.L;
    p->lowpass_out = 12500.L;
    p->epsilon = .3; /* the reference stores the double literal .3 */
    p->in_degree = 0;
    p->out_degree = 3;
    p->mode = 0x10;
    p->buf_size = 262144;
}

size_t orc_sections(size_t degree) {
    return (degree >> 1) + (degree & 1);
}

/* ------------------------------------------------------------------------ */
/* filter design (long double), src/filter.c:22-210 + src/matrix.c:25-80     */
/* ------------------------------------------------------------------------ */

typedef struct {
    long double re;   /* real part of the z-plane pole, 1 - zr */
    long double im;   /* imaginary part, +zj (its conjugate is implied) */
    long double zr;   /* value the warp function returns to the gain loop */
} pole_t;

/* warpButterGeneric, src/filter.c:22-40; highpass variant returns 2 - zr
 * (src/filter.c:42-49). */
static pole_t butter_pole(long double alpha, long double beta, size_t k, size_t n,
                          int highpass) {
    const long double inv_n = 1. / (long double) n;
    const long double odd = -1. + (long double) (k << 1);
    const long double theta = M_PI_2 * (inv_n * odd + 1.);
    const long double c = cosl(theta);
    const long double d = 1. / (c - alpha);
    const long double zr = (-beta + c) * d;
    const long double zj = sinl(theta) * d;
    pole_t p;
    p.re = -zr + 1.;
    p.im = zj;
    p.zr = highpass ? 2. - zr : zr;
    return p;
}

/* warpCheby1Generic, src/filter.c:60-83; the highpass variant passes 1/tng
 * (src/filter.c:94-101). */
static pole_t cheby1_pole(long double tng, long double ep, size_t k, size_t n) {
    const long double inv_n = 1. / (long double) n;
    const long double v =
        logl((1. + powl(10., 0.5 * ep)) / sqrtl(-1. + powl(10., ep))) * inv_n;
    const long double t = M_PI_2 * (inv_n * (-1. + (long double) (k << 1)));
    const long double a = cosl(t) * coshl(v) * tng;
    const long double b = sinl(t) * sinhl(v) * tng;
    const long double c = a * a + b * b;
    const long double d = 1. / (1. + c + 2. * b);
    const long double zj = 2. * a * d;
    const long double zr = 2. * (b + c) * d;
    pole_t p;
    p.re = -zr + 1.;
    p.im = zj;
    p.zr = zr;
    return p;
}

void orc_design_sos(unsigned kind, size_t degree, long double fc, long double fs,
                    long double epsilon, float (*sosf)[6]) {
    const size_t n = degree;
    const size_t rows = orc_sections(n);
    const size_t pairs = n >> 1;
    const int odd = (int) (n & 1);
    const int highpass = (kind == 2 || kind == 3);
    /* src/matrix.c:34,37 */
    const long double w = M_PI * fc / fs;
    const long double wh =
        coshl(1. / (long double) n * acoshl(1. / sqrtl(powl(10., epsilon) - 1.)));
    long double alpha, beta;
    pole_t *poles;
    long double(*sos)[6];
    long double acc_re, acc_im, gain;
    size_t k, j;

    if (!n) return;
    poles = calloc(rows, sizeof *poles);
    sos = calloc(rows, sizeof *sos);

    /* argument pairs of the four transformBilinear calls, src/matrix.c:48-73 */
    switch (kind) {
    case 1: alpha = tanl(w * wh); beta = epsilon; break;
    case 2: alpha = 1. / sinl(2. * w); beta = tanl(w); break;
    case 3: alpha = tanl(w * wh); beta = epsilon; break;
    default: alpha = 1. / sinl(2. * w); beta = tanl(w); break;
    }

    /* pole generation and gain accumulation, src/filter.c:153,165-184:
     * conjugate pairs multiply the gain by |zr + i zj|^2, an unpaired (real)
     * pole multiplies it as a complex number; even-order lowpass Chebyshev
     * starts from 1/sqrt(2). */
    acc_re = (!odd && kind == 1) ? M_SQRT1_2 : 1.;
    acc_im = 0;
    for (k = 1; k <= rows; ++k) {
        pole_t p;
        long double a;
        switch (kind) {
        case 1: p = cheby1_pole(alpha, beta, k, n); break;
        case 3: p = cheby1_pole(1. / alpha, beta, k, n); break;
        case 2: p = butter_pole(alpha, beta, k, n, 1); break;
        default: p = butter_pole(alpha, beta, k, n, 0); break;
        }
        poles[k - 1] = p;
        if (k <= pairs) {
            a = p.zr * p.zr + p.im * p.im;
            acc_re *= a;
            acc_im *= a;
        } else {
            a = p.zr * acc_re - p.im * acc_im;
            acc_im = p.zr * acc_im + p.im * acc_re;
            acc_re = a;
        }
    }
    gain = acc_re / (long double) (1 << n);

    /* zp2Sos, src/filter.c:104-140: all n zeros sit at z = -1 (lowpass) or
     * z = +1 (highpass), src/filter.c:159,186-189. */
    {
        const long double zero = highpass ? 1 : -1;
        for (j = 0; j < pairs; ++j) {
            sos[j][0] = 1.;
            sos[j][1] = -2. * zero;
            sos[j][2] = zero * zero + 0.L * 0.L;
            sos[j][3] = 1.;
            sos[j][4] = -2. * poles[j].re;
            sos[j][5] = poles[j].re * poles[j].re + poles[j].im * poles[j].im;
        }
        if (odd) {
            sos[pairs][0] = 1.;
            sos[pairs][1] = -zero;
            sos[pairs][2] = 0.;
            sos[pairs][3] = 1.;
            sos[pairs][4] = -poles[pairs].re;
            sos[pairs][5] = 0.;
        } else {
            sos[0][0] = 1.;
            sos[0][2] = 1.;
            sos[0][1] = -2. * zero;
        }
        /* the gain scales the numerator of the LAST row, src/filter.c:113,137-139 */
        j = odd ? pairs : pairs - 1;
        sos[j][0] *= gain;
        sos[j][1] *= gain;
        sos[j][2] *= gain;
    }

    for (k = 0; k < rows; ++k)
        for (j = 0; j < 6; ++j) sosf[k][j] = (float) sos[k][j]; /* src/matrix.c:75-79 */

    free(sos);
    free(poles);
}

/* ------------------------------------------------------------------------ */
/* per-chunk primitives                                                      */
/* ------------------------------------------------------------------------ */

void orc_unpack_shift(const uint8_t *bytes, size_t len, float *out) {
    /* src/matrix.c:92-96: every byte i maps to (int8_t)(byte - 127); the
     * reference's two-ended loop covers each index once when len % 4 == 0. */
    size_t i;
    for (i = 0; i < len; ++i) out[i] = (float) (int8_t) (bytes[i] - 127);
}

void orc_unpack_normalize(const uint8_t *bytes, size_t len, float *out) {
    /* src/matrix.c:106,112: float product, then "- 1." in double, stored as float */
    const float denom = 2. / 255.;
    size_t i;
    for (i = 0; i < len; ++i) out[i] = (float) bytes[i] * denom - 1.;
}

void orc_unpack_correct_iq(const uint8_t *bytes, size_t len, float *out, float esr, float off[2]) {
    /* correctIq, src/matrix.c:120-140: a two-ended loop -- iteration i/2 takes the complex
     * sample at bytes i, i+1 and its mirror image at len-i-2, len-i-1, subtracts the running
     * offsets from both and moves the offsets by (sum of the two corrected values) * esr.
     * `off` is the reference's static off[2]: it carries over from chunk to chunk. */
    const size_t N = len >> 1;
    size_t i;
    for (i = 0; i < N; i += 2) {
        out[i] = ((float) bytes[i]) - off[0];
        out[len - i - 2] = ((float) bytes[len - i - 2]) - off[0];

        out[i + 1] = ((float) bytes[i + 1]) - off[1];
        out[len - i - 1] = ((float) bytes[len - i - 1]) - off[1];

        off[0] += (out[i] + out[len - i - 2]) * esr;
        off[1] += (out[i + 1] + out[len - i - 1]) * esr;
    }
}

void orc_discriminate(const float *in, size_t len, float *out) {
    /* src/matrix.c:166-175 */
    size_t i;
    for (i = 0; i + 3 < len; i += 4) {
        const float a = in[i], b = in[i + 1], c = in[i + 2], d = in[i + 3];
        const float zr = a * c + b * d;
        const float zj = -a * d + b * c;
        const float ph = atan2f(zj, zr);
        out[i >> 2] = isnan(ph) ? 0.f : ph;
    }
}

void orc_filter_real(const float *x, float *y, size_t len, size_t sections,
                     const float (*sos)[6]) {
    /* src/filter.c:221-230 */
    size_t i, m;
    for (i = 0; i < len; ++i) {
        const float *xs = x + i + sections;
        float *ys = y + i + sections;
        for (m = 0; m < sections; ++m) {
            ys[m] = sos[m][0] * ys[m] + sos[m][1] * ys[m + 1] + 1;
            ys[m] -= sos[m][3] + sos[m][4] * xs[m] + sos[m][5] * xs[m + 1];
        }
    }
}

void orc_filter_complex(const float *x, float *y, size_t len, size_t sections,
                        const float (*sos)[6]) {
    /* src/filter.c:242-258: same recurrence on interleaved re/im with stride
     * 2; the "+ 1" exists on the real lane only. */
    size_t i, m;
    for (i = 0; i < len; i += 2) {
        const float *xs = x + i + 2 * sections;
        float *ys = y + i + 2 * sections;
        for (m = 0; m < sections; ++m) {
            const size_t l = 2 * m;
            ys[l] = sos[m][0] * ys[l] + sos[m][1] * ys[l + 2] + 1;
            ys[l] -= sos[m][3] + sos[m][4] * xs[l] + sos[m][5] * xs[l + 2];
            ys[l + 1] = sos[m][0] * ys[l + 1] + sos[m][1] * ys[l + 3];
            ys[l + 1] -= sos[m][3] + sos[m][4] * xs[l + 1] + sos[m][5] * xs[l + 3];
        }
    }
}

/* ------------------------------------------------------------------------ */
/* chunk loop, src/matrix.c:178-280                                          */
/* ------------------------------------------------------------------------ */

typedef struct {
    size_t chunk;           /* B */
    size_t sec_in, sec_out; /* SOS rows */
    int prefilter;          /* complex input filter enabled */
    int normalize;          /* -q3 unpack instead of shiftOrigin */
    int highpass;           /* -q2: highpassDc in front of everything else */
    int correct;            /* -q1: correctIq, offsets carried from chunk to chunk */
    float esr;              /* src/matrix.c:210 */
    float (*sos_in)[6];
    float (*sos_out)[6];
    float sos_hp[2][6];     /* src/matrix.c:147,151 */
} plan_t;

static int make_plan(const orc_params *p, plan_t *pl) {
    size_t in_degree = p->in_degree;
    const unsigned demod_mode = (p->mode >> 4) & 3;
    const unsigned iq_mode = (p->mode >> 2) & 3;

    memset(pl, 0, sizeof *pl);
    if (!demod_mode) return -1;            /* src/matrix.c:244-251, unreachable from the CLI */
    if (!p->out_degree || !p->buf_size || (p->buf_size & 3)) return -1;
    /* src/matrix.c:190-192 */
    if (p->lowpass_in != 0 && !in_degree) in_degree = p->out_degree;
    /* -D without -L reads an uninitialised sosIn in the reference
     * (src/matrix.c:227-231,254); not a defined behaviour to restate. */
    if (in_degree && p->lowpass_in == 0) return -1;

    pl->chunk = p->buf_size;
    pl->normalize = (iq_mode == 3);
    pl->highpass = (iq_mode == 2);
    pl->correct = (iq_mode == 1);
    pl->esr = (float) (50. / p->sample_rate); /* src/matrix.c:210 */
    /* src/matrix.c:151: Butterworth highpass, degree 3, 1 Hz, epsilon 0 */
    if (pl->highpass) orc_design_sos(2, 3, 1.L, p->sample_rate, 0.L, pl->sos_hp);
    pl->prefilter = in_degree != 0;
    pl->sec_out = orc_sections(p->out_degree);
    pl->sec_in = orc_sections(in_degree);
    pl->sos_out = calloc(pl->sec_out, sizeof *pl->sos_out);
    orc_design_sos(p->mode & 1, p->out_degree, p->lowpass_out, p->sample_rate, p->epsilon,
                   pl->sos_out); /* src/matrix.c:224-225 */
    if (pl->prefilter) {
        pl->sos_in = calloc(pl->sec_in, sizeof *pl->sos_in);
        orc_design_sos((p->mode >> 1) & 1, in_degree, p->lowpass_in, p->sample_rate,
                       p->epsilon, pl->sos_in); /* src/matrix.c:229-230 */
    }
    return 0;
}

static void free_plan(plan_t *pl) {
    free(pl->sos_in);
    free(pl->sos_out);
}

/* highpassDc's private buffer is B floats (src/matrix.c:152) and its filter
 * reads up to 4S+3 = 11 floats past it.  In the reference that is the rest of
 * the calloc'ed mapping (zeros for the chunk sizes the CLI can reach with mmap'ed
 * allocations, >= 32 KiB chunks; heap contents otherwise -- undefined there,
 * zeros here). */
#define HP_PAD 16

typedef struct {
    float *work;   /* filterRet: 2B floats, or 4B with the prefilter (src/matrix.c:186-187,228,234) */
    float *demod;  /* demodRet: B floats, only [0, B/4) ever written (src/matrix.c:184) */
    float *hp;     /* highpassDc's static buf, -q2 only */
    size_t work_len;
} scratch_t;

/* With chunks of a few bytes and many sections the recurrences reach past these buffers
 * (a stage with S sections touches 2S+1 cells -- 4S+3 floats for the complex one -- behind its
 * last output): heap contents in the reference, i.e. undefined there.  The checker must not
 * corrupt its own heap over it: SLACK zeroed floats behind each buffer (never part of work_len,
 * so the per-chunk memset matches the reference's).  Parity is only defined where the reach
 * stays inside the reference's own allocation; the tests keep to that. */
#define SLACK (4 * 16 + 8)

static int make_scratch(const plan_t *pl, scratch_t *s) {
    s->work_len = (pl->chunk << 1) << (pl->prefilter ? 1 : 0);
    s->work = calloc(s->work_len + SLACK, sizeof(float));
    s->demod = calloc(pl->chunk + SLACK, sizeof(float));
    s->hp = pl->highpass ? calloc(pl->chunk + HP_PAD, sizeof(float)) : NULL;
    return (s->work && s->demod && (s->hp || !pl->highpass)) ? 0 : -1;
}

static void free_scratch(scratch_t *s) {
    free(s->work);
    free(s->demod);
    free(s->hp);
}

/* One iteration of the consumer loop, src/matrix.c:253-272, on `len` bytes
 * (len == B for full chunks; shorter only for the build's tail policy, where
 * the buffers keep their full-chunk size and offsets as the reference's do). */
static void run_chunk(const plan_t *pl, scratch_t *s, const uint8_t *bytes, size_t len,
                      float *out, float off[2]) {
    float *F = s->work;
    const size_t B = len;
    const size_t n_out = B >> 2;

    if (pl->highpass) {
        /* highpassDc, src/matrix.c:142-157: shiftOrigin into its own buffer, then
         * the complex filter from there into filterRet; the 4S-2 floats the
         * filter writes past its B outputs land in filterRet + B, the next
         * stage's buffer.  (A short tail chunk, this build's tail policy, sees
         * zeros behind its last sample, not the previous chunk's samples.) */
        orc_unpack_shift(bytes, B, s->hp);
        memset(s->hp + B, 0, (pl->chunk + HP_PAD - B) * sizeof(float));
        orc_filter_complex(s->hp, F, B, 2, (const float(*)[6]) pl->sos_hp);
    } else if (pl->correct)
        orc_unpack_correct_iq(bytes, B, F, pl->esr, off);
    else if (pl->normalize)
        orc_unpack_normalize(bytes, B, F);
    else
        orc_unpack_shift(bytes, B, F);

    if (!pl->prefilter) {
        orc_discriminate(F, B, s->demod);
        orc_filter_real(s->demod, F + B, n_out, pl->sec_out, (const float(*)[6]) pl->sos_out);
        memcpy(out, F + B, n_out * sizeof(float));
    } else {
        orc_filter_complex(F, F + B, B, pl->sec_in, (const float(*)[6]) pl->sos_in);
        orc_discriminate(F + B, B, s->demod);
        orc_filter_real(s->demod, F + 2 * B, n_out, pl->sec_out,
                        (const float(*)[6]) pl->sos_out);
        memcpy(out, F + 2 * B, n_out * sizeof(float));
    }
    memset(F, 0, s->work_len * sizeof(float)); /* src/matrix.c:272 */
    /* demodRet is never cleared by the reference; a short tail chunk leaves
     * stale values beyond n_out that a later chunk would read -- only the LAST
     * chunk may be short, so nothing reads them. */
}

long long orc_process_stream(const orc_params *p, const uint8_t *in, size_t n_bytes,
                             float *out) {
    return orc_process_stream_mt(p, in, n_bytes, out, 1);
}

typedef struct {
    const plan_t *pl;
    const uint8_t *in;
    float *out;
    size_t first, last; /* chunk range [first, last) */
    float off[2];       /* -q1: the offsets chunk `first` starts from */
} job_t;

static void *worker(void *arg) {
    job_t *j = arg;
    scratch_t s;
    size_t c;
    if (make_scratch(j->pl, &s)) return (void *) 1;
    for (c = j->first; c < j->last; ++c)
        run_chunk(j->pl, &s, j->in + c * j->pl->chunk, j->pl->chunk,
                  j->out + c * (j->pl->chunk >> 2), j->off);
    free_scratch(&s);
    return NULL;
}

/* -q1: the offsets after the first `chunks` chunks (the recurrence of correctIq alone,
 * src/matrix.c:131-138, without storing the corrected samples) -- what lets the
 * chunk-parallel form start every worker from the state the serial loop would hand it. */
static void correct_iq_state(const plan_t *pl, const uint8_t *in, size_t first, size_t last, float off[2]) {
    const size_t len = pl->chunk, N = len >> 1;
    size_t c, i;
    for (c = first; c < last; ++c) {
        const uint8_t *b = in + c * len;
        for (i = 0; i < N; i += 2) {
            const float f0 = ((float) b[i]) - off[0], m0 = ((float) b[len - i - 2]) - off[0];
            const float f1 = ((float) b[i + 1]) - off[1], m1 = ((float) b[len - i - 1]) - off[1];
            off[0] += (f0 + m0) * pl->esr;
            off[1] += (f1 + m1) * pl->esr;
        }
    }
}

long long orc_process_stream_mt(const orc_params *p, const uint8_t *in, size_t n_bytes,
                                float *out, int threads) {
    plan_t pl;
    size_t full, tail, t;
    long long written;

    if (make_plan(p, &pl)) return -1;
    full = n_bytes / pl.chunk;
    tail = (n_bytes - full * pl.chunk) & ~(size_t) 3;
    if (threads < 1) threads = 1;
    if ((size_t) threads > full) threads = full ? (int) full : 1;

    float off[2] = {0.f, 0.f}; /* static REAL off[2] = {}, src/matrix.c:125 */
    if (threads == 1) {
        job_t j = {&pl, in, out, 0, full, {0.f, 0.f}};
        if (worker(&j)) { free_plan(&pl); return -1; }
        off[0] = j.off[0];
        off[1] = j.off[1];
    } else {
        pthread_t *tid = calloc(threads, sizeof *tid);
        job_t *jobs = calloc(threads, sizeof *jobs);
        for (t = 0; t < (size_t) threads; ++t) {
            jobs[t].pl = &pl;
            jobs[t].in = in;
            jobs[t].out = out;
            jobs[t].first = full * t / threads;
            jobs[t].last = full * (t + 1) / threads;
            jobs[t].off[0] = off[0];
            jobs[t].off[1] = off[1];
            if (pl.correct) correct_iq_state(&pl, in, jobs[t].first, jobs[t].last, off);
            pthread_create(&tid[t], NULL, worker, &jobs[t]);
        }
        for (t = 0; t < (size_t) threads; ++t) pthread_join(tid[t], NULL);
        free(jobs);
        free(tid);
    }
    written = (long long) (full * (pl.chunk >> 2));

    if (tail) {
        scratch_t s;
        if (make_scratch(&pl, &s)) { free_plan(&pl); return -1; }
        run_chunk(&pl, &s, in + full * pl.chunk, tail, out + written, off);
        free_scratch(&s);
        written += (long long) (tail >> 2);
    }
    free_plan(&pl);
    return written;
}

/* ------------------------------------------------------------------------ */
/* synthetic input, SURVEY.md section 8d                                     */
/* ------------------------------------------------------------------------ */

static uint64_t splitmix64_word(uint64_t seed, uint64_t j) {
    uint64_t z = seed + (j + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void orc_synth_bytes(uint64_t seed, uint64_t first_byte, size_t n, uint8_t *dst) {
    size_t i;
    for (i = 0; i < n; ++i) {
        const uint64_t g = first_byte + i;
        dst[i] = (uint8_t) (splitmix64_word(seed, g >> 3) >> (8 * (g & 7)));
    }
}
