/*
 * backend.c -- processMatrix / allocateBuffer over the demod_hip C ABI.
 *
 * Hand-off protocol as the reference defines it (/root/reference/src/main.c:72-91,
 * src/matrix.c:236-242): the producer fills args->buf under the mutex, stores
 * the byte count in args->bufSize and posts `full`; the backend copies the bytes
 * out under the mutex and posts `empty`.  Unlike the reference consumer this one
 * reads bufSize and exitFlag exactly once per hand-off, under the mutex, so the
 * last chunk is never lost or duplicated, and a final
 * partial chunk of r bytes yields r/4 outputs.  The chunk size B is read the same
 * way at start-up: the producer overwrites bufSize with the byte count of its
 * first read as soon as it wins the mutex (src/main.c:88), so what is found there
 * is either B or -- when the whole input is shorter than one chunk -- that count,
 * which is rounded up to the next power of two: a short only-chunk is filtered
 * as a chunk of its own length whatever B is, so the result is the same.
 *
 * GPU work is pipelined two deep (demod_stream_submit / _collect): while chunk
 * i is uploaded, demodulated and downloaded, chunk i-1 is written to outFile
 * and the producer reads chunk i+1.
 */
#define _GNU_SOURCE
#include "demod_backend.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "demod_hip.h"

static void die(const char *what, const char *detail) {
    fprintf(stderr, "demod backend: %s: %s\n", what, detail ? detail : "");
    exit(2);
}

void allocateBuffer(void **buf, size_t len) {
    if (demod_alloc_pinned(buf, len) != DEMOD_OK) die("allocateBuffer", demod_last_error(NULL));
}

/* sem_wait that gives up when the producer has stopped with an error flag
 * without posting (src/main.c:78-87 breaks out of its loop holding the mutex). */
static int wait_full(demod_consumer_args *args) {
    for (;;) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        ts.tv_nsec += 200 * 1000 * 1000;
        if (ts.tv_nsec >= 1000000000L) {
            ts.tv_sec += 1;
            ts.tv_nsec -= 1000000000L;
        }
        if (sem_timedwait(args->full, &ts) == 0) return 0;
        if (errno == EINTR) continue;
        if (errno == ETIMEDOUT) {
            const int flag = args->exitFlag;
            if (flag == -2 || flag == -3) return -1; /* producer bailed out */
            continue;
        }
        return -1;
    }
}

static void drain(demod_ctx *dm, FILE *out) {
    const void *res = NULL;
    uint64_t n = 0;
    if (demod_stream_collect(dm, &res, &n) != DEMOD_OK) die("collect", demod_last_error(dm));
    if (n) fwrite(res, (size_t) demod_output_bytes(dm), n, out);
}

void *processMatrix(void *ctx) {
    demod_consumer_args *args = ctx;
    demod_params p;
    demod_ctx *dm = NULL;
    size_t chunk; /* B */
    int pending = 0;
    /* DEMOD_SEAMLESS=1: filter the stream as one chunk (demod_set_seamless); not reference behaviour */
    const char *env_seamless = getenv("DEMOD_SEAMLESS");
    int seamless;
    unsigned char *held = NULL; /* seamless: the hand-off waiting for its look-ahead */
    size_t held_n = 0;
    uint64_t stream_pos = 0;
    unsigned char halo[DEMOD_HALO_BYTES];

    pthread_mutex_lock(&args->mutex);
    chunk = args->bufSize;
    pthread_mutex_unlock(&args->mutex);
    {
        size_t b = 4;
        while (b < chunk) b <<= 1;
        chunk = b;
    }
    seamless = env_seamless && env_seamless[0] == '1' && chunk >= 1024;
    demod_default_params(&p);
    p.sample_rate = (double) args->sampleRate;
    p.lowpass_in = (double) args->lowpassIn;
    p.lowpass_out = (double) args->lowpassOut;
    p.epsilon = (double) args->epsilon;
    p.in_degree = (uint32_t) args->inFilterDegree;
    p.out_degree = (uint32_t) args->outFilterDegree;
    p.mode = args->mode;
    p.buf_size = chunk;
    if (demod_create(&p, 0, &dm) != DEMOD_OK) die("cannot start the MI355X demodulation path", demod_last_error(NULL));
    if (chunk > demod_stream_capacity()) die("chunk size", "larger than the staging capacity of the streaming host path");

    {
        /* DEMOD_OUT_S16=<scale>: write signed 16-bit samples, clamp(y * scale, -1, 1) * 32767
         * (demod_set_output_format); not reference behaviour */
        const char *env_s16 = getenv("DEMOD_OUT_S16");
        if (env_s16 && env_s16[0]) {
            const float scale = strtof(env_s16, NULL);
            if (demod_set_output_format(dm, DEMOD_OUT_S16, scale) != DEMOD_OK) die("DEMOD_OUT_S16", demod_last_error(dm));
        }
    }
    if (seamless) {
        if (demod_set_seamless(dm, 1) != DEMOD_OK) die("seamless", demod_last_error(dm));
        held = malloc(chunk ? chunk : 1);
        if (!held) die("malloc", "out of memory");
    }

    for (;;) {
        size_t n;
        int flag;
        if (wait_full(args)) break;
        /* make room before touching the hand-off buffer: the oldest piece in
         * flight is written out while the producer is still parked on `empty` */
        if (pending == 2) {
            drain(dm, args->outFile);
            --pending;
        }
        pthread_mutex_lock(&args->mutex);
        n = args->bufSize;
        flag = args->exitFlag;
        if (n > chunk) n = chunk;
        n &= ~(size_t) 3;
        if (!seamless) {
            /* copy out under the mutex, as the protocol requires: submit copies the
             * bytes into a pinned staging slot and enqueues upload, kernel and
             * download on that slot's stream without waiting for them.  Every
             * hand-off is a chunk of its own: a short one (end of input) is
             * filtered as a short chunk. */
            if (n && demod_stream_submit(dm, args->buf, n, NULL) != DEMOD_OK) {
                pthread_mutex_unlock(&args->mutex);
                die("submit", demod_last_error(dm));
            }
            if (n) ++pending;
        } else {
            /* seamless: a hand-off can only be filtered once the first bytes of
             * the next one are known (they are its look-ahead), so it is held
             * back by one hand-off; the last one ends the stream. */
            if (held_n) {
                demod_span sp;
                memset(&sp, 0, sizeof sp);
                sp.chunk_offset = stream_pos;
                sp.end_of_stream = n ? 0 : 1;
                if (n) {
                    memset(halo, 0, sizeof halo);
                    memcpy(halo, args->buf, n < sizeof halo ? n : sizeof halo);
                    sp.halo = halo;
                }
                if (demod_stream_submit(dm, held, held_n, &sp) != DEMOD_OK) {
                    pthread_mutex_unlock(&args->mutex);
                    die("submit", demod_last_error(dm));
                }
                ++pending;
                stream_pos += held_n;
                held_n = 0;
            }
            if (n) {
                memcpy(held, args->buf, n);
                held_n = n;
            }
        }
        pthread_mutex_unlock(&args->mutex);
        sem_post(args->empty);
        if (flag) break;
    }
    if (seamless && held_n) { /* the stream's last hand-off */
        demod_span sp;
        memset(&sp, 0, sizeof sp);
        sp.chunk_offset = stream_pos;
        sp.end_of_stream = 1;
        if (pending == 2) {
            drain(dm, args->outFile);
            --pending;
        }
        if (demod_stream_submit(dm, held, held_n, &sp) != DEMOD_OK) die("submit", demod_last_error(dm));
        ++pending;
    }
    while (pending--) drain(dm, args->outFile);
    fflush(args->outFile);
    free(held);
    demod_destroy(dm);
    return NULL;
}
