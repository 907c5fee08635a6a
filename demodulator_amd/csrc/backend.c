/*
 * backend.c -- processMatrix / allocateBuffer over the demod_hip C ABI.
 *
 * Hand-off protocol as the reference defines it (/root/reference/src/main.c:72-91,
 * src/matrix.c:236-242): the producer fills args->buf under the mutex, stores
 * the byte count in args->bufSize and posts `full`; the backend takes the bytes
 * under the mutex and posts `empty`.  Unlike the reference consumer this one
 * reads bufSize and exitFlag exactly once per hand-off, under the mutex, so the
 * last chunk is never lost or duplicated, and a final
 * partial chunk of r bytes yields r/4 outputs.
 *
 * Ring mode (chunks of 1 KiB .. 32 MiB, the default): allocateBuffer hands the
 * producer the first slot of a pinned RING of chunk-sized slots, and after every
 * hand-off processMatrix points args->buf -- which the producer reads afresh
 * under the mutex each time, main.c:76 -- at the next slot.  The producer's fread
 * therefore lands in pinned memory next to the previous chunk: nothing is copied
 * on the host, and runs of chunks travel as one upload + one hot-path call + one
 * download (demod_ring_*: as many as arrived while the GPU was busy, at once
 * when it is idle).  A second thread writes finished batches to outFile, in
 * order, while the first keeps taking hand-offs.
 *
 * Copy mode (chunks below 1 KiB, DEMOD_SEAMLESS=1, or a hand-off buffer that did
 * not come from this allocateBuffer): every hand-off is copied into a staging
 * stage and pipelined three deep (demod_stream_submit / _collect).
 */
#define _GNU_SOURCE
#include "demod_backend.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "demod_hip.h"

/* The layout of consumerArgs IS the ABI between the reference's host and this backend
 * (include/matrix.h:43-57, measured on x86-64 with the reference's own build: SURVEY.md section 8 a7).
 * A mismatch must not wait for a GPU box to show: it fails the build. */
#if defined(__x86_64__) && defined(__linux__)
#define DEMOD_ABI_AT(field, off) \
    _Static_assert(offsetof(demod_consumer_args, field) == (off), "consumerArgs layout: " #field " is not at offset " #off)
DEMOD_ABI_AT(full, 0);
DEMOD_ABI_AT(empty, 8);
DEMOD_ABI_AT(mutex, 16);
DEMOD_ABI_AT(outFile, 56);
DEMOD_ABI_AT(buf, 64);
DEMOD_ABI_AT(exitFlag, 72);
DEMOD_ABI_AT(mode, 76);
DEMOD_ABI_AT(sampleRate, 80);
DEMOD_ABI_AT(lowpassIn, 96);
DEMOD_ABI_AT(lowpassOut, 112);
DEMOD_ABI_AT(epsilon, 128);
DEMOD_ABI_AT(inFilterDegree, 144);
DEMOD_ABI_AT(outFilterDegree, 152);
DEMOD_ABI_AT(bufSize, 160);
_Static_assert(sizeof(demod_consumer_args) == 176, "consumerArgs layout: size is not 176");
_Static_assert(sizeof(pthread_mutex_t) == 40 && sizeof(long double) == 16 && sizeof(size_t) == 8,
               "consumerArgs layout: member types differ from the reference host's");
#undef DEMOD_ABI_AT
#endif

static void die(const char *what, const char *detail) {
    fprintf(stderr, "demod backend: %s: %s\n", what, detail ? detail : "");
    exit(2);
}

/* the ring arena allocateBuffer made, waiting for the processMatrix that will use it
 * (allocateBuffer has no context argument: include/matrix.h:61) */
static struct {
    pthread_mutex_t lock;
    void *base;
    size_t chunk;
    unsigned slots;
} g_arena = {PTHREAD_MUTEX_INITIALIZER, NULL, 0, 0};

void allocateBuffer(void **buf, size_t len) {
    const unsigned slots = demod_ring_slots_for(len);
    if (demod_alloc_pinned(buf, slots ? (size_t) slots * len : len) != DEMOD_OK) die("allocateBuffer", demod_last_error(NULL));
    if (slots) {
        pthread_mutex_lock(&g_arena.lock);
        g_arena.base = *buf;
        g_arena.chunk = len;
        g_arena.slots = slots;
        pthread_mutex_unlock(&g_arena.lock);
    }
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

/* ------------------------------------------------------------------ ring mode */

struct writer {
    demod_ring *ring;
    demod_ctx *dm;
    FILE *out;
};

/* One thread writes: a second one on the same output file does not help -- writes to one
 * tmpfs/ext4 inode serialise on its lock (tried: two pwrite() halves per batch brought the
 * /dev/shm file -> file rate DOWN from 2.3 to 1.9 Gsamples/s, tools/host_path_rate.py). */
static double seconds(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static void *write_results(void *p) {
    struct writer *w = p;
    const size_t ob = (size_t) demod_output_bytes(w->dm);
    const char *env = getenv("DEMOD_RING_STATS");
    const int stats = env && env[0] == '1';
    double waiting = 0, writing = 0, bytes = 0;
    for (;;) {
        const void *res = NULL;
        uint64_t n = 0;
        const double t0 = stats ? seconds() : 0;
        if (demod_ring_collect(w->ring, &res, &n) != DEMOD_OK) die("collect", demod_last_error(w->dm));
        if (!n) break; /* the stream has ended and everything was handed out */
        const double t1 = stats ? seconds() : 0;
        fwrite(res, ob, n, w->out);
        if (stats) {
            waiting += t1 - t0;
            writing += seconds() - t1;
            bytes += (double) ob * (double) n;
        }
        if (demod_ring_release(w->ring) != DEMOD_OK) die("release", demod_last_error(w->dm));
    }
    fflush(w->out);
    if (stats && writing > 0)
        fprintf(stderr, "demod writer: %.3f s in fwrite (%.2f GB/s while writing), %.3f s waiting for results\n", writing,
                bytes / writing / 1e9, waiting);
    return NULL;
}

static void run_ring(demod_consumer_args *args, demod_ctx *dm, void *arena, size_t chunk, unsigned slots) {
    demod_ring *ring = NULL;
    struct writer w;
    pthread_t wt;
    if (demod_ring_create(dm, arena, slots, &ring) != DEMOD_OK) die("ring", demod_last_error(dm));
    w.ring = ring;
    w.dm = dm;
    w.out = args->outFile;
    if (pthread_create(&wt, NULL, write_results, &w)) die("pthread_create", strerror(errno));
    const char *env = getenv("DEMOD_RING_STATS");
    const int stats = env && env[0] == '1';
    double waiting = 0, handling = 0;
    for (;;) {
        size_t n;
        int flag;
        const double t0 = stats ? seconds() : 0;
        if (wait_full(args)) {
            (void) demod_ring_commit(ring, 0, 1);
            break;
        }
        const double t1 = stats ? seconds() : 0;
        pthread_mutex_lock(&args->mutex);
        n = args->bufSize;
        flag = args->exitFlag;
        if (n > chunk) n = chunk;
        n &= ~(size_t) 3;
        if (flag == -2 || flag == -3) n = 0; /* read error: nothing valid in the buffer */
        if (n < chunk && !flag) {
            /* a short read that does not end the input cannot be laid next to the chunk
             * that follows it; the reference's producers never make one (fread blocks until
             * the chunk is full or the input ends) */
            pthread_mutex_unlock(&args->mutex);
            die("hand-off", "short read in the middle of the input");
        }
        if (demod_ring_commit(ring, n, flag != 0) != DEMOD_OK) {
            pthread_mutex_unlock(&args->mutex);
            die("commit", demod_last_error(dm));
        }
        if (!flag) args->buf = demod_ring_input(ring); /* the producer's next fread goes here */
        pthread_mutex_unlock(&args->mutex);
        sem_post(args->empty);
        if (stats) {
            waiting += t1 - t0;
            handling += seconds() - t1;
        }
        if (flag) break;
    }
    if (stats)
        fprintf(stderr, "demod consumer: %.3f s waiting for hand-offs, %.3f s taking them (commit, next slot)\n", waiting, handling);
    pthread_join(wt, NULL);
    demod_ring_destroy(ring);
    /* the hand-off buffer is the backend's (allocateBuffer made it; the reference's CUDA
     * backend frees it the same way, cuda/matrix.cu:97) */
    args->buf = NULL;
    demod_free_pinned(arena);
}

/* ------------------------------------------------------------------ copy mode */

static void drain(demod_ctx *dm, FILE *out) {
    const void *res = NULL;
    uint64_t n = 0;
    if (demod_stream_collect(dm, &res, &n) != DEMOD_OK) die("collect", demod_last_error(dm));
    if (n) fwrite(res, (size_t) demod_output_bytes(dm), n, out);
}

static void run_copy(demod_consumer_args *args, demod_ctx *dm, size_t chunk, int seamless) {
    const int depth = demod_stream_depth();
    int pending = 0;
    unsigned char *held = NULL; /* seamless: the hand-off waiting for its look-ahead */
    size_t held_n = 0;
    uint64_t stream_pos = 0;
    unsigned char halo[DEMOD_HALO_BYTES];

    if (seamless) {
        held = malloc(chunk ? chunk : 1);
        if (!held) die("malloc", "out of memory");
    }
    for (;;) {
        size_t n;
        int flag;
        if (wait_full(args)) break;
        /* make room before touching the hand-off buffer: the oldest piece in
         * flight is written out while the producer is still parked on `empty` */
        if (pending == depth) {
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
             * bytes into a pinned staging stage and enqueues upload, kernel and
             * download without waiting for them.  Every hand-off is a chunk of its
             * own: a short one (end of input) is filtered as a short chunk. */
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
        if (pending == depth) {
            drain(dm, args->outFile);
            --pending;
        }
        if (demod_stream_submit(dm, held, held_n, &sp) != DEMOD_OK) die("submit", demod_last_error(dm));
        ++pending;
    }
    while (pending--) drain(dm, args->outFile);
    fflush(args->outFile);
    free(held);
}

/* ----------------------------------------------------------------------------- */

void *processMatrix(void *ctx) {
    demod_consumer_args *args = ctx;
    demod_params p;
    demod_ctx *dm = NULL;
    size_t chunk; /* B */
    void *arena = NULL;
    unsigned slots = 0;
    /* DEMOD_SEAMLESS=1: filter the stream as one chunk (demod_set_seamless); not reference behaviour */
    const char *env_seamless = getenv("DEMOD_SEAMLESS");
    int seamless;

    /* B: what allocateBuffer was asked for when the hand-off buffer is its ring (the
     * producer may already have overwritten args->bufSize with the byte count of its
     * first read, src/main.c:88); otherwise the field itself, read under the mutex and
     * rounded up to a power of two -- if the producer got there first the whole input is
     * shorter than one chunk, and a short only-chunk is filtered as a chunk of its own
     * length whatever B is */
    pthread_mutex_lock(&args->mutex);
    chunk = args->bufSize;
    pthread_mutex_lock(&g_arena.lock);
    if (g_arena.base && g_arena.base == args->buf) {
        arena = g_arena.base;
        chunk = g_arena.chunk;
        slots = g_arena.slots;
        g_arena.base = NULL;
    }
    pthread_mutex_unlock(&g_arena.lock);
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
    if (seamless && demod_set_seamless(dm, 1) != DEMOD_OK) die("seamless", demod_last_error(dm));

    if (arena && !seamless)
        run_ring(args, dm, arena, chunk, slots);
    else
        run_copy(args, dm, chunk, seamless);
    demod_destroy(dm);
    return NULL;
}
