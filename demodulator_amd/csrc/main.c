/*
 * main.c -- `demod`: the command-line host of the MI355X demodulation path.
 *
 * Same contract as the reference program (/root/reference/src/main.c:100-198,
 * README.md:9-31) so that it drops into the same pipes:
 *
 *     sox ... -traw -eunsigned-int -b8 -r192k - | demod -i - -o - -S96000 -l12500 | sox ... | dsd ...
 *
 *   -i FILE|-   uint8 interleaved IQ in          -o FILE|-   native float32 out
 *   -S rate     sampling rate (default 125000)   -l cutoff   output lowpass (default 12500)
 *   -L cutoff   input lowpass (default none)     -d/-D n     output / input filter degree
 *   -m bits     1: output filter Chebyshev-I, 2: input filter Chebyshev-I
 *   -e x        Chebyshev ripple parameter x/10  -b n        chunk = 262144 << n (n < 0 shifts right)
 *   -q 0|1|2|3  IQ conditioning (0 shiftOrigin, 1 correctIq, 2 highpassDc, 3 normalizeInput)   -c n   demod mode bits
 *   -r, -n      accepted and ignored, as in the reference
 *
 * Structure: the main thread is the producer (fread into the pinned hand-off
 * buffer), the backend's processMatrix() runs as the consumer thread -- the
 * reference's two-symbol backend boundary, here served by libdemod_backend.so.
 * Differences from the reference host, all deliberate:
 * a final partial read is processed, not raced; read errors end the run instead
 * of hanging it; "-" must be the whole argument to mean stdin/stdout.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "demod_backend.h"
#include "demod_hip.h"

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s -i <in|-> -o <out|-> [-S rate] [-l cutoff] [-L cutoff] [-d deg] [-D deg]\n"
            "          [-m mode] [-e ripple] [-b shift] [-q 0|1|2|3] [-c bits]\n",
            argv0);
}

static int run(FILE *in, demod_consumer_args *args) {
    pthread_t consumer;
    const size_t chunk = args->bufSize;
    int status = 0;

    allocateBuffer(&args->buf, chunk);
    if (pthread_create(&consumer, NULL, processMatrix, args)) {
        perror("pthread_create");
        return 1;
    }
    for (;;) {
        size_t got;
        int flag = 0;
        sem_wait(args->empty);
        pthread_mutex_lock(&args->mutex);
        got = fread(args->buf, 1, chunk, in);
        if (ferror(in)) {
            perror("read");
            flag = -2;
            got = 0;
            status = 1;
        } else if (got < chunk) {
            flag = EOF; /* short or empty read: end of input */
        }
        args->bufSize = got;
        args->exitFlag = flag;
        pthread_mutex_unlock(&args->mutex);
        sem_post(args->full);
        if (flag) break;
    }
    pthread_join(consumer, NULL);
    return status;
}

int main(int argc, char **argv) {
    demod_consumer_args args;
    sem_t full, empty;
    FILE *in = NULL;
    int opt, rc;

    memset(&args, 0, sizeof args);
    /* defaults of the reference, src/main.c:102-113 */
    args.bufSize = DEMOD_DEFAULT_BUF_SIZE;
    args.sampleRate = 125000.;
    args.lowpassIn = 0.;
    args.lowpassOut = 12500.;
    args.inFilterDegree = 0;
    args.outFilterDegree = 3;
    args.epsilon = .3;
    args.mode = 0x10;
    pthread_mutex_init(&args.mutex, NULL);
    sem_init(&empty, 0, 1);
    sem_init(&full, 0, 0);
    args.empty = &empty;
    args.full = &full;

    if (argc < 3) {
        usage(argv[0]);
        return -1; /* src/main.c:122-124 */
    }
    while ((opt = getopt(argc, argv, "i:o:r:L:l:S:D:d:e:m:b:c:q:n:")) != -1) {
        switch (opt) {
        case 'i':
            in = strcmp(optarg, "-") ? fopen(optarg, "rb") : stdin;
            if (!in) { perror(optarg); return 1; }
            break;
        case 'o':
            args.outFile = strcmp(optarg, "-") ? fopen(optarg, "wb") : stdout;
            if (!args.outFile) { perror(optarg); return 1; }
            break;
        case 'L': args.lowpassIn = strtold(optarg, NULL); break;
        case 'l': args.lowpassOut = strtold(optarg, NULL); break;
        case 'S': args.sampleRate = strtold(optarg, NULL); break;
        case 'D': args.inFilterDegree = strtoul(optarg, NULL, 10); break;
        case 'd': args.outFilterDegree = strtoul(optarg, NULL, 10); break;
        case 'e': args.epsilon = strtold(optarg, NULL) / 10.; break;
        case 'm': args.mode |= (uint8_t) strtoul(optarg, NULL, 10); break;
        case 'c': args.mode |= (uint8_t) (strtoul(optarg, NULL, 10) << 4); break;
        case 'q': args.mode |= (uint8_t) (strtoul(optarg, NULL, 10) << 2); break;
        case 'b': {
            const long shift = strtol(optarg, NULL, 10);
            if (labs(shift) < 17) /* src/main.c:166-172 */
                args.bufSize = shift < 1 ? (size_t) DEMOD_DEFAULT_BUF_SIZE >> -shift
                                         : (size_t) DEMOD_DEFAULT_BUF_SIZE << shift;
            break;
        }
        default: break; /* -r, -n and unknown options are ignored */
        }
    }
    if (!in || !args.outFile) {
        usage(argv[0]);
        return 1;
    }
    /* refuse up front what the backend would refuse (same messages as the library) */
    {
        demod_params p;
        demod_default_params(&p);
        p.sample_rate = (double) args.sampleRate;
        p.lowpass_in = (double) args.lowpassIn;
        p.lowpass_out = (double) args.lowpassOut;
        p.epsilon = (double) args.epsilon;
        p.in_degree = (uint32_t) args.inFilterDegree;
        p.out_degree = (uint32_t) args.outFilterDegree;
        p.mode = args.mode;
        p.buf_size = args.bufSize;
        if (demod_design_query(&p, 0, NULL, 0, NULL, 0, NULL, NULL, NULL) < 0) {
            fprintf(stderr, "demod: %s\n", demod_last_error(NULL));
            return 1;
        }
        if (args.bufSize < 4 || args.bufSize > demod_stream_capacity()) {
            fprintf(stderr, "demod: chunk size %zu is outside 4..%llu bytes\n", args.bufSize,
                    (unsigned long long) demod_stream_capacity());
            return 1;
        }
    }
    rc = run(in, &args);
    fclose(args.outFile);
    if (in != stdin) fclose(in);
    pthread_mutex_destroy(&args.mutex);
    sem_destroy(&full);
    sem_destroy(&empty);
    return rc;
}
