/*
 * oracle/ref_stream_main.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * One reference stream per process: reads an orc_params record and the IQ
 * bytes from stdin, runs the reference's own functions over them
 * (ref_process_stream in ref_harness.c) and writes the float32 results to
 * stdout.  Exists for -q2, whose reference implementation keeps static state
 * sized by its first call (src/matrix.c:147-153) and reads behind a calloc'ed
 * buffer -- zeros in a process as fresh as the reference CLI's, for chunks of
 * 32 KiB and more (callocs of >= 128 KiB are private zero-filled mappings).
 * Built only where /root/reference exists, into oracle/_ref/ref_stream.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "demod_oracle.h"

long long ref_process_stream(const orc_params *p, const uint8_t *in, size_t n_bytes, float *out);

static int read_all(void *dst, size_t n) {
    size_t got = 0;
    while (got < n) {
        const size_t r = fread((char *) dst + got, 1, n - got, stdin);
        if (!r) return -1;
        got += r;
    }
    return 0;
}

int main(void) {
    orc_params p;
    uint64_t n_bytes;
    uint8_t *in;
    float *out;
    long long n;

    if (read_all(&p, sizeof p) || read_all(&n_bytes, sizeof n_bytes)) return 2;
    in = malloc(n_bytes ? n_bytes : 1);
    out = malloc((n_bytes / 4 + 1) * sizeof(float));
    if (!in || !out || read_all(in, n_bytes)) return 2;
    n = ref_process_stream(&p, in, n_bytes, out);
    if (n < 0) return 3;
    if (fwrite(out, sizeof(float), (size_t) n, stdout) != (size_t) n) return 4;
    return 0;
}
