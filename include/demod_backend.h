/*
 * demod_backend.h -- the reference's backend boundary, provided by
 * libdemod_backend.so as a thin adapter over demod_hip.h.
 *
 * The reference host (/root/reference/src/main.c) knows its compute backend
 * through exactly two C symbols and one struct:
 *
 *     void *processMatrix(void *ctx);              include/matrix.h:60, src/main.c:45,68
 *     void  allocateBuffer(void **buf, size_t len); include/matrix.h:61, src/main.c:46,65
 *     consumerArgs                                  include/matrix.h:43-57
 *
 * (its CUDA backend, cuda/matrix.cu:63,106, plugs in the same way).  A host
 * built with -DIS_NVIDIA and linked against libdemod_backend.so instead of
 * cuda/matrix.cu runs this MI355X path with no source change.
 */
#ifndef DEMOD_BACKEND_H
#define DEMOD_BACKEND_H

#include <pthread.h>
#include <semaphore.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Field-for-field the reference's consumerArgs (include/matrix.h:43-57); the
 * layout is the ABI between host and backend.  x86-64: size 176, offsets
 * full 0, empty 8, mutex 16, outFile 56, buf 64, exitFlag 72, mode 76,
 * sampleRate 80, lowpassIn 96, lowpassOut 112, epsilon 128,
 * inFilterDegree 144, outFilterDegree 152, bufSize 160. */
typedef struct demod_consumer_args {
    sem_t *full, *empty;     /* single-slot hand-off: producer posts full, backend posts empty */
    pthread_mutex_t mutex;   /* guards buf / bufSize / exitFlag during the hand-off */
    FILE *outFile;           /* written only by the backend */
    void *buf;               /* allocated by allocateBuffer, filled by the producer */
    int exitFlag;            /* 0 running, EOF (-1) end of input, -2 read error, -3 empty read */
    uint8_t mode;            /* ww|dd|qq|ff, default 0x10 (src/main.c:112) */
    long double sampleRate;  /* -S */
    long double lowpassIn;   /* -L */
    long double lowpassOut;  /* -l */
    long double epsilon;     /* -e / 10 */
    size_t inFilterDegree;   /* -D */
    size_t outFilterDegree;  /* -d */
    size_t bufSize;          /* chunk bytes; the producer overwrites it with each read's length */
} demod_consumer_args;

/* Consumer thread entry (pthread start routine): takes every chunk the producer
 * hands over, runs the hot path on the GPU and fwrite()s bufSize/4 float32 per
 * chunk to outFile, in order.  Returns NULL.  Exits the process with a message
 * if no MI355X context can be created (there is no CPU fallback). */
void *processMatrix(void *ctx);

/* Pinned, zeroed host buffer for the producer to read into. */
void allocateBuffer(void **buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* DEMOD_BACKEND_H */
