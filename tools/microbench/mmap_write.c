// mmap_write.c -- can a result file in /dev/shm be written faster than one write() stream?  write() takes the
// inode lock (two pwrite() halves were SLOWER, profiles/r2_host_path_rates.txt); stores into a shared mapping
// fault their pages in without it, so several threads can fill one window.
//   gcc -O2 -o mmap_write mmap_write.c -lpthread && ./mmap_write <threads> <window MiB>
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
typedef struct { char *dst; const char *src; size_t n; } job_t;
static void *cp(void *a) { job_t *j = a; memcpy(j->dst, j->src, j->n); return NULL; }
int main(int argc, char **argv) {
    size_t total = (size_t) 2 << 30, batch = (size_t) (argc > 2 ? atoi(argv[2]) : 8) << 20;
    int T = argc > 1 ? atoi(argv[1]) : 4;
    char *src = malloc(batch * 4);
    memset(src, 7, batch * 4);
    const char *path = "/dev/shm/mw_test.bin";
    /* 1: write() */
    for (int rep = 0; rep < 2; ++rep) {
        unlink(path);
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        double t0 = now();
        for (size_t o = 0; o < total; o += batch) if (write(fd, src + (o % (batch * 4)), batch) != (ssize_t) batch) return 1;
        double dt = now() - t0;
        close(fd);
        printf("write() %zu MiB calls: %.2f GB/s\n", batch >> 20, total / dt / 1e9);
    }
    /* 2: mmap window + T threads */
    for (int rep = 0; rep < 2; ++rep) {
        unlink(path);
        int fd = open(path, O_CREAT | O_TRUNC | O_RDWR, 0600);
        double t0 = now();
        size_t grown = 0;
        for (size_t o = 0; o < total; o += batch) {
            if (o + batch > grown) { grown = o + ((size_t) 1 << 30); if (ftruncate(fd, grown)) return 2; }
            char *w = mmap(NULL, batch, PROT_READ | PROT_WRITE, MAP_SHARED, fd, o);
            if (w == MAP_FAILED) return 3;
            pthread_t th[64]; job_t jb[64];
            size_t per = batch / T;
            for (int i = 0; i < T; ++i) { jb[i] = (job_t){w + i * per, src + (o % (batch * 4)) + i * per, per}; if (i) pthread_create(&th[i], NULL, cp, &jb[i]); }
            cp(&jb[0]);
            for (int i = 1; i < T; ++i) pthread_join(th[i], NULL);
            munmap(w, batch);
        }
        if (ftruncate(fd, total)) return 4;
        double dt = now() - t0;
        close(fd);
        printf("mmap + %d threads, %zu MiB windows: %.2f GB/s\n", T, batch >> 20, total / dt / 1e9);
    }
    unlink(path);
    return 0;
}
