// GPU box: what does it cost to let the GPU write results straight into a file's pages?
// mmap a new /dev/shm file, hipHostRegister the mapping, copy device -> mapping, unregister.
//   hipcc -O2 -o register_probe register_probe.cpp && ./register_probe
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t win = 256u << 20;
    void *d = nullptr;
    if (hipMalloc(&d, win) != hipSuccess) return 1;
    hipMemset(d, 1, win);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        const char *path = "/dev/shm/register_probe.bin";
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT, 0600);
        double t0 = now();
        if (ftruncate(fd, win)) return 2;
        void *m = mmap(nullptr, win, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double t1 = now();
        hipError_t e = hipHostRegister(m, win, hipHostRegisterDefault);
        double t2 = now();
        if (e != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(e)); return 3; }
        hipMemcpy(m, d, win, hipMemcpyDeviceToHost);
        double t3 = now();
        hipHostUnregister(m);
        double t4 = now();
        munmap(m, win);
        close(fd);
        double t5 = now();
        printf("256 MiB window: ftruncate+mmap %.2f ms, register %.2f ms, D2H %.2f ms (%.1f GB/s), unregister %.2f ms, munmap %.2f ms -> %.2f GB/s overall\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, win / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (t5 - t4) * 1e3, win / (t5 - t0) / 1e9);
        unlink(path);
    }
    // for comparison: write() of the same bytes from pinned memory
    void *h = nullptr;
    hipHostMalloc(&h, win, hipHostMallocDefault);
    memset(h, 2, win);
    for (int rep = 0; rep < 2; ++rep) {
        const char *path = "/dev/shm/register_probe.bin";
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT, 0600);
        double t0 = now();
        for (size_t o = 0; o < win; o += 8u << 20)
            if (write(fd, (char *) h + o, 8u << 20) < 0) return 4;
        double t1 = now();
        close(fd);
        unlink(path);
        printf("write() of 256 MiB from pinned memory: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, win / (t1 - t0) / 1e9);
    }
    return 0;
}
