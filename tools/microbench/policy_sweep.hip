// policy_sweep.hip -- the streaming kernel's memory pattern as a bare copy (a wavefront asks for its 4 rows up front,
// stores each as it lands, workgroups of 4 wavefronts, segments interleaved over the 8 XCDs as in demod_rows) under every
// cache-policy suffix gfx950 has for loads and stores: which pair streams best?
//   hipcc --offload-arch=gfx950 -O3 -o policy_sweep policy_sweep.hip && ./policy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define LOADS(X) X(0, "") X(1, " nt") X(2, " sc0") X(3, " sc1") X(4, " sc0 sc1") X(5, " sc0 nt") X(6, " sc1 nt") X(7, " sc0 sc1 nt")

template <int LP> __device__ __forceinline__ void ld(u32x4 &v, const void *p) {
#define X(i, s) if constexpr (LP == i) asm volatile("global_load_dwordx4 %0, %1, off" s : "=v"(v) : "v"(p) : "memory");
    LOADS(X)
#undef X
}
template <int SP> __device__ __forceinline__ void st(void *p, const u32x4 &v) {
#define X(i, s) if constexpr (SP == i) asm volatile("global_store_dwordx4 %0, %1, off" s : : "v"(p), "v"(v) : "memory");
    LOADS(X)
#undef X
}

// MAP: which segments a workgroup (= an XCD, round-robin) takes: 0 as demod_rows (XCD x streams segments x, x+8, x+16, x+24 of a
// group of 32), 1 four consecutive ones, 2 two pairs (2x, 2x+1, 2x+16, 2x+17), 3 x, x+8, ... of a group of 64 over 16 workgroups
template <int LP, int SP, int ROWS, int MAP = 0> __global__ __launch_bounds__(256, 2) void k(const uint8_t *in, uint8_t *out, uint32_t n_seg) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t seg;
    if constexpr (MAP == 0) seg = (blockIdx.x / 8u) * 32u + wave * 8u + blockIdx.x % 8u;
    else if constexpr (MAP == 1) seg = blockIdx.x * 4u + wave;
    else if constexpr (MAP == 2) seg = (blockIdx.x / 8u) * 32u + (wave >> 1) * 16u + (blockIdx.x % 8u) * 2u + (wave & 1u);
    else seg = (blockIdx.x / 16u) * 64u + (wave * 2u + (blockIdx.x % 16u) / 8u) * 8u + blockIdx.x % 8u;
    if (seg >= n_seg) return;
    const size_t base = (size_t) seg * ROWS * 1024 + 16u * lane;
    u32x4 q[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) ld<LP>(q[r], in + base + (size_t) (ROWS - 1 - r) * 1024);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q[r]) : "i"(ROWS - 1) : "memory");   // r stores issued, ROWS-1-r loads younger
        q[r].x += 1;
        st<SP>(out + base + (size_t) (ROWS - 1 - r) * 1024, q[r]);
    }
}

static const char *names[] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
static uint8_t *a, *b;
static const size_t bytes = 1ull << 32;

template <int LP, int SP, int ROWS, int MAP = 0> float run() {
    const uint32_t n_seg = (uint32_t) (bytes / 1024 / ROWS);
    const uint32_t blocks = (n_seg + 3) / 4;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<LP, SP, ROWS, MAP>), dim3(blocks), dim3(256), 0, 0, a, b, n_seg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL((k<LP, SP, ROWS, MAP>), dim3(blocks), dim3(256), 0, 0, a, b, n_seg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 6;
}
template <int LP, int ROWS> void row() {
    float t[8] = {run<LP, 0, ROWS>(), run<LP, 1, ROWS>(), run<LP, 2, ROWS>(), run<LP, 3, ROWS>(), run<LP, 4, ROWS>(), run<LP, 5, ROWS>(), run<LP, 6, ROWS>(), run<LP, 7, ROWS>()};
    printf("%d rows/wave, loads %-11s:", ROWS, names[LP]);
    for (int s = 0; s < 8; ++s) printf("  %.3f", 2.0 * bytes / (t[s] * 1e-3) / 8e12);
    printf("\n");
}
int main() {
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    printf("fraction of 8 TB/s (read + write); columns = store policy:");
    for (int s = 0; s < 8; ++s) printf("  [%s]", names[s]);
    printf("\n");
    row<0, 4>(); row<1, 4>(); row<2, 4>(); row<3, 4>(); row<4, 4>(); row<5, 4>(); row<6, 4>(); row<7, 4>();
    row<0, 1>(); row<1, 1>(); row<4, 1>(); row<7, 1>();
    row<1, 2>(); row<1, 8>();
    printf("segment-to-workgroup maps (loads nt; stores nt / sc1 nt), 4 rows per wavefront: as demod_rows %.3f / %.3f, consecutive %.3f / %.3f, pairs %.3f / %.3f, group of 64 %.3f / %.3f\n",
           2.0 * bytes / (run<1, 1, 4, 0>() * 1e-3) / 8e12, 2.0 * bytes / (run<1, 6, 4, 0>() * 1e-3) / 8e12, 2.0 * bytes / (run<1, 1, 4, 1>() * 1e-3) / 8e12,
           2.0 * bytes / (run<1, 6, 4, 1>() * 1e-3) / 8e12, 2.0 * bytes / (run<1, 1, 4, 2>() * 1e-3) / 8e12, 2.0 * bytes / (run<1, 6, 4, 2>() * 1e-3) / 8e12,
           2.0 * bytes / (run<1, 1, 4, 3>() * 1e-3) / 8e12, 2.0 * bytes / (run<1, 6, 4, 3>() * 1e-3) / 8e12);
    printf("the same with 1 / 2 / 8 rows per wavefront, consecutive: %.3f / %.3f / %.3f (sc1 nt stores)\n", 2.0 * bytes / (run<1, 6, 1, 1>() * 1e-3) / 8e12,
           2.0 * bytes / (run<1, 6, 2, 1>() * 1e-3) / 8e12, 2.0 * bytes / (run<1, 6, 8, 1>() * 1e-3) / 8e12);
    return 0;
}
