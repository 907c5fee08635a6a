// stream_copy.hip -- which streaming pattern reaches the HBM copy ceiling on MI355X?
// 8 GiB read + 8 GiB written per launch, 16 B per lane per access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// classic: one 16-byte element per thread
template <bool NT> __global__ __launch_bounds__(256) void k_classic(const u32x4 *in, u32x4 *out, size_t n) {
    size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 v = ld<NT>(in + i); v.x += 1; st<NT>(out + i, v); }
}
// grid-stride persistent
template <bool NT> __global__ __launch_bounds__(256) void k_stride(const u32x4 *in, u32x4 *out, size_t n) {
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        u32x4 v = ld<NT>(in + i); v.x += 1; st<NT>(out + i, v);
    }
}
// each wave owns SEG consecutive 1 KiB rows; DIR=+1 ascending, -1 descending
template <bool NT, int DIR> __global__ __launch_bounds__(256) void k_seg(const u32x4 *in, u32x4 *out, size_t n, int seg) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t row0 = wave * seg;
    for (int j = 0; j < seg; ++j) {
        const size_t row = DIR > 0 ? row0 + j : row0 + (seg - 1 - j);
        const size_t i = row * 64 + lane;
        if (i < n) { u32x4 v = ld<NT>(in + i); v.x += 1; st<NT>(out + i, v); }
    }
}
// overlapped tiles: wave j reads 64 quads starting at quad 63*j and writes the first 63 of them
template <bool NT> __global__ __launch_bounds__(256) void k_overlap63(const u32x4 *in, u32x4 *out, size_t n) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t i = wave * 63 + lane;
    if (i < n) {
        u32x4 v = ld<NT>(in + i); v.x += 1;
        if (lane < 63) st<NT>(out + i, v);
    }
}
// same, tiles restart at every 16384-quad chunk (261 tiles per chunk, last one short)
template <bool NT> __global__ __launch_bounds__(256) void k_overlap63_chunked(const u32x4 *in, u32x4 *out, size_t n) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t wave = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t chunk = wave / 261, tile = wave % 261;
    const size_t q = tile * 63 + lane;
    const size_t i = chunk * 16384 + q;
    if (q < 16384 && i < n) {
        u32x4 v = ld<NT>(in + i); v.x += 1;
        if (lane < 63) st<NT>(out + i, v);
    }
}

// persistent, software-pipelined: every wavefront walks units of U rows at a stride of the
// whole grid; loads of the next unit are issued before the current one is consumed; stores
// are uncounted (inline asm) so that hipcc's vmcnt bookkeeping sees only loads
__device__ __forceinline__ void st_uncounted(u32x4 *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int U> __global__ __launch_bounds__(256) void k_pipe(const u32x4 *in, u32x4 *out, size_t n_units) {
    const uint32_t lane = threadIdx.x & 63;
    const size_t W = (size_t) gridDim.x * 4;
    size_t u = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_units) return;
    u32x4 a[U], b[U];
#pragma unroll
    for (int k = 0; k < U; ++k) a[k] = __builtin_nontemporal_load(in + (u * U + k) * 64 + lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    while (true) {
        size_t un = u + W < n_units ? u + W : u;
#pragma unroll
        for (int k = 0; k < U; ++k) b[k] = __builtin_nontemporal_load(in + (un * U + k) * 64 + lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < U; ++k) { u32x4 v = a[k]; v.x += 1; st_uncounted(out + (u * U + k) * 64 + lane, v); }
        if (u + W >= n_units) break;
        u += W;
        un = u + W < n_units ? u + W : u;
#pragma unroll
        for (int k = 0; k < U; ++k) a[k] = __builtin_nontemporal_load(in + (un * U + k) * 64 + lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < U; ++k) { u32x4 v = b[k]; v.x += 1; st_uncounted(out + (u * U + k) * 64 + lane, v); }
        if (u + W >= n_units) break;
        u += W;
    }
}

// block-cooperative segment: the 4 waves of a block interleave rows of a 4*SEG-row tile
template <bool NT, int DIR> __global__ __launch_bounds__(256) void k_blockseg(const u32x4 *in, u32x4 *out, size_t n, int seg) {
    const size_t row0 = (size_t) blockIdx.x * 4 * seg;
    for (int j = 0; j < seg; ++j) {
        const size_t i = (row0 + (DIR > 0 ? j : seg - 1 - j) * 4) * 64 + threadIdx.x;
        if (i < n) { u32x4 v = ld<NT>(in + i); v.x += 1; st<NT>(out + i, v); }
    }
}

template <class F> void timeit(const char *name, F launch, size_t bytes) {
    for (int i = 0; i < 2; ++i) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %.3f ms  %.2f TB/s  frac %.3f\n", name, ms, 2.0 * bytes / ms / 1e9, 2.0 * bytes / ms / 1e9 / 8.0);
}

int main() {
    const size_t bytes = 1ull << 33, n = bytes / 16;
    u32x4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    const unsigned nb = (unsigned) (n / 256);
    timeit("classic default", [&] { hipLaunchKernelGGL(k_classic<false>, dim3(nb), dim3(256), 0, 0, a, b, n); }, bytes);
    timeit("classic nt", [&] { hipLaunchKernelGGL(k_classic<true>, dim3(nb), dim3(256), 0, 0, a, b, n); }, bytes);
    timeit("grid-stride 2048 blocks default", [&] { hipLaunchKernelGGL(k_stride<false>, dim3(2048), dim3(256), 0, 0, a, b, n); }, bytes);
    timeit("grid-stride 2048 blocks nt", [&] { hipLaunchKernelGGL(k_stride<true>, dim3(2048), dim3(256), 0, 0, a, b, n); }, bytes);
    {
        const unsigned g63 = (unsigned) ((n / 63 + 4) / 4);
        timeit("overlap-63 tiles default", [&] { hipLaunchKernelGGL(k_overlap63<false>, dim3(g63), dim3(256), 0, 0, a, b, n); }, bytes);
        timeit("overlap-63 tiles nt", [&] { hipLaunchKernelGGL(k_overlap63<true>, dim3(g63), dim3(256), 0, 0, a, b, n); }, bytes);
        const unsigned gc = (unsigned) ((n / 16384 * 261 + 3) / 4);
        timeit("overlap-63 per-chunk tiles default", [&] { hipLaunchKernelGGL(k_overlap63_chunked<false>, dim3(gc), dim3(256), 0, 0, a, b, n); }, bytes);
        timeit("overlap-63 per-chunk tiles nt", [&] { hipLaunchKernelGGL(k_overlap63_chunked<true>, dim3(gc), dim3(256), 0, 0, a, b, n); }, bytes);
    }
    for (int bpc : {4, 5, 6, 8}) {
        char nm[96];
        snprintf(nm, sizeof nm, "pipelined persistent U=4, %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<4>, dim3(256 * bpc), dim3(256), 0, 0, a, b, n / 64 / 4); }, bytes);
        snprintf(nm, sizeof nm, "pipelined persistent U=2, %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<2>, dim3(256 * bpc), dim3(256), 0, 0, a, b, n / 64 / 2); }, bytes);
        snprintf(nm, sizeof nm, "pipelined persistent U=1, %d blocks/CU", bpc);
        timeit(nm, [&] { hipLaunchKernelGGL(k_pipe<1>, dim3(256 * bpc), dim3(256), 0, 0, a, b, n / 64); }, bytes);
    }
    for (int seg : {1, 2, 4}) {
        char nm[96];
        const unsigned g = (unsigned) (n / 64 / seg / 4);
        snprintf(nm, sizeof nm, "wave-segment %d rows ascending default", seg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_seg<false, 1>), dim3(g), dim3(256), 0, 0, a, b, n, seg); }, bytes);
        snprintf(nm, sizeof nm, "wave-segment %d rows descending default", seg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_seg<false, -1>), dim3(g), dim3(256), 0, 0, a, b, n, seg); }, bytes);
        snprintf(nm, sizeof nm, "wave-segment %d rows ascending nt", seg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_seg<true, 1>), dim3(g), dim3(256), 0, 0, a, b, n, seg); }, bytes);
        snprintf(nm, sizeof nm, "wave-segment %d rows descending nt", seg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_seg<true, -1>), dim3(g), dim3(256), 0, 0, a, b, n, seg); }, bytes);
        snprintf(nm, sizeof nm, "block-tile 4x%d rows ascending default", seg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_blockseg<false, 1>), dim3(g), dim3(256), 0, 0, a, b, n, seg); }, bytes);
        snprintf(nm, sizeof nm, "block-tile 4x%d rows descending default", seg);
        timeit(nm, [&] { hipLaunchKernelGGL((k_blockseg<false, -1>), dim3(g), dim3(256), 0, 0, a, b, n, seg); }, bytes);
    }
    return 0;
}
