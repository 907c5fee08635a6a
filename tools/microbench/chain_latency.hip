// chain_latency.hip -- what one wavefront alone pays per DEPENDENT float operation on gfx950
// (the regime of correct_iq_offsets: a long literal recurrence, one wavefront per CU).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o chain_latency chain_latency.hip && ./chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float pf2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void chain(float *out, int n, float e, float fa, float fb) {
    float x = out[threadIdx.x], y = x + 1.f;
    pf2 p = {x, y};
    const pf2 e2 = {e, e};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (KIND == 0) {            // one dependent v_add_f32 per step
                x = x + fa;
                asm volatile("" : "+v"(x));
            } else if constexpr (KIND == 1) {     // one dependent v_pk_add_f32 per step
                p = p + pf2{fa, fb};
                asm volatile("" : "+v"(p));
            } else if constexpr (KIND == 2) {     // the correctIq step, two scalar chains
                const float f0 = fa - x, m0 = fb - x, f1 = fa - y, m1 = fb - y;
                x += (f0 + m0) * e;
                y += (f1 + m1) * e;
                asm volatile("" : "+v"(x), "+v"(y));
            } else if constexpr (KIND == 3) {     // the correctIq step, packed
                const pf2 f = pf2{fa, fb} - p, m = pf2{fb, fa} - p;
                p = p + (f + m) * e2;
                asm volatile("" : "+v"(p));
            } else if constexpr (KIND == 4) {     // one dependent v_fma_f32 per step
                x = __builtin_fmaf(x, e, fa);
                asm volatile("" : "+v"(x));
            } else {                              // one dependent v_mul_f32 per step
                x = x * e;
                asm volatile("" : "+v"(x));
            }
        }
    }
    out[threadIdx.x] = x + y + p.x + p.y;
}

template <int KIND>
static void run(const char *name, int ops, int waves_per_cu) {
    float *d;
    hipMalloc(&d, 64 * sizeof(float));
    hipMemset(d, 0, 64 * sizeof(float));
    const int n = 1 << 16;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * waves_per_cu;
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(64), 0, 0, d, 64, 0.9992f, 3.f, 5.f);
    hipEventRecord(a);
    hipLaunchKernelGGL(chain<KIND>, dim3(blocks), dim3(64), 0, 0, d, n, 0.9992f, 3.f, 5.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / ((double) n * 16);
    printf("%-44s %d wave/CU: %7.2f ns per step = %6.1f cycles at 2.4 GHz (%d dependent ops: %5.1f each)\n", name, waves_per_cu, ns,
           ns * 2.4, ops, ns * 2.4 / ops);
    hipFree(d);
}

int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_add_f32 chain", 1, w);
        run<5>("v_mul_f32 chain", 1, w);
        run<4>("v_fma_f32 chain", 1, w);
        run<1>("v_pk_add_f32 chain", 1, w);
        run<2>("correctIq step, two scalar chains", 4, w);
        run<3>("correctIq step, packed", 4, w);
    }
    return 0;
}
