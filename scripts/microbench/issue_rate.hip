// Micro-benchmark (experiment, not product): VALU issue cost on gfx950 measured with inline asm so the
// compiler cannot merge or drop instructions.
//   mode 0: 16 independent v_mul_f32 per loop trip, all 64 lanes active
//   mode 1: same, only lanes 0..4 active (does a nearly empty EXEC issue faster?)
//   mode 2: 16 dependent v_add_f32 per trip (chain), all lanes
//   mode 3: 16 dependent v_add_f32 per trip, lanes 0..4
// Reports ns per wave-instruction per SIMD for 1, 2, 4, 8 resident waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    const float m = 1.0000001f;
    const bool few = (MODE == 1 || MODE == 3);
    if (!few || threadIdx.x < 5) {
        for (int it = 0; it < iters; ++it) {
            if (MODE < 2) {
                asm volatile(
                    "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                    "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                    "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                    "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                    : "v"(m));
            } else {
                asm volatile(
                    "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
                    "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
                    "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
                    "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %3\n v_add_f32 %0, %0, %4\n"
                    : "+v"(a0)
                    : "v"(a1), "v"(a2), "v"(a3), "v"(a4));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int iters, float *d) {
    const int blocks = 256 * 4 * waves_per_simd;  // one-wave workgroups; the dispatcher spreads them over CUs / SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * iters * 16.0;
    printf("%-36s waves/SIMD %d  %.3f ms  %.3f ns per wave-instr per SIMD, %.3f ns per instr per wave\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / (iters * 16.0));
}

int main() {
    float *d;
    hipMalloc(&d, sizeof(float) * 64 * 256 * 4 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mul_f32 independent, 64 lanes", w, 20000, d);
        run<1>("v_mul_f32 independent, 5 lanes", w, 20000, d);
        run<2>("v_add_f32 dependent, 64 lanes", w, 20000, d);
        run<3>("v_add_f32 dependent, 5 lanes", w, 20000, d);
    }
    return 0;
}
