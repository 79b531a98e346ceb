// Micro-benchmark (experiment, not product): VALU issue rate of scalar vs packed fp32 multiplies and the
// latency of a dependent v_add_f32 chain on gfx950.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float m = 1.0000001f; const v2f pm = {1.0000001f, 0.9999999f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] * m;            // 8 independent v_mul_f32
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = p[i] * pm;           // 8 independent v_pk_mul_f32
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[0] = a[0] + a[1 + (i & 3)];  // 8 dependent v_add_f32
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int blocks, int iters, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 4.0 / 1024.0;  // 256 CUs x 4 SIMDs
    const double instr_per_simd = waves_per_simd * iters * 8.0;
    printf("%-28s blocks %5d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, blocks, ms, ms * 1e6 / instr_per_simd,
           ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 8192);
    for (int blocks : {256, 2048, 8192}) {
        run<0>("v_mul_f32 independent", blocks, 20000, d);
        run<1>("v_pk_mul_f32 independent", blocks, 20000, d);
        run<2>("v_add_f32 dependent chain", blocks, 20000, d);
    }
    return 0;
}
