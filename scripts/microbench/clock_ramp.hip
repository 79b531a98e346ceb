// Micro-benchmark (experiment, not product): the shader clock a short kernel actually runs at, as a function of how long the
// chip has been kept busy.  A one-wave-per-SIMD kernel runs a fixed dependent v_add_f32 chain and stamps s_memtime (shader
// clock ticks) and s_memrealtime (100 MHz); the host launches it back to back and prints the clock and the time per add
// after 0 / 10 ms / 100 ms / 1 s / 3 s of continuous launches.
//   hipcc --offload-arch=gfx950 -O3 clock_ramp.hip -o clock_ramp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void __launch_bounds__(256) k(unsigned long long *stamps, float *out, int adds) {
    float a = 1.0f + threadIdx.x * 1e-3f, b = 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < adds; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a = a + b;
            asm volatile("" : "+v"(a));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        stamps[0] = t1 - t0;
        stamps[1] = r1 - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    unsigned long long *d_st, h_st[2];
    float *d_out;
    (void)hipMalloc(&d_st, 16);
    (void)hipMalloc(&d_out, sizeof(float) * 256 * 256);
    const int adds = 8000;  // ~40 us per launch
    auto probe = [&](const char *label) {
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_st, d_out, adds);
        (void)hipMemcpy(h_st, d_st, 16, hipMemcpyDeviceToHost);
        const double us = h_st[1] * 0.01;
        printf("%-28s kernel %.1f us, shader clock %.0f MHz, %.2f ns = %.2f shader cycles per dependent add (one wave per SIMD)\n", label, us,
               h_st[0] / us, us * 1e3 / adds, (double)h_st[0] / adds);
    };
    probe("cold (first launch)");
    probe("second launch");
    const double marks[] = {0.01, 0.1, 0.5, 1.0, 2.0, 4.0};
    auto t0 = std::chrono::steady_clock::now();
    int mi = 0;
    while (mi < 6) {
        for (int i = 0; i < 50; ++i) {
            hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_st, d_out, adds);
        }
        (void)hipDeviceSynchronize();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (el >= marks[mi]) {
            char label[64];
            snprintf(label, sizeof(label), "after %.2f s busy", el);
            probe(label);
            ++mi;
        }
    }
    return 0;
}
