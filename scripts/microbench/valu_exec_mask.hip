// Does a wave64 VALU instruction cost fewer cycles of the SIMD's vector unit when part of EXEC is zero?  The exact-order chain waves keep
// 20 lanes busy (5 sums x a quad); where a call is bound by vector issue (the headline: 74 % VALU utilisation), a chain whose upper lanes
// are switched OFF would be cheaper IF the hardware skips empty 16-lane passes.  Saturating launch (8 waves per SIMD), each wave a long
// chain of dependent v_add_f32_dpp (the chain's instruction); time per add against the EXEC mask.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_exec_mask scripts/microbench/valu_exec_mask.hip && /tmp/valu_exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>

#define ADD "v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define ADD8 ADD ADD ADD ADD ADD ADD ADD ADD
#define ADD64 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8 ADD8

__global__ void __launch_bounds__(256) chain(float *out, int rounds, unsigned long long mask) {
    float acc = (float)threadIdx.x, t = 1.0f;
    asm volatile("s_mov_b64 exec, %0\n s_nop 4" ::"s"(mask));
    for (int r = 0; r < rounds; ++r) {
        asm volatile(ADD64 : "+v"(acc) : "v"(t));
    }
    asm volatile("s_mov_b64 exec, -1\n s_nop 4");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int blocks = 256 * 8, rounds = 2000;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float *out = nullptr;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const unsigned long long masks[] = {~0ull, 0xFFFFFFFFFFFFull, 0xFFFFFFFFull, 0xFFFFFull, 0xFFFFull, 0xFull, 0xFFFF0000FFFFull};
    const char *names[] = {"all 64 lanes", "lanes 0 - 47", "lanes 0 - 31", "lanes 0 - 19", "lanes 0 - 15", "lanes 0 - 3", "lanes 0 - 15 and 32 - 47"};
    for (int rep = 0; rep < 2; ++rep) {
        for (int m = 0; m < 7; ++m) {
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, rounds, masks[m]);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, rounds, masks[m]);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, a, b);
            // per SIMD: 8 waves x rounds x 64 adds
            const double adds_per_simd = 8.0 * rounds * 64;
            if (rep == 1) {
                printf("%-26s %8.3f ms  %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", names[m], ms, ms * 1e6 / adds_per_simd, ms * 1e6 / adds_per_simd * 2.4);
            }
        }
    }
    return 0;
}
