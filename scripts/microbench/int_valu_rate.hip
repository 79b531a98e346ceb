// Micro-benchmark (experiment, not product): issue rate of the integer VALU instructions of the Hamming matcher's inner
// loop on gfx950 — v_xor_b32, v_bcnt_u32_b32 (popcount-accumulate), the xor+bcnt pair pattern, v_lshl_or_b32 + v_min3_u32 —
// beside v_add_f32 (independent and one dependent chain) as the yardstick, at 1 / 2 / 4 / 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 int_valu_rate.hip -o int_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned seed) {
    unsigned a[8], b[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed * 2654435761u + threadIdx.x * 40503u + i * 977u;
        b[i] = a[i] ^ (seed + i);
        f[i] = 1.0f + 1e-3f * (float)(threadIdx.x + i);
    }
    const unsigned m = seed | 0x5A5A5A5Au;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 8 independent v_xor_b32
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = a[i] ^ m;
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (MODE == 1) {  // 8 independent v_bcnt_u32_b32 accumulators
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_popcount(b[i]) + a[i];
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (MODE == 2) {  // the matcher's pattern: 4 x (xor, bcnt-accumulate) into 4 chains, twice
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned x = b[i + 4 * r] ^ m;
                    a[i] = __builtin_popcount(x) + a[i];
                }
            }
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        } else if (MODE == 3) {  // 4 x v_lshl_or_b32 + 4 x v_min3_u32 (key packing + running minimum)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned k0 = (b[i] << 16) | (unsigned)it;
                a[i] = min(a[i], min(k0, a[i + 4]));
            }
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        } else if (MODE == 4) {  // 8 independent v_add_f32
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = f[i] + 1.0000001f;
            asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
        } else if (MODE == 5) {  // 8 dependent v_add_f32 (one chain)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f[0] = f[0] + f[1 + (i & 3)];
                asm volatile("" : "+v"(f[0]));
            }
        } else {  // MODE 6: two interleaved dependent chains (8 adds each): does a second chain ride for free?
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f[0] = f[0] + f[2 + (i & 3)];
                f[1] = f[1] + f[3 + (i & 3)];
                asm volatile("" : "+v"(f[0]), "+v"(f[1]));
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i] + __float_as_uint(f[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int per_iter, unsigned *d) {
    const int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs, one 256-thread block = one wave per SIMD of a CU
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)waves_per_simd * iters * per_iter;
    const double ns = ms * 1e6 / instr_per_simd;
    printf("%-44s %d waves/SIMD  %8.3f ms  %6.3f ns = %5.2f cycles @2.4GHz per wave-instruction per SIMD (%5.2f per instruction of ONE wave)\n", name,
           waves_per_simd, ms, ns, ns * 2.4, ns * 2.4 * waves_per_simd);
}

int main() {
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 256 * 256 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_xor_b32 x8 independent", w, 8, d);
        run<1>("v_bcnt_u32_b32 x8 independent accumulators", w, 8, d);
        run<2>("4 chains x (v_xor, v_bcnt) x2 (matcher loop)", w, 16, d);
        run<3>("4 x (v_lshl_or + 2 v_min / v_min3)", w, 8, d);
        run<4>("v_add_f32 x8 independent", w, 8, d);
        run<5>("v_add_f32 x8 dependent (one chain)", w, 8, d);
        run<6>("v_add_f32 2 interleaved dependent chains", w, 16, d);
    }
    return 0;
}
