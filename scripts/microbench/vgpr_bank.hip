// Micro-benchmark (experiment, not product): does the VGPR bank of the source operands change the issue rate of the Hamming
// matcher's instructions on gfx950?  Hand-written loops of 8 independent instructions with both VGPR sources in the SAME bank
// (register index mod 4) vs in different banks, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 vgpr_bank.hip -o vgpr_bank
#include <hip/hip_runtime.h>
#include <cstdio>

#define LOOP(BODY)                                  \
    asm volatile(                                   \
        "s_mov_b32 s20, %1\n"                       \
        "v_mov_b32 v40, %2\n v_mov_b32 v41, %2\n v_mov_b32 v42, %2\n v_mov_b32 v43, %2\n" \
        "v_mov_b32 v44, %2\n v_mov_b32 v45, %2\n v_mov_b32 v46, %2\n v_mov_b32 v47, %2\n" \
        "v_mov_b32 v48, %2\n v_mov_b32 v49, %2\n v_mov_b32 v50, %2\n v_mov_b32 v51, %2\n" \
        "v_mov_b32 v52, %2\n v_mov_b32 v53, %2\n v_mov_b32 v54, %2\n v_mov_b32 v55, %2\n" \
        "1:\n" BODY                                  \
        "s_sub_u32 s20, s20, 1\n"                   \
        "s_cmp_lg_u32 s20, 0\n"                     \
        "s_cbranch_scc1 1b\n"                       \
        "v_add_u32 %0, v40, v44\n"                  \
        : "=v"(r)                                   \
        : "s"(iters), "v"(seed)                     \
        : "s20", "scc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned seed) {
    unsigned r = 0;
    if (MODE == 0) {  // bcnt, accumulator and source in the same bank (40/44/48/52 are all bank 0, ...)
        LOOP("v_bcnt_u32_b32 v40, v44, v40\n v_bcnt_u32_b32 v41, v45, v41\n v_bcnt_u32_b32 v42, v46, v42\n v_bcnt_u32_b32 v43, v47, v43\n"
             "v_bcnt_u32_b32 v48, v52, v48\n v_bcnt_u32_b32 v49, v53, v49\n v_bcnt_u32_b32 v50, v54, v50\n v_bcnt_u32_b32 v51, v55, v51\n");
    } else if (MODE == 1) {  // bcnt, different banks
        LOOP("v_bcnt_u32_b32 v40, v45, v40\n v_bcnt_u32_b32 v41, v46, v41\n v_bcnt_u32_b32 v42, v47, v42\n v_bcnt_u32_b32 v43, v44, v43\n"
             "v_bcnt_u32_b32 v48, v53, v48\n v_bcnt_u32_b32 v49, v54, v49\n v_bcnt_u32_b32 v50, v55, v50\n v_bcnt_u32_b32 v51, v52, v51\n");
    } else if (MODE == 2) {  // xor, two VGPR sources in the same bank
        LOOP("v_xor_b32 v40, v44, v40\n v_xor_b32 v41, v45, v41\n v_xor_b32 v42, v46, v42\n v_xor_b32 v43, v47, v43\n"
             "v_xor_b32 v48, v52, v48\n v_xor_b32 v49, v53, v49\n v_xor_b32 v50, v54, v50\n v_xor_b32 v51, v55, v51\n");
    } else if (MODE == 3) {  // xor, different banks
        LOOP("v_xor_b32 v40, v45, v40\n v_xor_b32 v41, v46, v41\n v_xor_b32 v42, v47, v42\n v_xor_b32 v43, v44, v43\n"
             "v_xor_b32 v48, v53, v48\n v_xor_b32 v49, v54, v49\n v_xor_b32 v50, v55, v50\n v_xor_b32 v51, v52, v51\n");
    } else if (MODE == 4) {  // add_f32, same bank
        LOOP("v_add_f32 v40, v44, v40\n v_add_f32 v41, v45, v41\n v_add_f32 v42, v46, v42\n v_add_f32 v43, v47, v43\n"
             "v_add_f32 v48, v52, v48\n v_add_f32 v49, v53, v49\n v_add_f32 v50, v54, v50\n v_add_f32 v51, v55, v51\n");
    } else {  // add_f32, different banks
        LOOP("v_add_f32 v40, v45, v40\n v_add_f32 v41, v46, v41\n v_add_f32 v42, v47, v42\n v_add_f32 v43, v44, v43\n"
             "v_add_f32 v48, v53, v48\n v_add_f32 v49, v54, v49\n v_add_f32 v50, v55, v50\n v_add_f32 v51, v52, v51\n");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int waves, unsigned *d) {
    const int blocks = 256 * waves, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)waves * iters * 8);
    printf("%-36s %d waves/SIMD  %7.3f ms  %5.3f ns per wave-instruction per SIMD (%.2f cycles @2.4 GHz)\n", name, waves, ms, ns, ns * 2.4);
}

int main() {
    unsigned *d;
    (void)hipMalloc(&d, sizeof(unsigned) * 256 * 256 * 8);
    for (int w : {2, 4, 8}) {
        run<0>("v_bcnt_u32_b32 same VGPR bank", w, d);
        run<1>("v_bcnt_u32_b32 different banks", w, d);
        run<2>("v_xor_b32 same VGPR bank", w, d);
        run<3>("v_xor_b32 different banks", w, d);
        run<4>("v_add_f32 same VGPR bank", w, d);
        run<5>("v_add_f32 different banks", w, d);
    }
    return 0;
}
