// Micro-benchmark (experiment, not product): what does ONE dependent fp32 add cost a lone wave on gfx950, in shader-clock ticks
// (s_memtime), and does the form of the instruction matter?  The exact-order sums of the trackers are such a chain.
//   hipcc --offload-arch=gfx950 -O3 -o dep_add_latency dep_add_latency.hip        Run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, unsigned long long *ticks, float seed, int lanes) {
    float acc = seed, t = seed * 0.5f, u = seed * 0.25f;
    float2 acc2 = {seed, seed}, t2 = {t, t};
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        for (int trial = 0; trial < 3; ++trial) {  // the last trial is reported (instruction cache warm)
            t0 = __builtin_amdgcn_s_memtime();
            if (MODE == 0) {
                asm volatile(REP256("v_add_f32 %0, %1, %0\n") : "+v"(acc) : "v"(t));  // acc in src1 (what the compiler emits)
            } else if (MODE == 1) {
                asm volatile(REP256("v_add_f32 %0, %0, %1\n") : "+v"(acc) : "v"(t));  // acc in src0
            } else if (MODE == 2) {
                asm volatile(REP256("v_add_f32_e64 %0, %1, %0\n") : "+v"(acc) : "v"(t));  // VOP3 encoding
            } else if (MODE == 3) {
                asm volatile(REP256("v_fma_f32 %0, %1, 1.0, %0\n") : "+v"(acc) : "v"(t));  // t * 1 + acc: the same rounding
            } else if (MODE == 4) {
                asm volatile(REP256("v_pk_add_f32 %0, %1, %0\n") : "+v"(acc2) : "v"(t2));
            } else if (MODE == 5) {
                asm volatile(REP256("v_add_f32 %0, %1, %0\n s_nop 0\n") : "+v"(acc) : "v"(t));
            } else if (MODE == 6) {
                asm volatile(REP256("v_add_f32 %0, %1, %0\n v_mul_f32 %2, %2, %1\n") : "+v"(acc), "+v"(u) : "v"(t));  // + one independent VALU
            } else if (MODE == 7) {
                asm volatile(REP256("v_add_f32 %0, %1, %0\n v_mul_f32 %2, %2, %1\n v_mul_f32 %2, %1, %2\n") : "+v"(acc), "+v"(u) : "v"(t));
            } else if (MODE == 8) {
                asm volatile(REP256("v_add_f32 %0, %1, %0\n s_nop 1\n") : "+v"(acc) : "v"(t));
            } else if (MODE == 9) {
                asm volatile(REP256("v_add_f32_dpp %0, %1, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n") : "+v"(acc) : "v"(t));
            } else if (MODE == 10) {
                asm volatile(REP256("v_add_f32 %0, %1, %0\n s_add_u32 s20, s20, 1\n") : "+v"(acc) : "v"(t) : "s20", "scc");  // + one SALU
            } else if (MODE == 11) {
                asm volatile(REP256("v_add_f64 %0, %1, %0\n") : "+v"(*(double *)&acc2) : "v"(*(double *)&t2));
            }
            asm volatile("s_nop 0" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + u + acc2.x + acc2.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ticks[0] = t1 - t0;
    }
}

template <int MODE>
void run(const char *name, float *d, unsigned long long *dt, int lanes, int blocks = 1) {
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, dt, 1.0f, lanes);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, dt, 1.0f, lanes);
    hipDeviceSynchronize();
    unsigned long long t;
    hipMemcpy(&t, dt, sizeof(t), hipMemcpyDeviceToHost);
    printf("%-58s lanes %2d blocks %5d: %6llu ticks / 256 = %5.2f per add\n", name, lanes, blocks, t, (double)t / 256.0);
}

int main() {
    float *d;
    unsigned long long *dt;
    hipMalloc(&d, sizeof(float) * 64 * 8192);
    hipMalloc(&dt, 8);
    // spin the clock up first
    for (int i = 0; i < 200; ++i) {
        hipLaunchKernelGGL(k<0>, dim3(4096), dim3(64), 0, 0, d, dt, 1.0f, 64);
    }
    hipDeviceSynchronize();
    for (int lanes : {64, 24, 5, 1}) {
        run<0>("v_add_f32 acc=src1", d, dt, lanes);
    }
    run<1>("v_add_f32 acc=src0", d, dt, 64);
    run<2>("v_add_f32_e64", d, dt, 64);
    run<3>("v_fma_f32 t,1.0,acc", d, dt, 64);
    run<4>("v_pk_add_f32", d, dt, 64);
    run<11>("v_add_f64", d, dt, 64);
    run<5>("v_add_f32 + s_nop 0", d, dt, 64);
    run<8>("v_add_f32 + s_nop 1", d, dt, 64);
    run<10>("v_add_f32 + s_add_u32", d, dt, 64);
    run<6>("v_add_f32 + 1 independent v_mul", d, dt, 64);
    run<7>("v_add_f32 + 2 independent v_mul (dependent on each other)", d, dt, 64);
    run<9>("v_add_f32_dpp (identity)", d, dt, 64);
    // the chip full of such waves: 1, 2, 4 per SIMD
    run<0>("v_add_f32, 1 wave per SIMD everywhere", d, dt, 24, 1024);
    run<0>("v_add_f32, 2 waves per SIMD everywhere", d, dt, 24, 2048);
    run<0>("v_add_f32, 4 waves per SIMD everywhere", d, dt, 24, 4096);
    return 0;
}
