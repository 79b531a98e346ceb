// Micro-benchmark (experiment, not product): what single instructions cost a lone wave on gfx950, in shader-clock ticks.
//   hipcc --offload-arch=gfx950 -O3 -o inst_cost inst_cost.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, unsigned long long *ticks, int lanes) {
    __shared__ float4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) {
        lds[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    }
    __syncthreads();
    float acc = 0.0f;
    float4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
    const unsigned addr = (unsigned)(size_t)(lds + threadIdx.x) & 0xffffu;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        for (int trial = 0; trial < 3; ++trial) {
            t0 = __builtin_amdgcn_s_memtime();
            if (MODE == 0) {  // 64 independent ds_read_b128, one wait at the end
                asm volatile(REP16("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr));
            } else if (MODE == 1) {  // ds_read_b32
                asm volatile(REP16("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(a.x), "=&v"(b.x), "=&v"(c.x), "=&v"(d.x) : "v"(addr));
            } else if (MODE == 2) {  // s_waitcnt with nothing outstanding
                asm volatile(REP64("s_waitcnt lgkmcnt(0)\n"));
            } else if (MODE == 3) {  // one read, wait: the LDS round trip
                asm volatile(REP64("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "=&v"(a) : "v"(addr));
            } else if (MODE == 4) {  // one b32 read, wait
                asm volatile(REP64("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "=&v"(a.x) : "v"(addr));
            } else if (MODE == 5) {  // 4 adds whose sources differ + nothing else
                asm volatile(REP16("v_add_f32 %0, %1, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %4, %0\n")
                             : "+v"(acc) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
            } else if (MODE == 6) {  // a read in flight while adding: 1 read + 4 adds, data never waited for until the end
                asm volatile(REP16("ds_read_b128 %1, %2\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(acc), "=&v"(a) : "v"(addr), "v"(b.x));
            } else if (MODE == 7) {  // 1 read + 8 adds
                asm volatile(REP16("ds_read_b128 %1, %2\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n"
                                   "v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(acc), "=&v"(a) : "v"(addr), "v"(b.x));
            } else if (MODE == 8) {  // 1 b32 read + 4 adds
                asm volatile(REP16("ds_read_b32 %1, %2\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %3, %0\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(acc), "=&v"(a.x) : "v"(addr), "v"(b.x));
            } else if (MODE == 9) {  // s_barrier alone (one-wave workgroup)
                asm volatile(REP64("s_barrier\n"));
            }
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    out[threadIdx.x] = acc + a.x + b.x + c.x + d.x + a.w;
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
    }
}

template <int MODE>
void run(const char *name, int per, float *d, unsigned long long *dt, int lanes) {
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, d, dt, lanes);
        hipDeviceSynchronize();
    }
    unsigned long long t;
    hipMemcpy(&t, dt, sizeof(t), hipMemcpyDeviceToHost);
    printf("%-52s lanes %2d: %6llu ticks = %6.2f per %s\n", name, lanes, t, (double)t / per, "unit");
}

int main() {
    float *d;
    unsigned long long *dt;
    hipMalloc(&d, 4096);
    hipMalloc(&dt, 8);
    for (int lanes : {64, 24, 5}) {
        run<0>("64 x ds_read_b128 back to back", 64, d, dt, lanes);
        run<1>("64 x ds_read_b32 back to back", 64, d, dt, lanes);
        run<3>("ds_read_b128 + wait (round trip)", 64, d, dt, lanes);
        run<4>("ds_read_b32 + wait (round trip)", 64, d, dt, lanes);
        run<6>("1 ds_read_b128 + 4 dependent adds (unit = 5 instr)", 16, d, dt, lanes);
        run<7>("1 ds_read_b128 + 8 dependent adds (unit = 9 instr)", 16, d, dt, lanes);
        run<8>("1 ds_read_b32 + 4 dependent adds (unit = 5 instr)", 16, d, dt, lanes);
    }
    run<2>("s_waitcnt, nothing outstanding", 64, d, dt, 64);
    run<5>("4 dependent adds, different sources (unit = 4 adds)", 16, d, dt, 64);
    run<9>("s_barrier, one-wave workgroup", 64, d, dt, 64);
    return 0;
}
