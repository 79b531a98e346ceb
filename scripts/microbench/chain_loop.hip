// Micro-benchmark (experiment, not product): ticks per term of the exact-order chain loop — K lanes of one wave each add the terms of
// one sum strictly left to right, the terms coming from LDS four at a time (ds_read_b128) — for several prefetch shapes.
// A lone wave issues ONE instruction of any kind per 4 cycles (dep_add_latency.hip), so the floor is 4 x (1 add + 1/4 read) = 5
// cycles per term plus waits; what is lost to LDS latency when the reads run only one round of 16 terms ahead?
//   hipcc --offload-arch=gfx950 -O3 -o chain_loop chain_loop.hip
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kTerms = 1024;       // per sum
constexpr int kStride4 = 25;       // float4 between consecutive reads of one lane (the affine product groups: 100 floats)

#ifndef REVERSED
#define REVERSED 0
#endif
template <int R>
__device__ __forceinline__ void load(float4 (&q)[R], const float4 *t) {
#pragma unroll
    for (int d = 0; d < R; ++d) {
        const int e = REVERSED ? R - 1 - d : d;  // reversed: the first-consumed float4 is issued LAST, so ONE s_waitcnt covers the round
        q[e] = t[e * kStride4];
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int R>
__device__ __forceinline__ float consume(float acc, const float4 (&q)[R]) {
#pragma unroll
    for (int d = 0; d < R; ++d) {
        acc += q[d].x;
        acc += q[d].y;
        acc += q[d].z;
        acc += q[d].w;
    }
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

// DEPTH register sets of R float4 each, reads (DEPTH - 1) rounds ahead
template <int R, int DEPTH>
__device__ __forceinline__ float chain(const float4 *t, int rounds, float acc) {
    float4 q[DEPTH][R];
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) {
        load<R>(q[s], t + s * R * kStride4);
    }
#pragma nounroll
    for (int r = 0; r < rounds; r += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int ahead = r + s + DEPTH - 1;
            load<R>(q[(s + DEPTH - 1) % DEPTH], t + (ahead < rounds ? ahead : 0) * R * kStride4);
            acc = consume<R>(acc, q[s]);
        }
    }
    return acc;
}

// The same chain with helper lanes: the K <= 16 / G chain lanes of each 16-lane row are followed by G - 1 groups of helper lanes that
// read the NEXT float4 of the same sums; the chain lanes add their own four terms, then the helpers' through DPP row shifts
// (v_add_f32_dpp takes src0 from lane + shift of the same row): 4 G terms per ds_read_b128 instead of 4.
template <int SHIFT>
__device__ __forceinline__ float add_from(float acc, float q) {
    if (SHIFT == 0) {
        return acc + q;
    }
    if (SHIFT == 5) {
        asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q));
    } else if (SHIFT == 6) {
        asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q));
    } else if (SHIFT == 10) {
        asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q));
    } else if (SHIFT == 8) {
        asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q));
    }
    return acc;
}
template <int R, int G, int KR>
__device__ __forceinline__ float consume_g(float acc, const float4 (&q)[R]) {
#pragma unroll
    for (int d = 0; d < R; ++d) {
        acc = add_from<0>(add_from<0>(add_from<0>(add_from<0>(acc, q[d].x), q[d].y), q[d].z), q[d].w);
        if (G >= 2) {
            acc = add_from<KR>(add_from<KR>(add_from<KR>(add_from<KR>(acc, q[d].x), q[d].y), q[d].z), q[d].w);
        }
        if (G >= 3) {
            acc = add_from<2 * KR>(add_from<2 * KR>(add_from<2 * KR>(add_from<2 * KR>(acc, q[d].x), q[d].y), q[d].z), q[d].w);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}
template <int R, int G>
__device__ __forceinline__ void load_g(float4 (&q)[R], const float4 *t) {
#pragma unroll
    for (int d = 0; d < R; ++d) {
        q[d] = t[d * G * kStride4];
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int R, int G, int KR>
__global__ void __launch_bounds__(64) kg(float *out, unsigned long long *ticks, int rows) {
    extern __shared__ float4 lds[];
    const int n4 = (kTerms / 4 + 8 * G * R) * kStride4;
    for (int i = threadIdx.x; i < n4; i += 64) {
        lds[i] = make_float4(1.0f + i, 0.5f, 0.25f, 0.125f);
    }
    __syncthreads();
    float acc = 0.0f;
    unsigned long long t0 = 0, t1 = 0;
    const int lane = threadIdx.x, row = lane >> 4, in_row = lane & 15, g = in_row / KR, j = in_row - g * KR;
    if (row < rows && g < G) {
        const float4 *t = lds + (row * KR + j) + g * kStride4;  // sum row * KR + j, pixel group g of every G
        const int rounds = kTerms / (4 * G * R);
        for (int trial = 0; trial < 3; ++trial) {
            t0 = __builtin_amdgcn_s_memtime();
            float4 qa[R], qb[R];
            load_g<R, G>(qa, t);
#pragma nounroll
            for (int r = 0; r < rounds; r += 2) {
                load_g<R, G>(qb, t + (r + 1) * R * G * kStride4);
                acc = consume_g<R, G, KR>(acc, qa);
                load_g<R, G>(qa, t + (r + 2 < rounds ? r + 2 : 0) * R * G * kStride4);
                acc = consume_g<R, G, KR>(acc, qb);
            }
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
    }
}
template <int R, int G, int KR>
void run_g(float *d, unsigned long long *dt, int rows) {
    const size_t lds = sizeof(float4) * (kTerms / 4 + 8 * G * R) * kStride4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kg<R, G, KR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL((kg<R, G, KR>), dim3(1), dim3(64), lds, 0, d, dt, rows);
        hipDeviceSynchronize();
    }
    unsigned long long t;
    hipMemcpy(&t, dt, sizeof(t), hipMemcpyDeviceToHost);
    printf("helper lanes: %d groups of %d chain lanes per row, %d rows, round %d reads (%2d terms): %6llu ticks / %d terms = %5.2f per term\n", G, KR, rows, R,
           4 * R * G, t, kTerms, (double)t / kTerms);
}

template <int R, int DEPTH>
__global__ void __launch_bounds__(64) k(float *out, unsigned long long *ticks, int lanes) {
    extern __shared__ float4 lds[];
    const int n4 = (kTerms / 4 + 8) * kStride4;
    for (int i = threadIdx.x; i < n4; i += 64) {
        lds[i] = make_float4(1.0f + i, 0.5f, 0.25f, 0.125f);
    }
    __syncthreads();
    float acc = 0.0f;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {
        for (int trial = 0; trial < 3; ++trial) {
            t0 = __builtin_amdgcn_s_memtime();
            acc = chain<R, DEPTH>(lds + threadIdx.x, kTerms / (4 * R), acc);
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
        }
    }
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
    }
}

template <int R, int DEPTH>
void run(float *d, unsigned long long *dt, int lanes) {
    const size_t lds = sizeof(float4) * (kTerms / 4 + 8) * kStride4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<R, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL((k<R, DEPTH>), dim3(1), dim3(64), lds, 0, d, dt, lanes);
        hipDeviceSynchronize();
    }
    unsigned long long t;
    hipMemcpy(&t, dt, sizeof(t), hipMemcpyDeviceToHost);
    printf("round %d float4 (%2d terms), %d register sets, %2d lanes: %6llu ticks / %d terms = %5.2f per term\n", R, 4 * R, DEPTH, lanes, t, kTerms, (double)t / kTerms);
}

int main() {
    float *d;
    unsigned long long *dt;
    hipMalloc(&d, 4096);
    hipMalloc(&dt, 8);
    for (int lanes : {24, 5}) {
        run<4, 2>(d, dt, lanes);
        run<4, 3>(d, dt, lanes);
        run<4, 4>(d, dt, lanes);
        run<8, 2>(d, dt, lanes);
        run<8, 3>(d, dt, lanes);
        run<2, 4>(d, dt, lanes);
        run<2, 8>(d, dt, lanes);
        run<1, 8>(d, dt, lanes);
    }
    run_g<4, 1, 6>(d, dt, 4);
    run_g<4, 2, 6>(d, dt, 4);
    run_g<2, 2, 6>(d, dt, 4);
    run_g<8, 2, 6>(d, dt, 4);
    run_g<4, 2, 8>(d, dt, 4);
    run_g<4, 3, 5>(d, dt, 1);
    run_g<2, 3, 5>(d, dt, 1);
    run_g<4, 3, 5>(d, dt, 2);
    return 0;
}
