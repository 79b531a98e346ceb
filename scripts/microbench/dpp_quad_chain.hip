// Micro-benchmark (experiment, not product): the exact-order chain with the terms handed to the accumulator through the DPP network.
//
// Today (klt_common.h chain_chunk): lane k of the chain wave reads 4 terms of ITS sum per ds_read_b128 and adds them — one read
// instruction per 4 terms, 8 - 9 cycles per term for a lone wave (one instruction of any kind per 4 cycles + LDS waits).
// Here: the 4 lanes of a QUAD all carry the same sum.  Lane i of quad q reads the float4 at term 16 j + 4 i of sum q, so ONE
// ds_read_b128 brings 16 consecutive terms of up to 16 sums into the wave, and 16 `v_add_f32_dpp acc, T, acc quad_perm:[i,i,i,i]`
// add them in order (i = 0: T.x T.y T.z T.w of quad lane 0, then lane 1's ...): every lane of the quad computes the same running
// sum.  One read instruction per 16 terms.  Variants timed: V0 today's loop, V1 quad DPP from LDS, V2 quad DPP from registers (the
// cost of the add itself), V3 row_shr (lane 15 of a 16-lane row accumulates its row's terms, one ds_read_b32 per 16 terms).
//   hipcc --offload-arch=gfx950 -O3 -o dpp_quad_chain dpp_quad_chain.hip && ./dpp_quad_chain
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

constexpr int kTerms = 448;          // per sum: 28 float4 steps of 16 terms
constexpr int kSums = 16;            // sums (rows) in LDS; V0 chains the first 5, V1 / V2 all 16 (quads), V3 four (rows)
constexpr int kPitch = kTerms + 4;   // floats between rows: quads 0..7 of a half-wave start 4 banks apart

#define QADD(i, reg) "v_add_f32_dpp %0, " reg ", %0 quad_perm:[" #i "," #i "," #i "," #i "] row_mask:0xf bank_mask:0xf\n"
#define QADD4(i) QADD(i, "%1") QADD(i, "%2") QADD(i, "%3") QADD(i, "%4")

// 16 ordered adds: the quad's four float4 in lane order, components in order
__device__ __forceinline__ float quad_step16(float acc, float4 q) {
    asm volatile("s_nop 1\n" QADD4(0) QADD4(1) QADD4(2) QADD4(3) : "+v"(acc) : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
    return acc;
}

#define RADD(n) "v_add_f32_dpp %0, %1, %0 row_shr:" #n " row_mask:0xf bank_mask:0xf\n"
// lane 15 of every row: acc += T[lane 0], T[lane 1], ... T[lane 15] of its row
__device__ __forceinline__ float row_step16(float acc, float t) {
    asm volatile("s_nop 1\n" RADD(15) RADD(14) RADD(13) RADD(12) RADD(11) RADD(10) RADD(9) RADD(8) RADD(7) RADD(6) RADD(5) RADD(4) RADD(3) RADD(2) RADD(1)
                 "v_add_f32 %0, %1, %0\n"
                 : "+v"(acc)
                 : "v"(t));
    return acc;
}

#define QSTEP(a, b, c, d) QADD(0, a) QADD(0, b) QADD(0, c) QADD(0, d) QADD(1, a) QADD(1, b) QADD(1, c) QADD(1, d) QADD(2, a) QADD(2, b) QADD(2, c) QADD(2, d) QADD(3, a) QADD(3, b) QADD(3, c) QADD(3, d)
// One 64-term chunk as ONE block: the four reads up front (16 VGPRs), then 4 x 16 adds, each group behind the wait for ITS read.
// (the 128-bit destinations are written as register quadruples through four-float vectors)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float quad_chunk64(float acc, const float *quad_row) {
    v4f q0, q1, q2, q3;
    asm volatile(
        "ds_read_b128 %1, %5\n ds_read_b128 %2, %5 offset:64\n ds_read_b128 %3, %5 offset:128\n ds_read_b128 %4, %5 offset:192\n"
        : "+v"(acc), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
        : "v"((uint32_t)(uintptr_t)quad_row)
        : "memory");
    asm volatile("s_waitcnt lgkmcnt(3)\n" QSTEP("%1", "%2", "%3", "%4") : "+v"(acc) : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w));
    asm volatile("s_waitcnt lgkmcnt(2)\n" QSTEP("%1", "%2", "%3", "%4") : "+v"(acc) : "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w));
    asm volatile("s_waitcnt lgkmcnt(1)\n" QSTEP("%1", "%2", "%3", "%4") : "+v"(acc) : "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w));
    asm volatile("s_waitcnt lgkmcnt(0)\n" QSTEP("%1", "%2", "%3", "%4") : "+v"(acc) : "v"(q3.x), "v"(q3.y), "v"(q3.z), "v"(q3.w));
    return acc;
}

template <int V>
__global__ void __launch_bounds__(64) k(const float *terms, float *out, unsigned long long *ticks, int trials) {
    __shared__ float4 lds4[kSums * kPitch / 4];
    float *lds = reinterpret_cast<float *>(lds4);
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        for (int i = lane; i < kSums * kTerms; i += 64) {
            lds[(i / kTerms) * kPitch + (i % kTerms)] = terms[(size_t)t * kSums * kTerms + i];
        }
        __syncthreads();
        float acc = 0.0f;
        unsigned long long t0, t1;
        if (V == 0) {
            const float4 *mine = reinterpret_cast<const float4 *>(lds + (lane < kSums ? lane : 0) * kPitch);
            float4 qa[4], qb[4];
            t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int d = 3; d >= 0; --d) {
                qa[d] = mine[d];
            }
#pragma unroll 1
            for (int r = 0; r < kTerms / 16; r += 2) {
#pragma unroll
                for (int d = 3; d >= 0; --d) {
                    qb[d] = mine[(r + 1) * 4 + d];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    acc += qa[d].x;
                    acc += qa[d].y;
                    acc += qa[d].z;
                    acc += qa[d].w;
                }
                __builtin_amdgcn_sched_barrier(0);
                const int ahead = r + 2 < kTerms / 16 ? r + 2 : 0;
#pragma unroll
                for (int d = 3; d >= 0; --d) {
                    qa[d] = mine[ahead * 4 + d];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    acc += qb[d].x;
                    acc += qb[d].y;
                    acc += qb[d].z;
                    acc += qb[d].w;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
            if (lane < kSums) {
                out[(size_t)t * kSums + lane] = acc;
            }
        } else if (V == 1) {
            // quad q = lane / 4 carries sum q; quad lane i reads terms 16 j + 4 i .. + 3
            const float4 *mine = reinterpret_cast<const float4 *>(lds + (lane >> 2) * kPitch) + (lane & 3);
            t0 = __builtin_amdgcn_s_memtime();
            float4 qa = mine[0], qb = mine[4];
#pragma unroll 1
            for (int j = 0; j < kTerms / 16; j += 2) {
                acc = quad_step16(acc, qa);
                qa = mine[(j + 2 < kTerms / 16 ? j + 2 : 0) * 4];
                acc = quad_step16(acc, qb);
                qb = mine[(j + 3 < kTerms / 16 ? j + 3 : 0) * 4];
            }
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
            if ((lane & 3) == 0) {
                out[(size_t)t * kSums + (lane >> 2)] = acc;
            }
        } else if (V == 2) {
            const float4 *mine = reinterpret_cast<const float4 *>(lds + (lane >> 2) * kPitch) + (lane & 3);
            float4 q[kTerms / 16];
#pragma unroll
            for (int j = 0; j < kTerms / 16; ++j) {
                q[j] = mine[j * 4];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int j = 0; j < kTerms / 16; ++j) {
                acc = quad_step16(acc, q[j]);
            }
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
            if ((lane & 3) == 0) {
                out[(size_t)t * kSums + (lane >> 2)] = acc;
            }
        } else if (V == 4) {
            const float *mine = lds + (lane >> 2) * kPitch + 4 * (lane & 3);
            t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
            for (int ch = 0; ch < kTerms / 64; ++ch) {
                acc = quad_chunk64(acc, mine + 64 * ch);
            }
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
            if ((lane & 3) == 0) {
                out[(size_t)t * kSums + (lane >> 2)] = acc;
            }
        } else {
            // row r = lane / 16 carries sum r in its lane 15; lane i of the row reads term 16 j + i
            const float *mine = lds + (lane >> 4) * kPitch + (lane & 15);
            t0 = __builtin_amdgcn_s_memtime();
            float ta = mine[0], tb = mine[16];
#pragma unroll 1
            for (int j = 0; j < kTerms / 16; j += 2) {
                acc = row_step16(acc, ta);
                ta = mine[(j + 2 < kTerms / 16 ? j + 2 : 0) * 16];
                acc = row_step16(acc, tb);
                tb = mine[(j + 3 < kTerms / 16 ? j + 3 : 0) * 16];
            }
            asm volatile("" : "+v"(acc));
            t1 = __builtin_amdgcn_s_memtime();
            if ((lane & 15) == 15) {
                out[(size_t)t * kSums + (lane >> 4)] = acc;
            }
        }
        if (lane == 0 && t == 0) {
            ticks[0] = t1 - t0;
        }
        __syncthreads();
    }
}

int main() {
    const int trials = 4000;
    std::mt19937 rng(11);
    std::vector<float> terms((size_t)trials * kSums * kTerms);
    for (auto &x : terms) {
        const int cls = rng() % 10;
        if (cls < 7) {
            x = std::ldexp((float)((int)(rng() % 2000001) - 1000000) / 1000000.0f, (int)(rng() % 30) - 15);
        } else if (cls < 8) {
            x = 0.0f;
        } else if (cls < 9) {
            x = std::ldexp(1.0f, -130 - (int)(rng() % 15));  // denormals
        } else {
            x = -std::ldexp((float)(rng() % 1000) / 1000.0f, (int)(rng() % 60) - 30);
        }
    }
    float *d_terms, *d_out;
    unsigned long long *d_ticks;
    hipMalloc(&d_terms, terms.size() * 4);
    hipMalloc(&d_out, (size_t)trials * kSums * 4);
    hipMalloc(&d_ticks, 8);
    hipMemcpy(d_terms, terms.data(), terms.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ref((size_t)trials * kSums);
    for (int t = 0; t < trials; ++t) {
        for (int s = 0; s < kSums; ++s) {
            volatile float acc = 0.0f;
            const float *row = &terms[((size_t)t * kSums + s) * kTerms];
            for (int i = 0; i < kTerms; ++i) {
                acc = acc + row[i];
            }
            ref[(size_t)t * kSums + s] = acc;
        }
    }
    int rc = 0;
    const char *names[5] = {"V0 plain chain, lanes 0..15, ds_read_b128 per 4 terms", "V1 quad DPP chain from LDS, ds_read_b128 per 16 terms",
                            "V2 quad DPP adds from registers (no reads)", "V3 row_shr DPP chain, ds_read_b32 per 16 terms",
                            "V4 quad DPP chain, one asm block per 64-term chunk (4 reads up front)"};
    const int sums[5] = {kSums, kSums, kSums, 4, kSums};
    for (int v = 0; v < 5; ++v) {
        hipMemset(d_out, 0, (size_t)trials * kSums * 4);
        for (int rep = 0; rep < 3; ++rep) {
            switch (v) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(64), 0, 0, d_terms, d_out, d_ticks, trials); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, d_terms, d_out, d_ticks, trials); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(64), 0, 0, d_terms, d_out, d_ticks, trials); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(64), 0, 0, d_terms, d_out, d_ticks, trials); break;
                default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(64), 0, 0, d_terms, d_out, d_ticks, trials); break;
            }
            hipDeviceSynchronize();
        }
        std::vector<float> out((size_t)trials * kSums);
        hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long ticks;
        hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int t = 0; t < trials; ++t) {
            for (int s = 0; s < sums[v]; ++s) {
                if (std::memcmp(&ref[(size_t)t * kSums + s], &out[(size_t)t * kSums + s], 4) != 0) {
                    ++bad;
                }
            }
        }
        printf("%-58s: %zu of %d sums differ; %llu ticks / %d terms = %.2f per term\n", names[v], bad, trials * sums[v], ticks, kTerms, (double)ticks / kTerms);
        rc |= bad != 0;
    }
    return rc;
}
