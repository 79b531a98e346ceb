// Micro-benchmark (experiment, not product): v_add_f32_dpp acc, q, acc row_shl:N with the ACCUMULATOR (src1, not the DPP operand)
// written by the previous VALU instruction and no wait state in between.  LLVM's hazard recogniser puts an s_nop between them when
// they come as separate statements (it treats every VGPR operand of a DPP instruction alike); the ISA rule is about the operand the
// DPP network moves (src0).  Is the result right without the s_nop, and what does the chain cost then?
//   hipcc --offload-arch=gfx950 -O3 -o dpp_chain_hazard dpp_chain_hazard.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

// lanes 0..4 of a row: chain lanes of sums 0..4; lanes 5..9 and 10..14 hold the NEXT two float4 of the same sums.
// q: this lane's four terms.  Returns acc after 12 terms (own four, then the four of lane + 5, then the four of lane + 10).
__device__ __forceinline__ float step12(float acc, float4 q) {
    asm volatile(
        "v_add_f32 %0, %1, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %4, %0\n"
        "v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %2, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %3, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %4, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %2, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %0, %3, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %4, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n"
        : "+v"(acc)
        : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
    return acc;
}

constexpr int kTerms = 444;  // 37 steps of 12

// terms: [trial][sum 0..4][kTerms]
__global__ void __launch_bounds__(64) k(const float *terms, float *out, unsigned long long *ticks, int trials) {
    const int lane = threadIdx.x, in_row = lane & 15, g = in_row / 5, j = in_row - 5 * g;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        const float *mine = terms + ((size_t)t * 5 + (j < 5 ? j : 0)) * kTerms + 4 * (g < 3 ? g : 0);
        float acc = 0.0f;
        float4 q[kTerms / 12];
        for (int s = 0; s < kTerms / 12; ++s) {
            q[s] = make_float4(mine[12 * s], mine[12 * s + 1], mine[12 * s + 2], mine[12 * s + 3]);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int s = 0; s < kTerms / 12; ++s) {
            acc = step12(acc, q[s]);
        }
        asm volatile("" : "+v"(acc));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < 5) {
            out[(size_t)t * 5 + lane] = acc;
        }
        if (lane == 0 && t == 0) {
            ticks[0] = t1 - t0;
        }
    }
}

int main() {
    const int trials = 20000;
    std::mt19937 rng(11);
    std::vector<float> terms((size_t)trials * 5 * kTerms);
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
    hipMalloc(&d_out, (size_t)trials * 5 * 4);
    hipMalloc(&d_ticks, 8);
    hipMemcpy(d_terms, terms.data(), terms.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(512), dim3(64), 0, 0, d_terms, d_out, d_ticks, trials);
    hipDeviceSynchronize();
    std::vector<float> out((size_t)trials * 5);
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long ticks;
    hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int t = 0; t < trials; ++t) {
        for (int s = 0; s < 5; ++s) {
            volatile float acc = 0.0f;
            const float *row = &terms[((size_t)t * 5 + s) * kTerms];
            for (int i = 0; i < kTerms; ++i) {
                acc = acc + row[i];
            }
            const float ref = acc;
            if (std::memcmp(&ref, &out[(size_t)t * 5 + s], 4) != 0) {
                if (bad < 5) {
                    printf("mismatch trial %d sum %d: %a vs %a\n", t, s, ref, out[(size_t)t * 5 + s]);
                }
                ++bad;
            }
        }
    }
    printf("dpp chain without wait states: %zu of %d sums differ from the sequential sum; %llu ticks / %d terms = %.2f per term\n", bad, trials * 5, ticks, kTerms,
           (double)ticks / kTerms);
    return bad != 0;
}
