// lds_atomic_chain.hip — can an LDS float atomic stand in for the exact-order chain?
//   ds_add_f32 with ALL lanes of a wave on ONE address: does the LDS add the lanes in ascending lane order, with IEEE round-to-nearest
//   fp32 adds (denormals kept), i.e. is one instruction == 64 sequential `acc += v[lane]`?  And what does it cost?
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_atomic_chain lds_atomic_chain.hip      Run on the GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

constexpr int kChunks = 7;

// layout 0: 64 lanes -> one address per instruction (chunk of 64 terms of ONE sum)
// layout 1: lanes (k = lane / 12, j = lane % 12), 5 sums x 12 terms per instruction (lanes 60..63 idle)
__global__ void chain_kernel(const float *vals, float *out, unsigned long long *ticks, int trials, int layout) {
    __shared__ float sums[8 * 32];  // sums k at sums[k * 32]: different banks
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        if (lane < 8) {
            sums[lane * 32] = 0.0f;
        }
        __syncthreads();
        const float *v = vals + (size_t)t * kChunks * 64;
        float reg[kChunks];
        for (int c = 0; c < kChunks; ++c) {
            reg[c] = v[c * 64 + lane];
        }
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int c = 0; c < kChunks; ++c) {
            if (layout == 0) {
                __hip_atomic_fetch_add(&sums[0], reg[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (lane < 60) {
                __hip_atomic_fetch_add(&sums[(lane / 12) * 32], reg[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < 8) {
            out[(size_t)t * 8 + lane] = sums[lane * 32];
        }
        if (lane == 0 && t == 0) {
            ticks[0] = t1 - t0;
        }
        __syncthreads();
    }
}

int main() {
    const int trials = 20000;
    std::mt19937 rng(7);
    std::vector<float> vals((size_t)trials * kChunks * 64);
    for (size_t i = 0; i < vals.size(); ++i) {
        const int cls = rng() % 10;
        float x;
        if (cls < 6) {
            x = std::ldexp((float)((int)(rng() % 2000001) - 1000000) / 1000000.0f, (int)(rng() % 40) - 20);  // mixed magnitudes
        } else if (cls == 6) {
            uint32_t b = rng() & 0x807FFFFFu;  // denormal
            std::memcpy(&x, &b, 4);
        } else if (cls == 7) {
            x = (rng() & 1) ? 1e8f : -1e8f;  // cancellation partners
        } else if (cls == 8) {
            x = 0.0f * ((rng() & 1) ? 1.0f : -1.0f);
        } else {
            uint32_t b = rng();  // any bit pattern but NaN / inf
            if (((b >> 23) & 0xFF) == 0xFF) b &= ~(1u << 30);
            std::memcpy(&x, &b, 4);
            if (std::fabs(x) > 1e30f) x = 1.0f;
        }
        vals[i] = x;
    }
    float *d_vals, *d_out;
    unsigned long long *d_ticks;
    hipMalloc(&d_vals, vals.size() * 4);
    hipMalloc(&d_out, (size_t)trials * 8 * 4);
    hipMalloc(&d_ticks, 8);
    hipMemcpy(d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
    for (int layout = 0; layout < 2; ++layout) {
        hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(64), 0, 0, d_vals, d_out, d_ticks, trials, layout);
        hipDeviceSynchronize();
        std::vector<float> out((size_t)trials * 8);
        unsigned long long ticks = 0;
        hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
        long asc = 0, desc = 0, other = 0, total = 0;
        for (int t = 0; t < trials; ++t) {
            const float *v = &vals[(size_t)t * kChunks * 64];
            const int n_sums = layout == 0 ? 1 : 5;
            for (int k = 0; k < n_sums; ++k) {
                volatile float a = 0.0f, d = 0.0f;
                for (int c = 0; c < kChunks; ++c) {
                    if (layout == 0) {
                        for (int l = 0; l < 64; ++l) a = a + v[c * 64 + l];
                        for (int l = 63; l >= 0; --l) d = d + v[c * 64 + l];
                    } else {
                        for (int j = 0; j < 12; ++j) a = a + v[c * 64 + k * 12 + j];
                        for (int j = 11; j >= 0; --j) d = d + v[c * 64 + k * 12 + j];
                    }
                }
                float got = out[(size_t)t * 8 + k], fa = a, fd = d;
                uint32_t gb, ab, db;
                std::memcpy(&gb, &got, 4);
                std::memcpy(&ab, &fa, 4);
                std::memcpy(&db, &fd, 4);
                ++total;
                if (gb == ab) ++asc;
                else if (gb == db) ++desc;
                else ++other;
            }
        }
        std::printf("layout %d: %ld sums: == ascending-lane order %ld, == descending only %ld, neither %ld; %d atomic instructions took %llu ticks (%.1f per instruction)\n",
                    layout, total, asc, desc, other, kChunks, ticks, (double)ticks / kChunks);
    }
    return 0;
}
