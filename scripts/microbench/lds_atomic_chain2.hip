// lds_atomic_chain2.hip — the planned consumer: 5 rows of P products in an LDS stream; lanes (k = lane / 12, j = lane % 12) read
// row k, position 12 g + j and ds_add_f32 it to sums[k], for g = 0 .. ceil(P / 12) - 1.  Is sums[k] == the sequential float sum
// of row k in ascending position order, bit for bit (zeros, denormals, huge / tiny magnitudes, cancellations)?  Cost per group?
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

constexpr int kP = 441, kRows = 5, kPitch = 448;

__global__ void consumer_kernel(const float *vals, float *out, unsigned long long *ticks, int trials) {
    __shared__ float ring[kRows * kPitch];
    __shared__ float sums[8];
    const int lane = threadIdx.x;
    const int k = lane / 12, j = lane - 12 * k;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        for (int i = lane; i < kRows * kPitch; i += 64) {
            const int r = i / kPitch, c = i - r * kPitch;
            ring[i] = c < kP ? vals[((size_t)t * kRows + r) * kP + c] : 0.0f;
        }
        if (lane < 8) {
            sums[lane] = 0.0f;
        }
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const int groups = (kP + 11) / 12;
        for (int g0 = 0; g0 < groups; g0 += 8) {  // eight reads in flight, then eight atomics back to back
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pos = 12 * (g0 + u) + j;
                v[u] = (lane < 60 && pos < kP) ? ring[k * kPitch + pos] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pos = 12 * (g0 + u) + j;
                if (lane < 60 && pos < kP && g0 + u < groups) {
                    __hip_atomic_fetch_add(&sums[k], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane < kRows) {
            out[(size_t)t * 8 + lane] = sums[lane];
        }
        if (lane == 0 && t == 0) {
            ticks[0] = t1 - t0;
        }
        __syncthreads();
    }
}

int main() {
    const int trials = 8000;
    std::mt19937 rng(11);
    std::vector<float> vals((size_t)trials * kRows * kP);
    for (size_t i = 0; i < vals.size(); ++i) {
        const int cls = rng() % 12;
        float x;
        if (cls < 6) {
            x = std::ldexp((float)((int)(rng() % 2000001) - 1000000) / 1000000.0f, (int)(rng() % 60) - 30);
        } else if (cls == 6) {
            uint32_t b = rng() & 0x807FFFFFu;
            std::memcpy(&x, &b, 4);
        } else if (cls == 7) {
            x = (rng() & 1) ? 1e8f : -1e8f;
        } else if (cls == 8) {
            x = 0.0f * ((rng() & 1) ? 1.0f : -1.0f);
        } else if (cls == 9) {
            x = (rng() & 1) ? 3.0e38f : -3.0e38f;  // overflow to +-inf and inf - inf = NaN somewhere in the row
        } else {
            x = (float)((int)(rng() % 65536) - 32768) * (float)((int)(rng() % 65536) - 32768);  // products of small integers: what fx * fx looks like
        }
        vals[i] = x;
    }
    float *d_vals, *d_out;
    unsigned long long *d_ticks;
    (void)hipMalloc(&d_vals, vals.size() * 4);
    (void)hipMalloc(&d_out, (size_t)trials * 8 * 4);
    (void)hipMalloc(&d_ticks, 8);
    (void)hipMemcpy(d_vals, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(consumer_kernel, dim3(256), dim3(64), 0, 0, d_vals, d_out, d_ticks, trials);
    (void)hipDeviceSynchronize();
    std::vector<float> out((size_t)trials * 8);
    unsigned long long ticks = 0;
    (void)hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
    long same = 0, both_nan = 0, diff = 0;
    for (int t = 0; t < trials; ++t) {
        for (int k = 0; k < kRows; ++k) {
            volatile float a = 0.0f;
            const float *v = &vals[((size_t)t * kRows + k) * kP];
            for (int i = 0; i < kP; ++i) a = a + v[i];
            const float got = out[(size_t)t * 8 + k], want = a;
            uint32_t gb, wb;
            std::memcpy(&gb, &got, 4);
            std::memcpy(&wb, &want, 4);
            if (gb == wb) ++same;
            else if (std::isnan(got) && std::isnan(want)) ++both_nan;
            else ++diff;
        }
    }
    std::printf("%d rows of %d terms: bit-identical to the sequential sum %ld, both NaN (payload differs) %ld, DIFFERENT %ld; %d groups took %llu ticks = %.1f per group\n",
                trials * kRows, kP, same, both_nan, diff, (kP + 11) / 12, ticks, (double)ticks / ((kP + 11) / 12));
    return diff ? 1 : 0;
}
