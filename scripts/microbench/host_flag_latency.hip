// Micro-benchmark (experiment, not product): what does the END of a host-buffer call cost?  A call is launch -> kernel ->
// "the host knows the results are in its memory".  Two ways to learn that: hipStreamSynchronize (the command processor's
// end-of-kernel release + completion signal + the runtime's wait), or a word in pinned host memory that the LAST workgroup
// of the kernel stores after every workgroup's results have been pushed out (two-level arrival counters: a device-scope
// atomic to ONE address serialises at ~11 ns chip-wide), polled by the calling thread.
//   hipcc --offload-arch=gfx950 -O3 -o host_flag_latency host_flag_latency.hip      Run on the GPU box.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

constexpr int kBins = 64;

// Every workgroup "tracks a feature": reads 8 B from host memory, spins for `ticks` of the 100 MHz wall clock, writes 9 B
// to host memory.  With `flag` != nullptr the last workgroup to arrive stores `seq` there.
__global__ void __launch_bounds__(128) work_kernel(const float2 *in, float2 *out, uint8_t *status, uint32_t *bins, uint32_t *top,
                                                   uint32_t *flag, uint32_t seq, uint32_t ticks) {
    const float2 v = in[blockIdx.x];
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0) {
        // results leave as system-scope stores (never parked in this XCD's L2): no fence, hence no L2 write-back, is needed
        // to push them out -- a __threadfence_system() per workgroup cost 25 ns EACH chip-wide (2 000 workgroups: +50 us)
        __hip_atomic_store(&out[blockIdx.x].x, v.x + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&out[blockIdx.x].y, v.y + (float)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&status[blockIdx.x], (uint8_t)(seq & 0xFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (flag != nullptr && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the stores above have been acknowledged before this workgroup is counted
        const uint32_t bin = blockIdx.x % kBins;
        const uint32_t members = (gridDim.x - bin + kBins - 1) / kBins;
        if (__hip_atomic_fetch_add(&bins[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            __hip_atomic_store(&bins[bin], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t live_bins = gridDim.x < (unsigned)kBins ? gridDim.x : (unsigned)kBins;
            if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == live_bins - 1u) {
                __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double median(std::vector<double> &v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    hipStream_t stream;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    const int max_n = 16384;
    float2 *h_in, *h_out;
    uint8_t *h_status;
    uint32_t *h_flag;
    CHECK(hipHostMalloc(&h_in, sizeof(float2) * max_n, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_out, sizeof(float2) * max_n, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_status, max_n, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
    float2 *m_in, *m_out;
    uint8_t *m_status;
    uint32_t *m_flag;
    CHECK(hipHostGetDevicePointer((void **)&m_in, h_in, 0));
    CHECK(hipHostGetDevicePointer((void **)&m_out, h_out, 0));
    CHECK(hipHostGetDevicePointer((void **)&m_status, h_status, 0));
    CHECK(hipHostGetDevicePointer((void **)&m_flag, h_flag, 0));
    uint32_t *d_bins, *d_top;
    CHECK(hipMalloc(&d_bins, sizeof(uint32_t) * kBins));
    CHECK(hipMalloc(&d_top, sizeof(uint32_t)));
    CHECK(hipMemset(d_bins, 0, sizeof(uint32_t) * kBins));
    CHECK(hipMemset(d_top, 0, sizeof(uint32_t)));
    for (int i = 0; i < max_n; ++i) {
        h_in[i] = make_float2((float)i, 0.0f);
    }
    *h_flag = 0;
    CHECK(hipDeviceSynchronize());

    const int sizes[] = {1, 200, 2000, 10000};
    const uint32_t tick_list[] = {0, 1700, 3900};  // 0 / 17 / 39 us of "tracking" per workgroup (100 MHz wall clock)
    uint32_t seq = 0;
    printf("%8s %8s | %12s %12s %12s | %10s\n", "n", "work_us", "sync_us", "flag_us", "flag+sync_us", "bad");
    for (int n : sizes) {
        for (uint32_t ticks : tick_list) {
            std::vector<double> t_sync, t_flag, t_both;
            long bad = 0;
            for (int mode = 0; mode < 3; ++mode) {
                for (int r = 0; r < reps + 50; ++r) {
                    ++seq;
                    const double t0 = now_us();
                    work_kernel<<<n, 128, 0, stream>>>(m_in, m_out, m_status, d_bins, d_top, mode == 0 ? nullptr : m_flag, seq, ticks);
                    if (mode == 0) {
                        CHECK(hipStreamSynchronize(stream));
                    } else {
                        long spins = 0;
                        while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) {
                            __builtin_ia32_pause();
                            if (++spins > 400000000L) {
                                fprintf(stderr, "flag never arrived (n %d seq %u)\n", n, seq);
                                return 2;
                            }
                        }
                        // the results must be there the moment the flag is
                        for (int i = 0; i < n; i += (n > 64 ? n / 64 : 1)) {
                            bad += (h_out[i].y != (float)seq) || (h_status[i] != (uint8_t)(seq & 0xFF));
                        }
                        bad += (h_out[n - 1].y != (float)seq) || (h_status[n - 1] != (uint8_t)(seq & 0xFF));
                        if (mode == 2) {
                            CHECK(hipStreamSynchronize(stream));
                        }
                    }
                    const double t1 = now_us();
                    if (r >= 50) {
                        (mode == 0 ? t_sync : mode == 1 ? t_flag : t_both).push_back(t1 - t0);
                    }
                }
                CHECK(hipStreamSynchronize(stream));
            }
            printf("%8d %8.1f | %12.2f %12.2f %12.2f | %10ld\n", n, ticks / 100.0, median(t_sync), median(t_flag), median(t_both), bad);
            fflush(stdout);
        }
    }
    return 0;
}
