// pyramid_kernels.hip — image-side helpers for gfx950 (MI355X).
//
//  * pyramid_downsample: ImagePyramid::CreateImagePyramid's level step (call sites
//    test/test_optical_flow.cpp:70-71): dst = truncating 2x2 box mean of src, (rows/2, cols/2).
//    A pure streaming kernel (1.25 B moved per source byte): every thread produces 4 adjacent
//    output pixels from two 8-byte row segments, so a wave reads 2 x 512 contiguous bytes.
//  * extract_patch: OpticalFlow::ExtractExtendPatchInReferenceImage (optical_flow.cpp:49-102),
//    exposed because the reference makes it a public method; one wave per call.
#include <stdlib.h>
#include "ftk_device.h"

#include <limits.h>
#include <math.h>

namespace ftk {
namespace {

constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock) downsample_kernel(const uint8_t *__restrict__ src, int src_cols, uint8_t *__restrict__ dst, int dst_rows,
                                                            int dst_cols) {
    // one thread -> up to 4 consecutive output pixels of one output row
    const int groups_per_row = (dst_cols + 3) >> 2;
    const long long gid = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long total = (long long)groups_per_row * dst_rows;
    if (gid >= total) {
        return;
    }
    const int r = (int)(gid / groups_per_row);
    const int c0 = (int)(gid - (long long)r * groups_per_row) << 2;
    const uint8_t *top = src + (long long)(2 * r) * src_cols + 2 * c0;
    const uint8_t *bottom = top + src_cols;
    uint8_t *out = dst + (long long)r * dst_cols + c0;
    const int n = min(4, dst_cols - c0);
    const bool aligned8 = ((reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(bottom)) & 7) == 0;
    if (n == 4 && aligned8) {
        const uint2 t = *reinterpret_cast<const uint2 *>(top);
        const uint2 b = *reinterpret_cast<const uint2 *>(bottom);
        const unsigned tw[2] = {t.x, t.y}, bw[2] = {b.x, b.y};
        unsigned packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned tv = tw[k >> 1] >> ((k & 1) * 16);
            const unsigned bv = bw[k >> 1] >> ((k & 1) * 16);
            const unsigned sum = (tv & 0xFF) + ((tv >> 8) & 0xFF) + (bv & 0xFF) + ((bv >> 8) & 0xFF);
            packed |= (sum >> 2) << (8 * k);
        }
        if ((reinterpret_cast<uintptr_t>(out) & 3) == 0) {
            *reinterpret_cast<unsigned *>(out) = packed;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                out[k] = (uint8_t)(packed >> (8 * k));
            }
        }
    } else {
        for (int k = 0; k < n; ++k) {
            const unsigned sum = (unsigned)top[2 * k] + top[2 * k + 1] + bottom[2 * k] + bottom[2 * k + 1];
            out[k] = (uint8_t)(sum >> 2);
        }
    }
}

__device__ __forceinline__ int f2i(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : INT_MIN; }
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

__global__ void __launch_bounds__(64) extract_patch_kernel(DevImage ref, float u, float v, int ex_rows, int ex_cols, float *patch, uint8_t *valid,
                                                           uint32_t *count) {
    const int lane = threadIdx.x;
    const float int_row = floorf(v);
    const float int_col = floorf(u);
    const float dec_row = v - int_row;
    const float dec_col = u - int_col;
    const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
    const float w_tr = (1.0f - dec_row) * dec_col;
    const float w_bl = dec_row * (1.0f - dec_col);
    const float w_br = dec_row * dec_col;
    const int min_row = wadd(f2i(int_row), -(ex_rows / 2));
    const int min_col = wadd(f2i(int_col), -(ex_cols / 2));
    const int total = ex_rows * ex_cols;
    uint32_t cnt = 0;
    for (int base = 0; base < total; base += 64) {
        const int e = base + lane;
        bool ok = false;
        if (e < total) {
            const int erow = e / ex_cols;
            const int ecol = e - erow * ex_cols;
            const int row = wadd(min_row, erow);
            const int col = wadd(min_col, ecol);
            ok = !(row < 0 || row > ref.rows - 2 || col < 0 || col > ref.cols - 2);
            float value = 0.0f;
            if (ok) {
                const uint8_t *q = ref.data + (long long)row * ref.cols + col;
                value = w_tl * (float)q[0] + w_tr * (float)q[1] + w_bl * (float)q[ref.cols] + w_br * (float)q[ref.cols + 1];
            }
            patch[e] = value;
            valid[e] = ok ? 1 : 0;
        }
        cnt += (uint32_t)__popcll(__ballot(ok));
    }
    if (lane == 0) {
        *count = cnt;
    }
}


// Multi-GPU result exchange (ftk_comm.cpp): the gathered buffer holds `world` packed shards ([u, v] pairs of `cap` features, then
// `cap` status bytes, `shard_bytes` apart); feature i of the global list lives in the shard of the rank whose block contains it
// (blocks as ftk_shard_bounds: the first n % world ranks hold one feature more).
__global__ void __launch_bounds__(kBlock) unpack_klt_shards_kernel(const uint8_t *__restrict__ gathered, int n, int world, int cap, long long shard_bytes,
                                                                   float2 *__restrict__ uv_out, uint8_t *__restrict__ status_out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) {
        return;
    }
    const int base = n / world, extra = n % world;
    const int big = extra * (base + 1);  // features held by the ranks with base + 1 of them
    int rank, local;
    if (i < big) {
        rank = i / (base + 1);
        local = i - rank * (base + 1);
    } else {
        rank = extra + (i - big) / (base > 0 ? base : 1);
        local = (i - big) - (rank - extra) * base;
    }
    const uint8_t *shard = gathered + shard_bytes * rank;
    uv_out[i] = reinterpret_cast<const float2 *>(shard)[local];
    status_out[i] = shard[(long long)cap * 8 + local];
}


// Launch order of the next tracker call: feature indices by iteration count of the previous call, longest first — a counting
// sort over min(count, 255) in ONE workgroup (wave_bin_claim keeps equal keys from serialising on one LDS address).  The workgroup walks the list in index
// order, so inside a bin the features keep list order at 1 024-feature granularity (neighbours in the list are usually
// neighbours in the image).  When the counts have no tail — the largest is within 1.5 x the mean of the tracked features —
// the order would buy nothing and the identity is written instead.  Any permutation yields the same results; only the
// schedule differs.
constexpr int kOrderBlock = 1024;

// A slot in its bin for every active lane.  One round of "the lowest active lane's bin: ONE LDS atomic for all the lanes that
// share it" — iteration counts cluster, and a call whose features all took 5 iterations would otherwise serialise thousands
// of atomics on one address — then one atomic per remaining lane (counts spread over many bins: little contention).
// v_readlane (the leader is wave-uniform), not a shuffle: the round is two ballots, one atomic and two scalar reads.
__device__ __forceinline__ int wave_bin_claim(int *bins, int bin, bool active) {
    int slot = 0;
    const unsigned long long todo = __ballot(active);
    if (todo == 0ull) {
        return 0;
    }
    const int leader = __ffsll((long long)todo) - 1;
    const int leader_bin = __builtin_amdgcn_readlane(bin, leader);
    const bool with_leader = active && bin == leader_bin;
    const unsigned long long same = __ballot(with_leader);
    int base = 0;
    if ((int)(threadIdx.x & 63) == leader) {
        base = atomicAdd(&bins[leader_bin], __popcll(same));
    }
    base = __builtin_amdgcn_readlane(base, leader);
    if (with_leader) {
        slot = base + __popcll(same & ((1ull << (threadIdx.x & 63)) - 1ull));
    } else if (active) {
        slot = atomicAdd(&bins[bin], 1);
    }
    return slot;
}

constexpr int kOrderPerThread = 32;  // features per thread: the kernel orders up to 32 768 features
__global__ void __launch_bounds__(kOrderBlock) klt_order_kernel(const uint32_t *iters, int32_t *order, int n, int *flat_out, int *skip_calls) {
    __shared__ int bin_count[256];
    __shared__ int bin_start[256];
    __shared__ int flat;
    // after a no-tail verdict the identity stays in order[] and the next calls' launches of this kernel return at once
    // (the host cannot know the verdict without synchronising; when it does see it, it stops launching for a while)
    __shared__ int skipping;
    if (threadIdx.x == 0) {
        const int left = *skip_calls;
        skipping = left;
        if (left > 0) {
            *skip_calls = left - 1;
        }
    }
    __syncthreads();
    if (skipping > 0) {
        return;
    }
    // every count this thread owns (features t, t + 1024, ...) is fetched up front: the loads overlap instead of paying one
    // memory latency per 1 024 features, twice
    int bins[kOrderPerThread];
#pragma unroll
    for (int k = 0; k < kOrderPerThread; ++k) {
        const int i = (int)threadIdx.x + k * kOrderBlock;
        bins[k] = i < n ? 255 - (int)min(iters[i], 255u) : -1;
    }
    for (int k = (int)threadIdx.x; k < 256; k += kOrderBlock) {
        bin_count[k] = 0;
    }
    __syncthreads();
    // counting pass: nothing is read back, so the atomics are fire-and-forget (no round trip per 1 024 features); lanes that
    // share the lowest active lane's bin still go in as one addition
#pragma unroll
    for (int k = 0; k < kOrderPerThread; ++k) {
        if (k * kOrderBlock < n) {  // block-uniform
            const bool active = bins[k] >= 0;
            const unsigned long long todo = __ballot(active);
            if (todo != 0ull) {
                const int leader = __ffsll((long long)todo) - 1;
                const int leader_bin = __builtin_amdgcn_readlane(bins[k], leader);
                const bool with_leader = active && bins[k] == leader_bin;
                const unsigned long long same = __ballot(with_leader);
                if ((int)(threadIdx.x & 63) == leader) {
                    atomicAdd(&bin_count[leader_bin], __popcll(same));
                } else if (active && !with_leader) {
                    atomicAdd(&bin_count[bins[k]], 1);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // exclusive scan of 256 bins by one wave: four bins per lane; and the no-tail test
        const int l = (int)threadIdx.x;
        const int c0 = bin_count[4 * l], c1 = bin_count[4 * l + 1], c2 = bin_count[4 * l + 2], c3 = bin_count[4 * l + 3];
        int run = c0 + c1 + c2 + c3;
        // bin k holds count 255 - k: weighted sum and largest count over the tracked features (count > 0: bins 0..254)
        long long weighted = (long long)c0 * (255 - 4 * l) + (long long)c1 * (254 - 4 * l) + (long long)c2 * (253 - 4 * l) + (long long)c3 * (252 - 4 * l);
        int tracked = run - (l == 63 ? c3 : 0);
        int largest = c0 ? 255 - 4 * l : (c1 ? 254 - 4 * l : (c2 ? 253 - 4 * l : (c3 ? 252 - 4 * l : 0)));
        for (int off = 32; off >= 1; off >>= 1) {
            weighted += __shfl_xor(weighted, off);
            tracked += __shfl_xor(tracked, off);
            largest = max(largest, __shfl_xor(largest, off));
        }
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(run, off);
            if (l >= off) {
                run += up;
            }
        }
        const int before = run - (c0 + c1 + c2 + c3);
        bin_start[4 * l] = before;
        bin_start[4 * l + 1] = before + c0;
        bin_start[4 * l + 2] = before + c0 + c1;
        bin_start[4 * l + 3] = before + c0 + c1 + c2;
        if (l == 0) {
            flat = (tracked == 0 || 2ll * largest * tracked <= 3ll * weighted) ? 1 : 0;  // largest <= 1.5 x mean
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && flat) {
        *skip_calls = 15;
        if (flat_out) {
            *flat_out = 1;  // host-mapped: the host reads it, unsynchronised, before a later call (a hint: stale is fine)
        }
    }
    if (flat) {
        for (int i = (int)threadIdx.x; i < n; i += kOrderBlock) {
            order[i] = i;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kOrderPerThread; ++k) {
        if (k * kOrderBlock < n) {  // block-uniform
            const int slot = wave_bin_claim(bin_start, max(bins[k], 0), bins[k] >= 0);
            if (bins[k] >= 0) {
                order[slot] = (int)threadIdx.x + k * kOrderBlock;
            }
        }
    }
}

}  // namespace

hipError_t pyramid_downsample_launch(const uint8_t *src, int32_t src_rows, int32_t src_cols, uint8_t *dst, hipStream_t stream) {
    const int dst_rows = src_rows / 2, dst_cols = src_cols / 2;
    if (dst_rows <= 0 || dst_cols <= 0) {
        return hipSuccess;
    }
    const long long groups = (long long)((dst_cols + 3) >> 2) * dst_rows;
    const unsigned blocks = (unsigned)((groups + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(downsample_kernel, dim3(blocks), dim3(kBlock), 0, stream, src, src_cols, dst, dst_rows, dst_cols);
    return hipGetLastError();
}

hipError_t extract_patch_launch(DevImage ref, float u, float v, int32_t ex_rows, int32_t ex_cols, float *d_patch, uint8_t *d_valid,
                                uint32_t *d_count, hipStream_t stream) {
    hipLaunchKernelGGL(extract_patch_kernel, dim3(1), dim3(64), 0, stream, ref, u, v, ex_rows, ex_cols, d_patch, d_valid, d_count);
    return hipGetLastError();
}

hipError_t klt_order_launch(const uint32_t *iters, int32_t *order, int32_t n, int *flat_out, int *skip_calls, hipStream_t stream) {
    if (n <= 0 || n > kKltOrderMaxFeatures) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(klt_order_kernel, dim3(1), dim3(kOrderBlock), 0, stream, iters, order, n, flat_out, skip_calls);
    return hipGetLastError();
}

hipError_t unpack_klt_shards_launch(const uint8_t *d_gathered, int32_t n, int32_t world, int32_t cap, int64_t shard_bytes, float *d_uv_out,
                                   uint8_t *d_status_out, hipStream_t stream) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(unpack_klt_shards_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_gathered, n, world, cap,
                       (long long)shard_bytes, reinterpret_cast<float2 *>(d_uv_out), d_status_out);
    return hipGetLastError();
}

}  // namespace ftk
