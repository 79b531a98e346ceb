// pyramid_kernels.hip — image-side helpers for gfx950 (MI355X).
//
//  * pyramid_downsample: ImagePyramid::CreateImagePyramid's level step (call sites
//    test/test_optical_flow.cpp:70-71): dst = truncating 2x2 box mean of src, (rows/2, cols/2).
//    A pure streaming kernel (1.25 B moved per source byte): every thread produces 4 adjacent
//    output pixels from two 8-byte row segments, so a wave reads 2 x 512 contiguous bytes.
//  * extract_patch: OpticalFlow::ExtractExtendPatchInReferenceImage (optical_flow.cpp:49-102),
//    exposed because the reference makes it a public method; one wave per call.
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

hipError_t unpack_klt_shards_launch(const uint8_t *d_gathered, int32_t n, int32_t world, int32_t cap, int64_t shard_bytes, float *d_uv_out,
                                   uint8_t *d_status_out, hipStream_t stream) {
    if (n <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(unpack_klt_shards_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, d_gathered, n, world, cap,
                       (long long)shard_bytes, reinterpret_cast<float2 *>(d_uv_out), d_status_out);
    return hipGetLastError();
}

// First-use cost out of the callers' timed regions (ftk_warmup): launching this empty kernel makes the runtime load this
// translation unit's code object onto the device, which otherwise happens inside the first real call.
__global__ void pyramid_warm_kernel() {}
hipError_t pyramid_warm(hipStream_t stream) {
    hipLaunchKernelGGL(pyramid_warm_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace ftk
