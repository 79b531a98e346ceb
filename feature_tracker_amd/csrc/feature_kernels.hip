// feature_kernels.hip — producers of the matcher's / tracker's inputs on gfx950.
//
//  * brief_kernel: BRIEF-n descriptors, bit-packed, straight into the layout ftk_hamming_match reads
//    (removes the host-side per-bit -> word packing of the reference flow,
//    test/test_descriptor_matcher_brief.cpp:69-88 times descriptor computation + matching together).
//    One wavefront per feature: the (2h+3)^2 neighbourhood is staged in LDS once, the (2h+1)^2
//    3x3 box sums are computed once, then each lane evaluates 4 of the 256 binary tests and the
//    64 results of a round are packed with one ballot.  Integer-only; bit-exact by construction.
//    Definition (normative, the reference's own BRIEF lives in the un-vendored Feature_Detector
//    repo): see oracle/oracle_brief.c.
#include "ftk_device.h"

#include <limits.h>

namespace ftk {
namespace {

__device__ __forceinline__ int f2i(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : INT_MIN; }

__global__ void __launch_bounds__(64) brief_kernel(const BriefParams p) {
    extern __shared__ unsigned short brief_lds[];
    const int lane = threadIdx.x;
    const int f = blockIdx.x;
    const int h = p.half;
    const int side = 2 * h + 1;   // box-sum lattice
    const int wside = side + 2;   // pixel window
    unsigned short *pix = brief_lds;
    unsigned short *sums = brief_lds + wside * wside;

    const int r = f2i(p.uv[2 * f + 1] + 0.5f);
    const int c = f2i(p.uv[2 * f] + 0.5f);
    const int margin = h + 1;
    uint32_t *out = p.words + (size_t)f * p.n_words;
    if (!(r >= margin && c >= margin && r < p.img.rows - margin && c < p.img.cols - margin)) {
        for (int w = lane; w < p.n_words; w += 64) {
            out[w] = 0u;
        }
        return;
    }
    const uint8_t *base = p.img.data + (long long)(r - margin) * p.img.cols + (c - margin);
    for (int idx = lane; idx < wside * wside; idx += 64) {
        const int pr = idx / wside, pc = idx - pr * wside;
        pix[idx] = base[(long long)pr * p.img.cols + pc];
    }
    __syncthreads();
    for (int idx = lane; idx < side * side; idx += 64) {
        const int sr = idx / side, sc = idx - sr * side;
        const unsigned short *q = pix + sr * wside + sc;
        sums[idx] = (unsigned short)(q[0] + q[1] + q[2] + q[wside] + q[wside + 1] + q[wside + 2] + q[2 * wside] + q[2 * wside + 1] +
                                      q[2 * wside + 2]);
    }
    __syncthreads();
    for (int base_bit = 0; base_bit < p.n_bits; base_bit += 64) {
        const int i = base_bit + lane;
        bool bit = false;
        if (i < p.n_bits) {
            const char4 o = reinterpret_cast<const char4 *>(p.pattern)[i];
            bit = sums[(o.x + h) * side + (o.y + h)] < sums[(o.z + h) * side + (o.w + h)];
        }
        const unsigned long long mask = __ballot(bit);
        if (lane == 0) {
            const int w = base_bit >> 5;
            out[w] = (uint32_t)mask;
            if (w + 1 < p.n_words) {
                out[w + 1] = (uint32_t)(mask >> 32);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Harris corners (definition: oracle/oracle_harris.c).  Four streaming passes over the image:
// Sobel gradients (int16), 5x5 structure tensor + response (fp32, integer sums are exact so the
// summation order is free), separable window maximum of a sortable 64-bit key
// (response bits << 32 | ~pixel index), and compaction of the survivors.
// ---------------------------------------------------------------------------------------------
constexpr int kHarrisHalf = 2;
constexpr int kHarrisBorder = 11;

__global__ void __launch_bounds__(256) harris_gradient_kernel(DevImage im, short *gx, short *gy) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)im.rows * im.cols;
    if (i >= total) {
        return;
    }
    const int r = (int)(i / im.cols), c = (int)(i - (long long)r * im.cols);
    int vx = 0, vy = 0;
    if (r >= 1 && c >= 1 && r < im.rows - 1 && c < im.cols - 1) {
        const uint8_t *p = im.data + (long long)r * im.cols + c;
        const int tl = p[-im.cols - 1], tc = p[-im.cols], tr = p[-im.cols + 1];
        const int ml = p[-1], mr = p[1];
        const int bl = p[im.cols - 1], bc = p[im.cols], br = p[im.cols + 1];
        vx = (tr + 2 * mr + br) - (tl + 2 * ml + bl);
        vy = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
    }
    gx[i] = (short)vx;
    gy[i] = (short)vy;
}

__device__ __forceinline__ unsigned sortable_bits(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256) harris_response_kernel(int rows, int cols, const short *gx, const short *gy, float min_response,
                                                              float *response, unsigned long long *key) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)rows * cols;
    if (i >= total) {
        return;
    }
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    float resp = 0.0f;
    unsigned long long k = 0ull;
    if (r >= kHarrisBorder && c >= kHarrisBorder && r < rows - kHarrisBorder && c < cols - kHarrisBorder) {
        int a = 0, b = 0, d = 0;
        for (int dr = -kHarrisHalf; dr <= kHarrisHalf; ++dr) {
            const long long base = (long long)(r + dr) * cols + c;
#pragma unroll
            for (int dc = -kHarrisHalf; dc <= kHarrisHalf; ++dc) {
                const int x = gx[base + dc], y = gy[base + dc];
                a += x * x;
                b += x * y;
                d += y * y;
            }
        }
        const float fa = (float)a, fb = (float)b, fd = (float)d;
        const float det = fa * fd - fb * fb;
        const float tr = fa + fd;
        resp = (det - (0.04f * tr) * tr) * 1e-6f;
        if (resp > min_response) {
            k = ((unsigned long long)sortable_bits(resp) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
    }
    if (response) {
        response[i] = resp;
    }
    key[i] = k;
}

// out[r][c] = max over |offset| <= reach of in along a row (horizontal != 0) or a column
__global__ void __launch_bounds__(256) harris_window_max_kernel(int rows, int cols, int reach, int horizontal, const unsigned long long *in,
                                                                unsigned long long *out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)rows * cols;
    if (i >= total) {
        return;
    }
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    unsigned long long m = 0ull;
    if (horizontal) {
        const int lo = max(c - reach, 0), hi = min(c + reach, cols - 1);
        const unsigned long long *row = in + (long long)r * cols;
        for (int cc = lo; cc <= hi; ++cc) {
            const unsigned long long v = row[cc];
            m = v > m ? v : m;
        }
    } else {
        const int lo = max(r - reach, 0), hi = min(r + reach, rows - 1);
        for (int rr = lo; rr <= hi; ++rr) {
            const unsigned long long v = in[(long long)rr * cols + c];
            m = v > m ? v : m;
        }
    }
    out[i] = m;
}

__global__ void __launch_bounds__(256) harris_collect_kernel(long long total, const unsigned long long *key, const unsigned long long *window_max,
                                                             unsigned long long *list, unsigned *count, unsigned capacity) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) {
        return;
    }
    const unsigned long long k = key[i];
    if (k != 0ull && k == window_max[i]) {
        const unsigned slot = atomicAdd(count, 1u);
        if (slot < capacity) {
            list[slot] = k;
        }
    }
}

}  // namespace

hipError_t harris_launch(const HarrisParams &p, hipStream_t stream) {
    const long long total = (long long)p.img.rows * p.img.cols;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(harris_gradient_kernel, dim3(blocks), dim3(256), 0, stream, p.img, p.gx, p.gy);
    hipLaunchKernelGGL(harris_response_kernel, dim3(blocks), dim3(256), 0, stream, p.img.rows, p.img.cols, p.gx, p.gy, p.min_response, p.response,
                       p.key);
    if (p.list) {
        const int reach = (p.min_distance > 1 ? p.min_distance : 1) - 1;
        hipLaunchKernelGGL(harris_window_max_kernel, dim3(blocks), dim3(256), 0, stream, p.img.rows, p.img.cols, reach, 1, p.key, p.tmp);
        hipLaunchKernelGGL(harris_window_max_kernel, dim3(blocks), dim3(256), 0, stream, p.img.rows, p.img.cols, reach, 0, p.tmp, p.wmax);
        hipError_t e = hipMemsetAsync(p.count, 0, sizeof(unsigned), stream);
        if (e != hipSuccess) {
            return e;
        }
        hipLaunchKernelGGL(harris_collect_kernel, dim3(blocks), dim3(256), 0, stream, total, p.key, p.wmax, p.list, p.count, p.capacity);
    }
    return hipGetLastError();
}

hipError_t brief_launch(const BriefParams &p, hipStream_t stream) {
    if (p.n <= 0) {
        return hipSuccess;
    }
    const int side = 2 * p.half + 1, wside = side + 2;
    const size_t lds = sizeof(unsigned short) * (size_t)(wside * wside + side * side);
    hipLaunchKernelGGL(brief_kernel, dim3((unsigned)p.n), dim3(64), lds, stream, p);
    return hipGetLastError();
}

// First-use cost out of the callers' timed regions (ftk_warmup): launching this empty kernel makes the runtime load this
// translation unit's code object onto the device, which otherwise happens inside the first real call.
__global__ void feature_warm_kernel() {}
hipError_t feature_warm(hipStream_t stream) {
    hipLaunchKernelGGL(feature_warm_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace ftk
