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

}  // namespace

hipError_t brief_launch(const BriefParams &p, hipStream_t stream) {
    if (p.n <= 0) {
        return hipSuccess;
    }
    const int side = 2 * p.half + 1, wside = side + 2;
    const size_t lds = sizeof(unsigned short) * (size_t)(wside * wside + side * side);
    hipLaunchKernelGGL(brief_kernel, dim3((unsigned)p.n), dim3(64), lds, stream, p);
    return hipGetLastError();
}

}  // namespace ftk
