// matcher_kernels.hip — brute-force / windowed Hamming matcher for gfx950 (MI355X).
//
// DescriptorMatcher<BriefType>::ForceMatch / NearbyMatch (descriptor_matcher.h:55-79, :90-124)
// with the per-bit distance of test/test_descriptor_matcher_brief.cpp:33-45, on bit-packed
// descriptors: distance = sum_w popcount(ref[w] ^ cur[w]).
//
// Layout of the work: thread i of a 256-thread workgroup keeps ref descriptor i in registers;
// the workgroup streams a slice of `cur` through LDS in tiles of 256 descriptors (coalesced
// 16-byte global loads, then LDS broadcast reads: every lane reads the same candidate), and
// each thread runs the reference's j-ascending scan with a strict '<' on the running minimum,
// so ties resolve to the lowest j exactly like the scalar loop.  The candidate range is split
// over blockIdx.y to fill the chip (10 000 refs are only 157 waves); partial results meet in a
// 64-bit atomicMin on the packed key (distance << 32 | j), which is the same lexicographic
// order, and a tiny epilogue kernel writes index_pairs[i] only where a match exists (the
// reference leaves the entry untouched otherwise).  Integer / popcount VALU work: v_xor_b32 +
// v_bcnt_u32_b32 (popcount-accumulate) — no MFMA, the data are bits.
#include "ftk_device.h"

namespace ftk {
namespace {

constexpr int kBlock = 256;
constexpr int kTile = 256;  // candidates staged per LDS tile
constexpr unsigned long long kNoMatch = ~0ull;

template <int NW>
__global__ void __launch_bounds__(kBlock) hamming_match_kernel(const MatchParams p) {
    __shared__ uint32_t tile_words[kTile * NW];
    __shared__ float2 tile_uv[kTile];

    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < p.n_ref;
    const bool nearby = p.pred_uv != nullptr;

    uint32_t ref[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        ref[w] = active ? p.ref_words[(long long)i * NW + w] : 0u;
    }
    float pred_u = 0.0f, pred_v = 0.0f;
    if (nearby && active) {
        pred_u = p.pred_uv[2 * i];
        pred_v = p.pred_uv[2 * i + 1];
    }

    const int j_begin = blockIdx.y * p.cur_per_block;
    const int j_end = min(j_begin + p.cur_per_block, p.n_cur);

    // running minimum of the reference loop: starts AT the threshold, strict '<' to improve
    float min_distance = p.max_distance;
    unsigned best_d = 0xFFFFFFFFu;
    int best_j = -1;

    for (int tile_begin = j_begin; tile_begin < j_end; tile_begin += kTile) {
        const int tile_n = min(kTile, j_end - tile_begin);
        __syncthreads();
        for (int idx = (int)threadIdx.x; idx < tile_n * NW; idx += kBlock) {
            tile_words[idx] = p.cur_words[(long long)tile_begin * NW + idx];
        }
        if (nearby && (int)threadIdx.x < tile_n) {
            tile_uv[threadIdx.x] = make_float2(p.cur_uv[2 * (tile_begin + threadIdx.x)], p.cur_uv[2 * (tile_begin + threadIdx.x) + 1]);
        }
        __syncthreads();
        if (!active) {
            continue;
        }
        for (int t = 0; t < tile_n; ++t) {
            if (nearby) {
                const float2 c = tile_uv[t];
                if (fabsf(pred_u - c.x) > p.max_col || fabsf(pred_v - c.y) > p.max_row) {
                    continue;
                }
            }
            unsigned d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                d += __popc(ref[w] ^ tile_words[t * NW + w]);
            }
            // ComputeDistance returns kMaxInt32 for empty descriptors (n_bits == 0)
            const unsigned du = (p.n_bits == 0) ? 0x7FFFFFFFu : d;
            const float distance = (p.n_bits == 0) ? 2147483648.0f : (float)d;
            if (distance < min_distance && distance < p.max_distance) {
                min_distance = distance;
                best_d = du;
                best_j = tile_begin + t;
            }
        }
    }

    if (active && best_j >= 0) {
        const unsigned long long key = ((unsigned long long)best_d << 32) | (unsigned)best_j;
        atomicMin(&p.keys[i], key);
    }
}

__global__ void __launch_bounds__(kBlock) match_epilogue_kernel(const unsigned long long *keys, int32_t *index_pairs, int n_ref) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_ref) {
        const unsigned long long key = keys[i];
        if (key != kNoMatch) {
            index_pairs[i] = (int32_t)(key & 0xFFFFFFFFull);
        }
    }
}

template <int NW>
hipError_t launch_nw(const MatchParams &p, hipStream_t stream) {
    const int row_blocks = (p.n_ref + kBlock - 1) / kBlock;
    const int splits = (p.n_cur + p.cur_per_block - 1) / p.cur_per_block;
    hipLaunchKernelGGL(hamming_match_kernel<NW>, dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

hipError_t match_launch(const MatchParams &p, hipStream_t stream) {
    if (p.n_ref <= 0 || p.n_cur <= 0) {
        return hipSuccess;
    }
    hipError_t e = hipMemsetAsync(p.keys, 0xFF, sizeof(unsigned long long) * (size_t)p.n_ref, stream);
    if (e != hipSuccess) {
        return e;
    }
    switch (p.n_words) {
        case 1: e = launch_nw<1>(p, stream); break;
        case 2: e = launch_nw<2>(p, stream); break;
        case 4: e = launch_nw<4>(p, stream); break;
        case 8: e = launch_nw<8>(p, stream); break;
        case 16: e = launch_nw<16>(p, stream); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(match_epilogue_kernel, dim3((p.n_ref + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, p.keys, p.index_pairs, p.n_ref);
    return hipGetLastError();
}

}  // namespace ftk
