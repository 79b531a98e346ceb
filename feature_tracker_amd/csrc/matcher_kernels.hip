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

// Register-tiled scan (n_bits > 0): thread i keeps kRefs reference descriptors in registers, so one
// broadcast LDS read of a candidate feeds kRefs popcount chains, and the running best is a packed
// integer key (distance << 16 | position in the tile) maintained with one v_lshl_or + one v_min_u32
// per pair — the minimum of that key is the smallest distance and, among equals, the lowest j, which
// is what the reference's strict '<' scan returns.  Tiles are visited in ascending j and a later tile
// replaces the best only with a strictly smaller distance.  18 VALU per pair (8 xor, 8 bcnt, 2 key).
constexpr int kRefs = 2;

template <int NW, bool kNearby>
__global__ void __launch_bounds__(kBlock) hamming_match_tiled_kernel(const MatchParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t tile_words[kTile * NW];
    __shared__ float2 tile_uv[kTile];

    const int i_base = (blockIdx.x * kBlock + threadIdx.x) * kRefs;
    uint32_t ref[kRefs][NW];
    float pred_u[kRefs], pred_v[kRefs];
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        const int i = i_base + r;
        const bool active = i < p.n_ref;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            ref[r][w] = active ? p.ref_words[(long long)i * NW + w] : 0u;
        }
        pred_u[r] = (kNearby && active) ? p.pred_uv[2 * i] : 0.0f;
        pred_v[r] = (kNearby && active) ? p.pred_uv[2 * i + 1] : 0.0f;
    }
    const int j_begin = blockIdx.y * p.cur_per_block;
    const int j_end = min(j_begin + p.cur_per_block, p.n_cur);

    uint32_t best_d[kRefs];
    int best_j[kRefs];
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        best_d[r] = 0xFFFFu;  // above any real distance (<= 512)
        best_j[r] = -1;
    }
    for (int tile_begin = j_begin; tile_begin < j_end; tile_begin += kTile) {
        const int tile_n = min(kTile, j_end - tile_begin);
        __syncthreads();
        for (int idx = (int)threadIdx.x; idx < tile_n * NW; idx += kBlock) {
            tile_words[idx] = p.cur_words[(long long)tile_begin * NW + idx];
        }
        if (kNearby && (int)threadIdx.x < tile_n) {
            tile_uv[threadIdx.x] = make_float2(p.cur_uv[2 * (tile_begin + threadIdx.x)], p.cur_uv[2 * (tile_begin + threadIdx.x) + 1]);
        }
        __syncthreads();
        uint32_t key[kRefs];
#pragma unroll
        for (int r = 0; r < kRefs; ++r) {
            key[r] = 0xFFFFFFFFu;
        }
#pragma unroll 2
        for (int t = 0; t < tile_n; ++t) {
            uint32_t cw[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                cw[w] = tile_words[t * NW + w];
            }
            float2 c = make_float2(0.0f, 0.0f);
            if (kNearby) {
                c = tile_uv[t];
            }
#pragma unroll
            for (int r = 0; r < kRefs; ++r) {
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    d += __popc(ref[r][w] ^ cw[w]);
                }
                uint32_t k = (d << 16) | (uint32_t)t;
                if (kNearby) {
                    // descriptor_matcher.h:108-111: outside the window -> not a candidate
                    const bool out = (int)(fabsf(pred_u[r] - c.x) > p.max_col) | (int)(fabsf(pred_v[r] - c.y) > p.max_row);
                    k = out ? 0xFFFFFFFFu : k;
                }
                key[r] = min(key[r], k);
            }
        }
#pragma unroll
        for (int r = 0; r < kRefs; ++r) {
            const uint32_t d = key[r] >> 16;
            if (d < best_d[r]) {  // strict: an earlier tile keeps ties
                best_d[r] = d;
                best_j[r] = tile_begin + (int)(key[r] & 0xFFFFu);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        const int i = i_base + r;
        // `distance < min_distance && distance < threshold` with min_distance starting at the threshold
        if (i < p.n_ref && best_j[r] >= 0 && (float)best_d[r] < p.max_distance) {
            const unsigned long long packed = ((unsigned long long)best_d[r] << 32) | (unsigned)best_j[r];
            atomicMin(&p.keys[i], packed);
        }
    }
}

__global__ void __launch_bounds__(kBlock) match_epilogue_kernel(unsigned long long *keys, int32_t *index_pairs, int n_ref) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_ref) {
        const unsigned long long key = keys[i];
        if (key != kNoMatch) {
            index_pairs[i] = (int32_t)(key & 0xFFFFFFFFull);
            keys[i] = kNoMatch;  // leave the workspace ready for the next call
        }
    }
}

template <int NW>
hipError_t launch_nw(const MatchParams &p, hipStream_t stream) {
    const int splits = (p.n_cur + p.cur_per_block - 1) / p.cur_per_block;
    if (p.n_bits == 0) {
        // ComputeDistance's "empty descriptor" answer (kMaxInt32) does not fit the packed key: plain scan
        const int row_blocks = (p.n_ref + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(hamming_match_kernel<NW>, dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
        return hipGetLastError();
    }
    const int row_blocks = (p.n_ref + kBlock * kRefs - 1) / (kBlock * kRefs);
    if (p.pred_uv) {
        hipLaunchKernelGGL((hamming_match_tiled_kernel<NW, true>), dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
    } else {
        hipLaunchKernelGGL((hamming_match_tiled_kernel<NW, false>), dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t match_launch(const MatchParams &p, hipStream_t stream) {
    if (p.n_ref <= 0 || p.n_cur <= 0) {
        return hipSuccess;
    }
    hipError_t e = hipSuccess;
    if (!p.keys_clean) {
        e = hipMemsetAsync(p.keys, 0xFF, sizeof(unsigned long long) * (size_t)p.n_ref, stream);
        if (e != hipSuccess) {
            return e;
        }
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
