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

#include <stdlib.h>

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

// Any descriptor width (n_words at run time; used above 16 words, where no register-tiled instantiation exists): the
// reference's scan as written, thread i against candidates [j_begin, j_end) in ascending j, words straight from
// global memory (every lane reads the same candidate: one request per wave).
__global__ void __launch_bounds__(kBlock) hamming_match_generic_kernel(const MatchParams p) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.n_ref) {
        return;
    }
    const bool nearby = p.pred_uv != nullptr;
    const int nw = p.n_words;
    const uint32_t *ref = p.ref_words + (long long)i * nw;
    const float pred_u = nearby ? p.pred_uv[2 * i] : 0.0f, pred_v = nearby ? p.pred_uv[2 * i + 1] : 0.0f;
    const int j_begin = blockIdx.y * p.cur_per_block;
    const int j_end = min(j_begin + p.cur_per_block, p.n_cur);
    float min_distance = p.max_distance;
    unsigned best_d = 0xFFFFFFFFu;
    int best_j = -1;
    for (int j = j_begin; j < j_end; ++j) {
        if (nearby && (fabsf(pred_u - p.cur_uv[2 * j]) > p.max_col || fabsf(pred_v - p.cur_uv[2 * j + 1]) > p.max_row)) {
            continue;
        }
        const uint32_t *cur = p.cur_words + (long long)j * nw;
        unsigned d = 0;
        for (int w = 0; w < nw; ++w) {
            d += __popc(ref[w] ^ cur[w]);
        }
        const unsigned du = (p.n_bits == 0) ? 0x7FFFFFFFu : d;
        const float distance = (p.n_bits == 0) ? 2147483648.0f : (float)d;
        if (distance < min_distance && distance < p.max_distance) {
            min_distance = distance;
            best_d = du;
            best_j = j;
        }
    }
    if (best_j >= 0) {
        atomicMin(&p.keys[i], ((unsigned long long)best_d << 32) | (unsigned)best_j);
    }
}

// NearbyMatch: bounding boxes {u min, u max, v min, v max} for the early exit of the tiled scan — block b < row_blocks: the
// predictions of reference rows [512 b, 512 b + 512); block row_blocks + s: the candidates of split s.  A NaN coordinate
// opens the box to the whole plane; no point at all leaves the empty box.
__global__ void __launch_bounds__(kBlock) hamming_box_kernel(const MatchParams p, int row_blocks) {
    __shared__ float part[kBlock / 64][5];
    const float pos_inf = __uint_as_float(0x7F800000u), neg_inf = __uint_as_float(0xFF800000u);
    float u0 = pos_inf, u1 = neg_inf, v0 = pos_inf, v1 = neg_inf;
    bool unordered = false;
    const bool pred = (int)blockIdx.x < row_blocks;
    const float *uv = pred ? p.pred_uv : p.cur_uv;
    const int first = pred ? (int)blockIdx.x * kBlock * kMatchRefs : ((int)blockIdx.x - row_blocks) * p.cur_per_block;
    const int last = pred ? min(first + kBlock * kMatchRefs, p.n_ref) : min(first + p.cur_per_block, p.n_cur);
    for (int i = first + (int)threadIdx.x; i < last; i += kBlock) {
        const float u = uv[2 * i], v = uv[2 * i + 1];
        if (isnan(u) || isnan(v)) {
            unordered = true;
        } else {
            u0 = fminf(u0, u);
            u1 = fmaxf(u1, u);
            v0 = fminf(v0, v);
            v1 = fmaxf(v1, v);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        u0 = fminf(u0, __shfl_xor(u0, off));
        u1 = fmaxf(u1, __shfl_xor(u1, off));
        v0 = fminf(v0, __shfl_xor(v0, off));
        v1 = fmaxf(v1, __shfl_xor(v1, off));
    }
    const bool any_unordered = __ballot(unordered) != 0ull;
    const int w = (int)threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[w][0] = u0;
        part[w][1] = u1;
        part[w][2] = v0;
        part[w][3] = v1;
        part[w][4] = any_unordered ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool open_box = false;
        for (int k = 0; k < kBlock / 64; ++k) {
            u0 = fminf(k == 0 ? part[0][0] : u0, part[k][0]);
            u1 = fmaxf(k == 0 ? part[0][1] : u1, part[k][1]);
            v0 = fminf(k == 0 ? part[0][2] : v0, part[k][2]);
            v1 = fmaxf(k == 0 ? part[0][3] : v1, part[k][3]);
            open_box = open_box || part[k][4] != 0.0f;
        }
        p.boxes[blockIdx.x] = open_box ? make_float4(neg_inf, pos_inf, neg_inf, pos_inf) : make_float4(u0, u1, v0, v1);
    }
}

// Register-tiled scan (n_bits > 0): thread i keeps kRefs reference descriptors in registers, so one
// broadcast LDS read of a candidate feeds kRefs popcount chains, and the running best is a packed
// integer key (distance << 16 | position in the tile) maintained with one v_lshl_or + one v_min_u32
// per pair — the minimum of that key is the smallest distance and, among equals, the lowest j, which
// is what the reference's strict '<' scan returns.  Tiles are visited in ascending j and a later tile
// replaces the best only with a strictly smaller distance.  18 VALU per pair (8 xor, 8 bcnt, 2 key) on the full path,
// 13 (6 xor, 6 bcnt, 1 compare) on the early-exit path described in the kernel.
constexpr int kRefs = kMatchRefs;

// popcount(x) + acc in ONE instruction (v_bcnt_u32_b32 adds its second operand).  Written out because the compiler, left to
// itself, re-associates the eight counts of a distance into bcnt(x, 0) pairs joined by v_add3_u32 — two more VALU
// instructions per distance in a loop that is bound by exactly that.
static __device__ __forceinline__ uint32_t popc_add(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
static __device__ __forceinline__ uint32_t popc_first(uint32_t x) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}

template <int NW, bool kNearby>
__global__ void __launch_bounds__(kBlock) hamming_match_tiled_kernel(const MatchParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t tile_words[kTile * NW];
    __shared__ float2 tile_uv[kTile];

#ifdef FTK_MATCH_STAMPS
    const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime();
#endif
    // A workgroup that starts while older ones are in their popcount loops is the youngest on its SIMDs and gets only the
    // issue slots the others leave (oldest first): its descriptor loads went out 10 us late (stamps build).  Raised priority
    // until the descriptors are in registers lets it get its memory requests in flight at once.
    __builtin_amdgcn_s_setprio(3);
    // thread t owns rows t, t + 256, ... of the workgroup's block: a wave's 16-byte loads of one row segment then walk the
    // descriptors with a 32-byte stride (half of every fetched line is used by this load, the other half by the next one),
    // where rows 2t, 2t + 1 would stride 64 bytes — measured with the stamps build: the descriptor load of a workgroup
    // took 8 us (median) of its 30 us that way
    const int i_base = blockIdx.x * kBlock * kRefs + threadIdx.x;
    const int j_begin = blockIdx.y * p.cur_per_block;
#ifdef FTK_MATCH_STAMPS
    asm volatile("" ::"s"(j_begin));
    const unsigned long long stamp_args = __builtin_amdgcn_s_memrealtime();  // kernel arguments have arrived
#endif
    const int j_end = min(j_begin + p.cur_per_block, p.n_cur);
    if (kNearby && p.boxes != nullptr) {
        // NearbyMatch: when the bounding box of this workgroup's 512 predictions and the bounding box of its candidates
        // (hamming_box_kernel) are more than window + 1 px apart on an axis, no pair passes
        // fabs(du) <= max_col && fabs(dv) <= max_row (the pixel covers the rounding of the fp32 difference) and the
        // workgroup is done before it stages a candidate.  A NaN coordinate passes every window test
        // (descriptor_matcher.h:108-111), so it makes its box the whole plane.  Exact for any input; it pays when the
        // features come in spatial order (a detector scanning the image), where most workgroups leave here.
        const float4 pb = p.boxes[blockIdx.x], cb = p.boxes[gridDim.x + blockIdx.y];
        const float reach_u = p.max_col + 1.0f, reach_v = p.max_row + 1.0f;
        if (cb.x - pb.y > reach_u || pb.x - cb.y > reach_u || cb.z - pb.w > reach_v || pb.z - cb.w > reach_v) {
            return;  // block-uniform
        }
    }
    uint32_t ref[kRefs][NW];
    float pred_u[kRefs], pred_v[kRefs];
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        const int i = i_base + r * kBlock;
        const bool active = i < p.n_ref;
        // unconditional wide loads from a clamped row (a per-word `active ? load : 0` compiles to one exec-masked 4-byte load per
        // word: 16 scattered dword loads per thread — the stamps build showed the descriptor load at 8 us of a 30-us workgroup)
        const uint32_t *row = p.ref_words + (long long)(active ? i : p.n_ref - 1) * NW;
        if (NW % 4 == 0) {
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const uint4 q = *reinterpret_cast<const uint4 *>(row + w);
                ref[r][w] = q.x;
                ref[r][w + 1] = q.y;
                ref[r][w + 2] = q.z;
                ref[r][w + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                ref[r][w] = row[w];
            }
        }
        if (!active) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                ref[r][w] = 0u;
            }
        }
        pred_u[r] = (kNearby && active) ? p.pred_uv[2 * i] : 0.0f;
        pred_v[r] = (kNearby && active) ? p.pred_uv[2 * i + 1] : 0.0f;
    }

#ifdef FTK_MATCH_STAMPS
    asm volatile("" ::"v"(ref[0][0]));
    const unsigned long long stamp_after_load = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_setprio(0);
    uint32_t best_d[kRefs];
    int best_j[kRefs];
    // A pair matters only when its distance is below BOTH the row's running minimum (strict '<': ties keep the earlier j)
    // and the threshold (descriptor_matcher.h:68-71, :106-114: min_distance starts AT kMaxValidDescriptorDistance), and a
    // Hamming distance only grows word by word.  So after kEarly of the NW words, a candidate whose partial count has
    // already reached the row's limit is dead for that row; when that holds for every row of the wave — the normal case
    // once each row has met its true match, and from the first candidate on under a threshold well below the ~NW * 16 bits
    // of unrelated descriptors — the remaining words, the key and the minimum are skipped by one wave-uniform branch.
    // Exact for any input (the branch only skips work that cannot change the result); `limit` is kept conservative
    // (>= the true bound), the final comparison against max_distance below is the authoritative one.
    constexpr int kEarly = NW >= 8 ? (NW * 3) / 4 : NW;
    uint32_t limit[kRefs];
    const uint32_t limit0 = (p.max_distance >= 0.0f && p.max_distance < 65000.0f) ? (uint32_t)p.max_distance + 1u : (p.max_distance < 0.0f ? 0u : 0xFFFFu);
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        best_d[r] = 0xFFFFu;  // above any real distance (<= 512)
        best_j[r] = -1;
        limit[r] = limit0;
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
        // two candidates per step (indices clamped to the tile's last: seeing a candidate twice changes nothing), their first
        // kEarly words fetched one step ahead into the other register set so that the LDS latency sits under the previous
        // pair's popcounts; two steps per trip so that the two sets swap roles without register moves
        uint32_t cw[2][2][NW];
        auto fetch = [&](int set, int t) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int tc = min(t + c, tile_n - 1);
#pragma unroll
                for (int w = 0; w < kEarly; ++w) {
                    cw[set][c][w] = tile_words[tc * NW + w];
                }
            }
        };
        auto step = [&](int set, int t) {
            uint32_t d[2][kRefs];
            bool alive = false;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int r = 0; r < kRefs; ++r) {
                    d[c][r] = popc_first(ref[r][0] ^ cw[set][c][0]);
#pragma unroll
                    for (int w = 1; w < kEarly; ++w) {
                        d[c][r] = popc_add(ref[r][w] ^ cw[set][c][w], d[c][r]);
                    }
                    alive = alive | (d[c][r] < limit[r]);
                }
            }
            if (kEarly == NW || __builtin_expect(__ballot(alive) != 0ull, 0)) {  // wave-uniform
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int tc = min(t + c, tile_n - 1);
#pragma unroll
                    for (int w = kEarly; w < NW; ++w) {
                        cw[set][c][w] = tile_words[tc * NW + w];
                    }
                    float2 cuv = make_float2(0.0f, 0.0f);
                    if (kNearby) {
                        cuv = tile_uv[tc];
                    }
#pragma unroll
                    for (int r = 0; r < kRefs; ++r) {
#pragma unroll
                        for (int w = kEarly; w < NW; ++w) {
                            d[c][r] = popc_add(ref[r][w] ^ cw[set][c][w], d[c][r]);
                        }
                        uint32_t k = (d[c][r] << 16) | (uint32_t)tc;
                        if (kNearby) {
                            // descriptor_matcher.h:108-111: outside the window -> not a candidate
                            const bool out = (int)(fabsf(pred_u[r] - cuv.x) > p.max_col) | (int)(fabsf(pred_v[r] - cuv.y) > p.max_row);
                            k = out ? 0xFFFFFFFFu : k;
                        }
                        key[r] = min(key[r], k);
                        limit[r] = min(limit[r], key[r] >> 16);
                    }
                }
            }
        };
        fetch(0, 0);
        for (int t = 0; t < tile_n; t += 4) {
            fetch(1, t + 2);
            step(0, t);
            fetch(0, t + 4);
            step(1, t + 2);
        }
#pragma unroll
        for (int r = 0; r < kRefs; ++r) {
            const uint32_t dist = key[r] >> 16;
            if (dist < best_d[r]) {  // strict: an earlier tile keeps ties
                best_d[r] = dist;
                best_j[r] = tile_begin + (int)(key[r] & 0xFFFFu);
            }
        }
    }
#ifdef FTK_MATCH_STAMPS
    if (threadIdx.x == 0 && p.stamps) {
        unsigned long long *st = p.stamps + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        st[0] = stamp_t0;
        st[1] = __builtin_amdgcn_s_memtime() - stamp_c0;  // shader-clock ticks over the workgroup's life
        (void)stamp_after_load;
        st[2] = __builtin_amdgcn_s_memrealtime();
        st[3] = ((stamp_args - stamp_t0) << 40) | ((unsigned long long)(__builtin_amdgcn_s_getreg((20 << 11) | 20) & 0xF) << 32) |
                (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        const int i = i_base + r * kBlock;
        // `distance < min_distance && distance < threshold` with min_distance starting at the threshold
        if (i < p.n_ref && best_j[r] >= 0 && (float)best_d[r] < p.max_distance) {
            const unsigned long long packed = ((unsigned long long)best_d[r] << 32) | (unsigned)best_j[r];
            atomicMin(&p.keys[i], packed);
        }
    }
}

// The same scan with the candidates on the SCALAR path: every lane of a wave compares its own reference rows with the same
// candidate, so the candidate's words are wave-uniform — they are fetched with s_load_dwordx8 through the scalar cache
// into SGPRs (constant address space) and feed v_xor_b32 as its scalar operand.  No LDS tile, no staging pass, no barrier,
// no LDS address arithmetic on the VALU: what is left per pair is 6 xor + 6 bcnt + 1 compare on the early-exit path.
// The waves of a workgroup never synchronise; a candidate pair's 64 bytes are requested one step ahead of their use.
typedef const __attribute__((address_space(4))) uint32_t *scalar_words;
typedef const __attribute__((address_space(4))) float *scalar_floats;
constexpr int kChunk = 32768;  // candidates per packed-key epoch (position in 16 bits)

template <int NW, bool kNearby>
__global__ void __launch_bounds__(kBlock) hamming_match_scalar_kernel(const MatchParams p) {
    __builtin_amdgcn_s_setprio(3);  // see hamming_match_tiled_kernel: get the descriptor loads out at once
    const int i_base = blockIdx.x * kBlock * kRefs + threadIdx.x;
    const int j_begin = blockIdx.y * p.cur_per_block;
    const int j_end = min(j_begin + p.cur_per_block, p.n_cur);
    if (kNearby && p.boxes != nullptr) {
        const float4 pb = p.boxes[blockIdx.x], cb = p.boxes[gridDim.x + blockIdx.y];
        const float reach_u = p.max_col + 1.0f, reach_v = p.max_row + 1.0f;
        if (cb.x - pb.y > reach_u || pb.x - cb.y > reach_u || cb.z - pb.w > reach_v || pb.z - cb.w > reach_v) {
            return;  // block-uniform (hamming_match_tiled_kernel explains the box test)
        }
    }
    uint32_t ref[kRefs][NW];
    float pred_u[kRefs], pred_v[kRefs];
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        const int i = i_base + r * kBlock;
        const bool active = i < p.n_ref;
        const uint32_t *row = p.ref_words + (long long)(active ? i : p.n_ref - 1) * NW;
        if (NW % 4 == 0) {
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const uint4 q = *reinterpret_cast<const uint4 *>(row + w);
                ref[r][w] = q.x;
                ref[r][w + 1] = q.y;
                ref[r][w + 2] = q.z;
                ref[r][w + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                ref[r][w] = row[w];
            }
        }
        if (!active) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                ref[r][w] = 0u;
            }
        }
        pred_u[r] = (kNearby && active) ? p.pred_uv[2 * i] : 0.0f;
        pred_v[r] = (kNearby && active) ? p.pred_uv[2 * i + 1] : 0.0f;
    }
    __builtin_amdgcn_s_setprio(0);

    constexpr int kEarly = NW >= 8 ? (NW * 3) / 4 : NW;  // see hamming_match_tiled_kernel
    uint32_t best_d[kRefs], limit[kRefs];
    int best_j[kRefs];
    const uint32_t limit0 = (p.max_distance >= 0.0f && p.max_distance < 65000.0f) ? (uint32_t)p.max_distance + 1u : (p.max_distance < 0.0f ? 0u : 0xFFFFu);
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        best_d[r] = 0xFFFFu;
        best_j[r] = -1;
        limit[r] = limit0;
    }
    const scalar_words cur = (scalar_words)p.cur_words;
    const scalar_floats cur_uv = (scalar_floats)p.cur_uv;
    for (int chunk_begin = j_begin; chunk_begin < j_end; chunk_begin += kChunk) {
        const int chunk_end = min(chunk_begin + kChunk, j_end);
        const int last = chunk_end - 1;
        uint32_t key[kRefs];
#pragma unroll
        for (int r = 0; r < kRefs; ++r) {
            key[r] = 0xFFFFFFFFu;
        }
        // two candidates per step (indices clamped to the chunk's last: seeing a candidate twice changes nothing); two steps per
        // trip so that the two SGPR sets swap roles without moves
        uint32_t cw[2][2][NW];
        auto fetch = [&](int set, int j) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const scalar_words q = cur + (long long)min(j + c, last) * NW;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    cw[set][c][w] = q[w];
                }
            }
        };
        auto step = [&](int set, int j) {
            uint32_t d[2][kRefs];
            bool alive = false;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int r = 0; r < kRefs; ++r) {
                    d[c][r] = popc_first(ref[r][0] ^ cw[set][c][0]);
#pragma unroll
                    for (int w = 1; w < kEarly; ++w) {
                        d[c][r] = popc_add(ref[r][w] ^ cw[set][c][w], d[c][r]);
                    }
                    alive = alive | (d[c][r] < limit[r]);
                }
            }
            if (kEarly == NW || __builtin_expect(__ballot(alive) != 0ull, 0)) {  // wave-uniform
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int jc = min(j + c, last);
                    float cu = 0.0f, cv = 0.0f;
                    if (kNearby) {
                        cu = cur_uv[2 * (long long)jc];
                        cv = cur_uv[2 * (long long)jc + 1];
                    }
#pragma unroll
                    for (int r = 0; r < kRefs; ++r) {
#pragma unroll
                        for (int w = kEarly; w < NW; ++w) {
                            d[c][r] = popc_add(ref[r][w] ^ cw[set][c][w], d[c][r]);
                        }
                        uint32_t k = (d[c][r] << 16) | (uint32_t)(jc - chunk_begin);
                        if (kNearby) {
                            // descriptor_matcher.h:108-111: outside the window -> not a candidate
                            const bool out = (int)(fabsf(pred_u[r] - cu) > p.max_col) | (int)(fabsf(pred_v[r] - cv) > p.max_row);
                            k = out ? 0xFFFFFFFFu : k;
                        }
                        key[r] = min(key[r], k);
                        limit[r] = min(limit[r], key[r] >> 16);
                    }
                }
            }
        };
        fetch(0, chunk_begin);
        for (int j = chunk_begin; j < chunk_end; j += 4) {
            fetch(1, j + 2);
            step(0, j);
            fetch(0, j + 4);
            step(1, j + 2);
        }
#pragma unroll
        for (int r = 0; r < kRefs; ++r) {
            const uint32_t dist = key[r] >> 16;
            if (dist < best_d[r]) {  // strict: an earlier chunk keeps ties
                best_d[r] = dist;
                best_j[r] = chunk_begin + (int)(key[r] & 0xFFFFu);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kRefs; ++r) {
        const int i = i_base + r * kBlock;
        // `distance < min_distance && distance < threshold` with min_distance starting at the threshold
        if (i < p.n_ref && best_j[r] >= 0 && (float)best_d[r] < p.max_distance) {
            const unsigned long long packed = ((unsigned long long)best_d[r] << 32) | (unsigned)best_j[r];
            atomicMin(&p.keys[i], packed);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The all-pairs scan on the matrix cores.  Write a reference descriptor's bits as signed bytes a_k = +s_k (bit set) / -s_k (bit
// clear) and a candidate's bits as bytes b_k = 8 / s_k (bit set) / 0, with s_k one of 8, 4, 2, 1: then
//     sum_k a_k b_k = 8 (#(both set) - #(candidate set, reference clear)) = 8 (popcount(a) - hamming(a, b)),
// and that sum for 32 reference rows x 32 candidates x 32 bits is ONE v_mfma_i32_32x32x32_i8.  Started from -8 popcount(a) (the
// C operand of the first product) the accumulator IS -8 hamming: exact integer arithmetic, the same distances the popcount
// kernels compute.  The power-of-two scales exist so that a candidate dword needs ONE instruction: (word >> 4 h) & (0x01010101
// << v) puts bits 4 h + v + 8 j (j = 0..3) into its four bytes with value 2^v.  At 10 000 x 10 000 x 256 bits the popcount scan
// needs 13 VALU instructions per 64 pairs and is bound by exactly that (44 us); here 64 pairs cost half an MFMA (16 of its 32
// cycles) and ~2 VALU instructions, most of them issued in the shadow of the MFMAs.
//
//   * A wave owns 64 reference rows — two 32-row A tiles, expanded from the bits into registers once — and one split of the
//     candidates.  Nothing is shared between waves: no LDS, no barrier.  Lane (c = l & 31, h = l >> 5) loads the words of
//     candidate c of the current 32-candidate tile straight from global memory (one tile ahead of their use) and expands its
//     half of word m — dwords v = 0..3: bits 4 h + v + 8 j — between the MFMAs that consume them.  MFMA m multiplies word m;
//     lane (c, h) supplies the same 16 k-slots of row c (A) and candidate c (B), so whatever order the instruction gives its
//     32 k-slots, A and B agree on it.
//   * Result layout (C/D of every 32x32 MFMA): the lane is the candidate, the 16 registers are rows (i & 3) + 8 (i >> 2) + 4 h.
//     Each (lane, register) keeps a running key (-8 distance << 16 | 0xFFFE - position): its signed maximum is the smallest
//     distance and among equals the earliest candidate, the reference's strict '<'.  The keys start at the threshold
//     (descriptor_matcher.h:68-71: min_distance starts AT kMaxValidDescriptorDistance) with 0xFFFF in the position field.
//   * Per tile the 32 results of a lane are folded with v_max3_i32 and compared with the lane's lowest limit: only when some
//     lane holds a result above it (a match candidate: once per row with a true match, rarely otherwise) does the wave take the
//     path that compares per row, applies the NearbyMatch window and updates keys and limit.  Exact for any input.
//   * At the end the keys are reduced over the 32 lanes of each half (the candidates) and merged across splits with the same
//     64-bit atomicMin as the other scans; match_epilogue_kernel turns them into indices.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int kMfmaRows = 64;  // reference rows per workgroup: ONE wave (nothing is shared, so nothing is gained by larger groups,
                               // and single waves pack the SIMDs' two slots evenly)

// The rare path of the matrix-core scan: some lane of the wave holds a result above its lowest limit.  The caller folded the 32
// results into four group maxima (8 registers each); only groups, and in them only registers, in which SOME lane beats its key
// are looked at (wave-uniform skips: a true match touches one register of one group).  A result that beats the row's key
// replaces it if the candidate is inside the NearbyMatch window; `lowest` (the minimum of the lane's 32 limits) and the
// number of registers sitting on it are kept up to date.
template <bool kNearby>
static __device__ __forceinline__ void mfma_update_keys(const MatchParams &p, const v16i &acc0, const v16i &acc1, const int (&group_top)[4],
                                                        int *key, const float2 *pred_rows, int &lowest, int &at_lowest, int h, int cand, int j_begin) {
    const int inv_pos = 0xFFFE - (cand - j_begin);
    float cu = 0.0f, cv = 0.0f;
    if (kNearby) {
        cu = p.cur_uv[2 * (long long)cand];
        cv = p.cur_uv[2 * (long long)cand + 1];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (__ballot(group_top[g] > lowest) == 0ull) {
            continue;  // wave-uniform
        }
        int group_keys[8];  // all eight reads in flight before the first is looked at
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            group_keys[e] = key[64 * (8 * g + e)];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int s = 8 * g + e;
            const int neg_d8 = s < 16 ? acc0[s & 15] : acc1[s & 15];
            const int limit = group_keys[e] >> 16;
            const bool beats = neg_d8 > limit;
            if (__ballot(beats) == 0ull) {
                continue;  // wave-uniform
            }
            if (beats) {
                bool in_window = true;
                if (kNearby) {
                    // descriptor_matcher.h:108-111: outside the window -> not a candidate
                    const float2 pr = pred_rows[(s >> 4) * 32 + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h];
                    in_window = !((fabsf(pr.x - cu) > p.max_col) | (fabsf(pr.y - cv) > p.max_row));
                }
                if (in_window) {
                    key[64 * s] = (int)(((unsigned)neg_d8 << 16) | (unsigned)inv_pos);
                    at_lowest -= limit == lowest ? 1 : 0;
                }
            }
        }
    }
    // keys only grow, so the lane's lowest limit moves only once the last register that sat on it has moved: almost never
    if (__ballot(at_lowest == 0) != 0ull) {
        if (at_lowest == 0) {
            int low = 0x7FFF;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                low = min(low, key[64 * s] >> 16);
            }
            lowest = low;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                at_lowest += (key[64 * s] >> 16) == low ? 1 : 0;
            }
        }
    }
}

template <int NW, bool kNearby>
static __device__ __forceinline__ void mfma_scan(const MatchParams &p, int lane, int c, int h, int row0, int j_begin, int j_end) {
    const int last = j_end - 1;
#ifdef FTK_MATCH_STAMPS
    const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_slow = 0, st_slow_n = 0;
#endif
    // ---- candidate words: two register sets, the tile after the current one always in flight ----
    v4i words[2][NW / 4];
#define FTK_FETCH_WORDS(set_, tile_begin_)                                                                     \
    {                                                                                                          \
        /* 32-bit byte offset from the uniform base (the host keeps n_cur * NW * 4 below 2^31): two VALU instructions of address */ \
        const v4i *src_ = reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(p.cur_words) + (uint32_t)min((tile_begin_) + c, last) * (uint32_t)(4 * NW)); \
        _Pragma("unroll") for (int q_ = 0; q_ < NW / 4; ++q_) {                                                \
            words[set_][q_] = src_[q_];                                                                        \
        }                                                                                                      \
    }
    FTK_FETCH_WORDS(0, j_begin)

    // ---- the A operand: 64 rows as +-8 / +-4 / +-2 / +-1 bytes, and -8 popcount(row) as the accumulators' start ----
    // Only the first kEarly words are multiplied on the common path: a distance only grows word by word, so a candidate whose
    // partial distance has reached the row's limit is dead whatever the other words hold (as in the popcount scans above).
    constexpr int kEarly = (NW * 3) / 4;
    v4i a[2][NW];
    int pop_early[2], pop_rest[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = row0 + t * 32 + c;
        const v4i *src = reinterpret_cast<const v4i *>(p.ref_words + (long long)min(row, p.n_ref - 1) * NW);
        pop_early[t] = 0;
        pop_rest[t] = 0;
#pragma unroll
        for (int q = 0; q < NW / 4; ++q) {
            const v4i four = src[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t w = (uint32_t)four[e];
                if (4 * q + e < kEarly) {
                    pop_early[t] += __popc(w);
                } else {
                    pop_rest[t] += __popc(w);
                }
                const uint32_t half = w >> (4 * h);
                // dword v: bits 4 h + v + 8 j of the word -> byte j = +2^(3-v) (set) / -2^(3-v) (clear): v_perm_b32 picks byte 1 / byte 0
                a[t][4 * q + e][0] = (int)__builtin_amdgcn_perm(0u, 0x000008F8u, half & 0x01010101u);
                a[t][4 * q + e][1] = (int)__builtin_amdgcn_perm(0u, 0x000004FCu, (half >> 1) & 0x01010101u);
                a[t][4 * q + e][2] = (int)__builtin_amdgcn_perm(0u, 0x000002FEu, (half >> 2) & 0x01010101u);
                a[t][4 * q + e][3] = (int)__builtin_amdgcn_perm(0u, 0x000001FFu, (half >> 3) & 0x01010101u);
            }
        }
    }
    // accumulator starts: -8 popcount of the row's first kEarly words; the popcounts of the remaining words wait, four to a
    // register (each <= 128), for the rare completion
    v16i start0, start1;
    uint32_t rest_packed[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rest_packed[t][q] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2) + 4 * h;  // lane r (and r + 32) holds row r's popcounts
        start0[i] = -8 * __shfl(pop_early[0], r);
        start1[i] = -8 * __shfl(pop_early[1], r);
        rest_packed[0][i >> 2] |= (uint32_t)__shfl(pop_rest[0], r) << (8 * (i & 3));
        rest_packed[1][i >> 2] |= (uint32_t)__shfl(pop_rest[1], r) << (8 * (i & 3));
    }
    // d < limit0 is the conservative integer form of `distance < kMaxValidDescriptorDistance`
    const int limit0 = (p.max_distance >= 0.0f && p.max_distance < 4000.0f) ? (int)p.max_distance + 1 : (p.max_distance < 0.0f ? 0 : 4001);
    // the running keys live in LDS (register s of lane l at [64 s + l]: conflict-free): only the rare path touches them,
    // like the predicted positions of the wave's 64 rows (NearbyMatch)
    __shared__ int key_store[32 * 64];
    __shared__ float2 pred_rows[64];
    int *key = &key_store[lane];
    if (kNearby) {
        const int row = min(row0 + lane, p.n_ref - 1);
        pred_rows[lane] = make_float2(p.pred_uv[2 * (long long)row], p.pred_uv[2 * (long long)row + 1]);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        key[64 * s] = (int)(((unsigned)(-8 * limit0) << 16) | 0xFFFFu);
    }
    int lowest = -8 * limit0;  // min over this lane's 32 rows of the -8 distance a candidate has to exceed
    int at_lowest = 32;        // how many of the 32 sit on it
    const int shift = 4 * h;

#ifdef FTK_MATCH_STAMPS
#define FTK_SLOW_BEGIN const unsigned long long slow0_ = __builtin_amdgcn_s_memtime();
#define FTK_SLOW_END                                                \
    asm volatile("" ::"v"(lowest));                                 \
    st_slow += __builtin_amdgcn_s_memtime() - slow0_;               \
    ++st_slow_n;
#else
#define FTK_SLOW_BEGIN
#define FTK_SLOW_END
#endif
#define FTK_EXPAND_WORD(set_, m_)                                                         \
    const uint32_t half_ = (uint32_t)words[set_][(m_) / 4][(m_) % 4] >> shift;           \
    v4i b_;                                                                              \
    b_[0] = (int)(half_ & 0x01010101u);                                                  \
    b_[1] = (int)(half_ & 0x02020202u);                                                  \
    b_[2] = (int)(half_ & 0x04040404u);                                                  \
    b_[3] = (int)(half_ & 0x08080808u);
#define FTK_MFMA_TILE(set_, tile_begin_)                                                                                       \
    {                                                                                                                          \
        v16i acc0 = start0, acc1 = start1;                                                                                     \
        _Pragma("unroll") for (int m_ = 0; m_ < kEarly; ++m_) {                                                                \
            FTK_EXPAND_WORD(set_, m_)                                                                                          \
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][m_], b_, acc0, 0, 0, 0);                                          \
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1][m_], b_, acc1, 0, 0, 0);                                          \
        }                                                                                                                      \
        int gtop_[4]; /* maxima of registers 0-7 / 8-15 of either accumulator */                                               \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                                     \
            const v16i &acc_ = g_ < 2 ? acc0 : acc1;                                                                           \
            const int o_ = 8 * (g_ & 1);                                                                                       \
            gtop_[g_] = max(max(max(max(acc_[o_], acc_[o_ + 1]), acc_[o_ + 2]), max(acc_[o_ + 3], acc_[o_ + 4])),              \
                            max(max(acc_[o_ + 5], acc_[o_ + 6]), acc_[o_ + 7]));                                               \
        }                                                                                                                      \
        const int top_ = max(max(gtop_[0], gtop_[1]), max(gtop_[2], gtop_[3]));                                                \
        if (__builtin_expect(__ballot(top_ > lowest) != 0ull, 0)) { /* wave-uniform */                                          \
            FTK_SLOW_BEGIN                                                                                                     \
            /* complete the distances: the remaining words' popcounts and products (group maxima of the partial results */     \
            /* stay valid as a filter: a completed result is never larger) */                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                                                                \
                acc0[i_] -= 8 * (int)((rest_packed[0][i_ >> 2] >> (8 * (i_ & 3))) & 0xFFu);                                    \
                acc1[i_] -= 8 * (int)((rest_packed[1][i_ >> 2] >> (8 * (i_ & 3))) & 0xFFu);                                    \
            }                                                                                                                  \
            _Pragma("unroll") for (int m_ = kEarly; m_ < NW; ++m_) {                                                           \
                FTK_EXPAND_WORD(set_, m_)                                                                                      \
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][m_], b_, acc0, 0, 0, 0);                                      \
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[1][m_], b_, acc1, 0, 0, 0);                                      \
            }                                                                                                                  \
            mfma_update_keys<kNearby>(p, acc0, acc1, gtop_, key, pred_rows, lowest, at_lowest, h, min((tile_begin_) + c, last), j_begin); \
            FTK_SLOW_END                                                                                                       \
        }                                                                                                                      \
    }
#ifdef FTK_MATCH_STAMPS
    const unsigned long long st_prologue = __builtin_amdgcn_s_memtime() - st_c0;
#endif
    for (int tile_begin = j_begin; tile_begin < j_end; tile_begin += 64) {
        FTK_FETCH_WORDS(1, tile_begin + 32)
        __builtin_amdgcn_sched_barrier(0);  // the loads go out BEFORE the products of the tile in hand (the scheduler sinks them otherwise)
        FTK_MFMA_TILE(0, tile_begin)
        if (tile_begin + 32 < j_end) {
            FTK_FETCH_WORDS(0, tile_begin + 64)
            __builtin_amdgcn_sched_barrier(0);
            FTK_MFMA_TILE(1, tile_begin + 32)
        }
    }
#ifdef FTK_MATCH_STAMPS
    const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // ---- the best candidate of every row: maximum over the 32 lanes of the half, then one lane per row merges across splits ----
    // A register in which no lane of the wave found anything — nearly all of them under a real threshold — is skipped; with up
    // to eight finders per register every one of them merges its own key (atomicMin takes the smallest distance, then the
    // lowest index, whoever sends it); only beyond that is the register first reduced over the lanes.
    int final_keys[32];  // all reads in flight before the first is looked at
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        final_keys[s] = key[64 * s];
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        int k = final_keys[s];
        const unsigned long long finders = __ballot((k & 0xFFFF) != 0xFFFF);
        if (finders == 0ull) {
            continue;  // wave-uniform
        }
        bool send = (k & 0xFFFF) != 0xFFFF;
        if (__popcll(finders) > 8) {
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                k = max(k, __shfl_xor(k, off));
            }
            send = c == s;
        }
        const int row = row0 + (s >> 4) * 32 + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h;
        const int d = -(k >> 16) / 8;
        // `distance < min_distance && distance < threshold` with min_distance starting at the threshold
        if (send && row < p.n_ref && (k & 0xFFFF) != 0xFFFF && (float)d < p.max_distance) {
            const unsigned long long packed = ((unsigned long long)(unsigned)d << 32) | (unsigned)(j_begin + 0xFFFE - (k & 0xFFFF));
            atomicMin(&p.keys[row], packed);
        }
    }
#ifdef FTK_MATCH_STAMPS
    if (threadIdx.x == 0 && p.stamps) {
        unsigned long long *st = p.stamps + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        st[0] = st_rt0;
        st[1] = __builtin_amdgcn_s_memrealtime();
        st[2] = st_prologue;
        st[3] = st_loop_end - st_c0 - st_prologue;  // the tile loop
        st[4] = st_slow;
        st[5] = st_slow_n;
        st[6] = __builtin_amdgcn_s_memtime() - st_loop_end;  // the final reduction
        st[7] = __builtin_amdgcn_s_memtime() - st_c0;
    }
#endif
}

#undef FTK_FETCH_WORDS
#undef FTK_EXPAND_WORD
#undef FTK_MFMA_TILE
#undef FTK_SLOW_BEGIN
#undef FTK_SLOW_END

template <int NW, bool kNearby>
__global__ void __launch_bounds__(64) hamming_match_mfma_kernel(const MatchParams p) {
    static_assert(NW % 4 == 0, "whole 16-byte loads of a candidate");
    const int lane = (int)threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    const int row0 = (int)blockIdx.x * kMfmaRows;  // this wave's 64 rows
    const int j_begin = (int)blockIdx.y * p.cur_per_block;
    const int j_end = min(j_begin + p.cur_per_block, p.n_cur);
    bool out_of_reach = false;
    if (kNearby && p.boxes != nullptr) {
        // wave-uniform early exit on the bounding boxes (hamming_match_tiled_kernel); row boxes cover 256 * kMatchRefs rows
        const int n_row_boxes = (p.n_ref + kBlock * kMatchRefs - 1) / (kBlock * kMatchRefs);
        const float4 pb = p.boxes[((int)blockIdx.x * kMfmaRows) / (kBlock * kMatchRefs)], cb = p.boxes[n_row_boxes + blockIdx.y];
        const float reach_u = p.max_col + 1.0f, reach_v = p.max_row + 1.0f;
        out_of_reach = cb.x - pb.y > reach_u || pb.x - cb.y > reach_u || cb.z - pb.w > reach_v || pb.z - cb.w > reach_v;
    }
    if (!out_of_reach) {
        mfma_scan<NW, kNearby>(p, lane, c, h, row0, j_begin, j_end);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Small calls — the sizes the reference's own programs match (<= 300 features: test_descriptor_matcher_brief.cpp:59-65) — are
// bound by their LAUNCHES, not by their pairs: boxes + scan + epilogue are three dependent launches of 4 us for 90 000 pairs.
// Here the whole call is ONE launch with no workspace: a wave owns a reference row, its lanes share the candidates (lane l takes
// j = l, l + 64, ...: coalesced 16-byte loads), each keeps the minimum of the packed key (distance << 20 | j) — smallest distance,
// then lowest j: the reference's strict '<' scan (descriptor_matcher.h:68-75) — the wave reduces the 64 keys and lane 0 writes
// index_pairs[row] if a match exists (untouched otherwise, :60-62).  NearbyMatch applies the window per pair (:108-111).
// ---------------------------------------------------------------------------------------------------------------------------
// Where the one-launch form wins (scripts/match_small_ab.py, event-bracketed calls, launches / one launch): 300 x 300 x 256 bits
// 10.4 / 6.3 us, 1000 x 1000 10.2 / 7.0, 2000 x 2000 11.7 / 9.7, 3000 x 300 11.0 / 6.9, 300 x 3000 10.0 / 8.9 — and where it does not:
// 2000 x 2000 x 512 15.9 / 20.4, 300 x 3000 x 512 10.6 / 15.2, 64 x 60 000 10.3 / 73 (a wave walks its row's candidates alone).
constexpr long long kSmallMatchWork = 32ll << 20;  // n_ref * n_cur * n_words up to which the one-launch form is used ...
constexpr int kSmallMatchRowWork = 24576;          // ... while a row's walk stays below n_cur * n_words = this
constexpr int kSmallNoIndex = 0xFFFFF;

template <int NW, bool kNearby>
__global__ void __launch_bounds__(kBlock) hamming_match_small_kernel(const MatchParams p) {
    const int lane = (int)threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (kBlock / 64) + ((int)threadIdx.x >> 6));
    if (row >= p.n_ref) {
        return;
    }
    uint32_t ref[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        ref[w] = p.ref_words[(long long)row * NW + w];  // wave-uniform: scalar loads
    }
    float pred_u = 0.0f, pred_v = 0.0f;
    if (kNearby) {
        pred_u = p.pred_uv[2 * (long long)row];
        pred_v = p.pred_uv[2 * (long long)row + 1];
    }
    // d < limit0 is the conservative integer form of `distance < kMaxValidDescriptorDistance` (the exact test closes the scan)
    const int limit0 = (p.max_distance >= 0.0f && p.max_distance < 1000.0f) ? (int)p.max_distance + 1 : (p.max_distance < 0.0f ? 0 : 1023);
    uint32_t best = ((uint32_t)limit0 << 20) | (uint32_t)kSmallNoIndex;
#pragma unroll 2
    for (int j = lane; j < p.n_cur; j += 64) {
        uint32_t cur[NW];
        if constexpr (NW % 4 == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(p.cur_words + (long long)j * NW);
#pragma unroll
            for (int q = 0; q < NW / 4; ++q) {
                const uint4 four = src[q];
                cur[4 * q] = four.x;
                cur[4 * q + 1] = four.y;
                cur[4 * q + 2] = four.z;
                cur[4 * q + 3] = four.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                cur[w] = p.cur_words[(long long)j * NW + w];
            }
        }
        bool in_window = true;
        if (kNearby) {
            const float2 c = reinterpret_cast<const float2 *>(p.cur_uv)[j];
            in_window = !(fabsf(pred_u - c.x) > p.max_col || fabsf(pred_v - c.y) > p.max_row);
        }
        uint32_t d = __popc(ref[0] ^ cur[0]);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            d += __popc(ref[w] ^ cur[w]);
        }
        const uint32_t key = (d << 20) | (uint32_t)j;
        best = in_window ? min(best, key) : best;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        best = min(best, (uint32_t)__shfl_xor((int)best, off));
    }
    const int j_best = (int)(best & (uint32_t)kSmallNoIndex);
    // `distance < min_distance && distance < threshold` with min_distance starting at the threshold
    if (lane == 0 && j_best != kSmallNoIndex && (float)(best >> 20) < p.max_distance) {
        p.index_pairs[row] = j_best;
    }
}

template <int NW>
hipError_t launch_small(const MatchParams &p, hipStream_t stream) {
    const dim3 grid((unsigned)((p.n_ref + kBlock / 64 - 1) / (kBlock / 64)));
    if (p.pred_uv) {
        hipLaunchKernelGGL((hamming_match_small_kernel<NW, true>), grid, dim3(kBlock), 0, stream, p);
    } else {
        hipLaunchKernelGGL((hamming_match_small_kernel<NW, false>), grid, dim3(kBlock), 0, stream, p);
    }
    return hipGetLastError();
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
    if (p.matrix_cores) {
        if constexpr (NW >= 8) {
            if (p.pred_uv && p.boxes) {
                hipLaunchKernelGGL(hamming_box_kernel, dim3((unsigned)(row_blocks + splits)), dim3(kBlock), 0, stream, p, row_blocks);
            }
            const dim3 grid((unsigned)((p.n_ref + kMfmaRows - 1) / kMfmaRows), (unsigned)splits);
            if (p.pred_uv) {
                hipLaunchKernelGGL((hamming_match_mfma_kernel<NW, true>), grid, dim3(64), 0, stream, p);
            } else {
                hipLaunchKernelGGL((hamming_match_mfma_kernel<NW, false>), grid, dim3(64), 0, stream, p);
            }
            return hipGetLastError();
        }
    }
    if (p.pred_uv) {
        if (p.boxes) {
            hipLaunchKernelGGL(hamming_box_kernel, dim3((unsigned)(row_blocks + splits)), dim3(kBlock), 0, stream, p, row_blocks);
        }
        if (p.lds_tiles) {
            hipLaunchKernelGGL((hamming_match_tiled_kernel<NW, true>), dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
        } else {
            hipLaunchKernelGGL((hamming_match_scalar_kernel<NW, true>), dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
        }
    } else if (p.lds_tiles) {
        hipLaunchKernelGGL((hamming_match_tiled_kernel<NW, false>), dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
    } else {
        hipLaunchKernelGGL((hamming_match_scalar_kernel<NW, false>), dim3(row_blocks, splits), dim3(kBlock), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace

bool match_small_form(int n_ref, int n_cur, int n_words, int n_bits, bool small_off) {
    const bool allowed = !small_off;  // FTK_MATCH_SMALL=0 (experiment switch of the context; scripts/match_small_ab.py flips it)
    const bool width = n_words == 1 || n_words == 2 || n_words == 4 || n_words == 8 || n_words == 16;
    return allowed && width && n_bits > 0 && n_cur < kSmallNoIndex && (long long)n_cur * n_words <= kSmallMatchRowWork &&
           (long long)n_ref * n_cur * n_words <= kSmallMatchWork;
}

hipError_t match_launch(const MatchParams &p, hipStream_t stream) {
    if (p.n_ref <= 0 || p.n_cur <= 0) {
        return hipSuccess;
    }
    hipError_t e = hipSuccess;
    if (match_small_form(p.n_ref, p.n_cur, p.n_words, p.n_bits, p.small_off != 0)) {
        switch (p.n_words) {
            case 1: return launch_small<1>(p, stream);
            case 2: return launch_small<2>(p, stream);
            case 4: return launch_small<4>(p, stream);
            case 8: return launch_small<8>(p, stream);
            default: return launch_small<16>(p, stream);
        }
    }
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
        default: {
            const int splits = (p.n_cur + p.cur_per_block - 1) / p.cur_per_block;
            hipLaunchKernelGGL(hamming_match_generic_kernel, dim3((p.n_ref + kBlock - 1) / kBlock, splits), dim3(kBlock), 0, stream, p);
            e = hipGetLastError();
            break;
        }
    }
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(match_epilogue_kernel, dim3((p.n_ref + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, p.keys, p.index_pairs, p.n_ref);
    return hipGetLastError();
}

// First-use cost out of the callers' timed regions (ftk_warmup): launching this empty kernel makes the runtime load this
// translation unit's code object onto the device, which otherwise happens inside the first real call.
__global__ void matcher_warm_kernel() {}
hipError_t matcher_warm(hipStream_t stream) {
    hipLaunchKernelGGL(matcher_warm_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace ftk
