// float_matcher_kernels.hip — brute-force / windowed matcher for FLOAT descriptors (SuperPoint-256,
// DISK-128) on gfx950: SURVEY.md section 8(f) rank 3, the one dense contraction of the path.
//
// DescriptorMatcher<T>::ForceMatch / NearbyMatch (descriptor_matcher.h:55-79, :90-124) with the
// distance the reference's float-descriptor callers define
// (test/test_descriptor_matcher_superpoint.cpp:32-34, test_descriptor_matcher_disk.cpp:32-34):
//
//     d(i, j) = 0.5f - ref_i.dot(cur_j) / ref_i.norm() / cur_j.norm() * 0.5f          (fp32, Eigen)
//
// The result is an index per ref row and must be the index the scalar loop picks, so every
// comparison that decides it is made on d evaluated EXACTLY as the scalar code does (Eigen's
// SSE2 reduction order, no FMA, correctly rounded divide / sqrt).  The N x M x D contraction is
// only used to find out which pairs can possibly decide:
//
//   1. cosine_prep_kernel     per descriptor: exact Eigen norm; the row scaled to unit length and
//                             rounded to fp16 (rows / K zero-padded to the tile grid).
//   2. cosine_gemm_kernel<0>  MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) over all pairs:
//                             approximate cosine c~ = <x_i, y_j> (d~ = 0.5 - 0.5 c~), window test,
//                             per-row maximum (atomicMax on an order-preserving key).
//   3. cosine_gemm_kernel<1>  the same contraction again; pairs with c~ >= rowmax - 2 * margin
//                             (i.e. d~ <= min d~ + margin) are appended to the row's candidate list.
//   4. cosine_recheck_kernel  8 lanes per ref row: exact d of every candidate in Eigen's order,
//                             minimum with the lowest index on ties, strict threshold test, write.
//
// |d~ - d| <= eps for every "regular" pair (both norms in [2^-40, 2^40]): fp16 rounding of unit
// vectors perturbs the cosine by at most 2^-10 * sum|x_k y_k| <= 2^-10 (Cauchy-Schwarz) plus
// 256 * 2^-24 of subnormal fp16 components, fp32 accumulation adds < 4e-5, the exact side's own
// rounding < 3e-5 — so eps < 5.5e-4 on the distance and margin = 2 * eps bounds the set {j : d(i, j)
// can equal min_j d(i, j)}.  kMargin = 1.5e-3 leaves > 25 % slack.  Descriptors outside the regular
// range (zero / non-finite / extreme norm) never enter the approximation: an irregular ref row, a
// row whose candidate list overflowed, or more irregular cur rows than the side list holds sends
// that row through the exact scan over every j in step 4 (same code, longer list).
//
// For dim <= 256 steps 2 and 3 are ONE walk over the candidates (cosine_gemm_rr_kernel, default, and
// cosine_gemm_rs_kernel<2>): a running row maximum replaces the known one, which only makes the list a
// superset; entries carry their approximate score and step 4 keeps those within 2 * margin of the final
// maximum.  The rr kernel also writes the accumulator element's index into the 5 low mantissa bits of the
// score (so a maximum names its own element): a perturbation of at most 31 ulp < 2e-6 of the cosine, i.e.
// 1e-6 on the distance — eps stays below 5.6e-4 and 2 * eps below kMargin with > 25 % to spare.
#include <limits.h>
#include <stdlib.h>

#include "ftk_device.h"

namespace ftk {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int kTile = 128;     // rows of one operand tile (cur: MFMA rows, ref: MFMA columns)
constexpr int kChunkK = 64;    // K staged per LDS chunk
constexpr int kPitch = kChunkK + 8;  // halfs; +16 B keeps the 128-bit fragment reads off one bank group
constexpr float kMargin = 1.5e-3f;
constexpr float kNormLo = 9.094947017729282e-13f;  // 2^-40
constexpr float kNormHi = 1.099511627776e12f;      // 2^40
// Candidates per row up to which a call runs as ONE exact launch (cosine_match_small_kernel: a wave walks its row's candidates
// alone, so its time grows with n_cur); measured, scripts/cosine_small_ab.py
constexpr int kCosineSmallCurNearby = 2048;  // 300 x 300 x 256 NearbyMatch 48.7 -> 10.7 us, 1 000 x 1 000 47.4 -> 21.3, 2 000 x 2 000 54.6 -> 40.3 (3 000 candidates: even)
constexpr int kCosineSmallCurForce = 384;    // ForceMatch computes every pair exactly: 100 x 100 x 256 35.4 -> 14.8 us, 300 x 300 40.3 -> 32.1, 600 x 600 41.6 -> 59.2 (not taken)
constexpr int kCosineSmallRefMax = 4096;

__device__ __forceinline__ uint32_t order_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float order_value(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

// x.dot(y) in Eigen 3.3.7's order for SSE2 packets (Core/Redux.h, LinearVectorizedTraversal /
// NoUnrolling; oracle/oracle_float_matcher.c states it in scalar form), evaluated by an octet of
// lanes: lane c = 4 * a + q runs lane q of packet accumulator a.  All 8 lanes of the octet must
// call; every lane returns the result.  `base` = wave lane index of the octet's lane 0.
__device__ __forceinline__ float eigen_dot_octet(const float *__restrict__ x, const float *__restrict__ y, int size, int c, int base) {
    const int aligned_size = (size / 4) * 4;
    const int aligned_end2 = (size / 8) * 8;
    if (aligned_size == 0) {
        float res = x[0] * y[0];
        for (int index = 1; index < size; ++index) {
            res = res + x[index] * y[index];
        }
        return res;
    }
    const int q = c & 3, a = c >> 2;
    const int off = 4 * a + q;
    float acc = 0.0f;
    if (a == 0 || aligned_size > 4) {
        acc = x[off] * y[off];
#pragma unroll 8
        for (int index = 8; index < aligned_end2; index += 8) {  // eight steps of loads in flight (a rolled loop waits for each pair); adds in index order
            acc = acc + x[index + off] * y[index + off];
        }
    }
    float p = acc;  // meaningful in lanes a == 0
    if (aligned_size > 4) {
        const float p1 = __shfl(acc, base + 4 + q);
        p = acc + p1;
        if (aligned_size > aligned_end2) {
            p = p + x[aligned_end2 + q] * y[aligned_end2 + q];
        }
    }
    const float p0 = __shfl(p, base + 0), p1v = __shfl(p, base + 1), p2 = __shfl(p, base + 2), p3 = __shfl(p, base + 3);
    float res = (p0 + p2) + (p1v + p3);  // SSE2 predux<Packet4f>
    for (int index = aligned_size; index < size; ++index) {
        res = res + x[index] * y[index];
    }
    return res;
}

__device__ __forceinline__ void cosine_prep_row_outputs(const CosineParams &p, int which, int row, int n, float norm, bool regular) {
    if (which) {
        const float bias = regular ? 0.0f : __uint_as_float(0xFF800000u);  // -inf keeps the column out of every maximum
        p.cur_bias[row] = bias;
        if (p.cur_info) {
            const bool windowed = p.pred_uv != nullptr && row < n;
            // finite "never": cosine_gemm_rr_kernel writes index bits into the score and -inf would turn into a NaN
            p.cur_info[row] = make_float4(regular ? 0.0f : -3.0e38f, windowed ? p.cur_uv[2 * row] : 0.0f, windowed ? p.cur_uv[2 * row + 1] : 0.0f, 0.0f);
        }
        if (row < n) {
            p.cur_norm[row] = norm;
            if (!regular) {
                const uint32_t slot = atomicAdd(p.irregular_count, 1u);
                if (slot < (uint32_t)kCosineIrregularCap) {
                    p.irregular_list[slot] = row;
                }
            }
        }
    } else if (row < n) {
        p.ref_norm[row] = norm;
        p.ref_irregular[row] = regular ? 0 : 1;
    }
}

// ---- 1. norms + unit-length fp16 copies ------------------------------------------------------
// One octet per (padded) row of one operand; blockIdx.y == 0: ref, 1: cur.
__global__ void __launch_bounds__(256) cosine_prep_kernel(const CosineParams p) {
    const int which = (int)blockIdx.y;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = tid >> 3, c = tid & 7, base = (threadIdx.x & 63) & ~7;
    const int n = which ? p.n_cur : p.n_ref;
    const int n_pad = which ? p.n_cur_pad : p.n_ref_pad;
    if (row >= n_pad) {
        return;  // whole octets leave together (n_pad * 8 is a multiple of the block size or the tail octets are complete)
    }
    const float *src = (which ? p.cur : p.ref) + (size_t)(row < n ? row : 0) * p.dim;
    _Float16 *dst = (which ? p.cur_h : p.ref_h) + (size_t)row * p.dim_pad;
    float norm = 0.0f;
    bool regular = false;
    if (row < n) {  // uniform within the octet
        norm = sqrtf(eigen_dot_octet(src, src, p.dim, c, base));  // Eigen norm() = sqrt(squaredNorm())
        regular = norm >= kNormLo && norm <= kNormHi;             // false for NaN
    }
    const float inv = regular ? 1.0f / norm : 0.0f;
    if ((p.dim & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        // four elements per lane and step: the octet reads 128 and writes 64 contiguous bytes (dst rows are 128-byte aligned)
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        for (int f = c; f < p.dim_pad / 4; f += 8) {
            float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (regular && 4 * f < p.dim) {
                x = src4[f];
            }
            half4 h;
            h[0] = (_Float16)(x.x * inv);
            h[1] = (_Float16)(x.y * inv);
            h[2] = (_Float16)(x.z * inv);
            h[3] = (_Float16)(x.w * inv);
            *reinterpret_cast<half4 *>(dst + 4 * f) = h;
        }
    } else {
        for (int k = c; k < p.dim_pad; k += 8) {
            const float v = (regular && k < p.dim) ? src[k] * inv : 0.0f;
            dst[k] = (_Float16)v;
        }
    }
    if (c == 0) {
        cosine_prep_row_outputs(p, which, row, n, norm, regular);
    }
}

// Same result for the common layouts (dim a multiple of 8, rows 16-byte aligned: SuperPoint-256, DISK-128), TWO lanes
// per row: lane a IS packet accumulator a of Eigen's reduction and holds its four components as a float4, so every
// load is a whole 16-byte packet (the octet form reads 4 bytes per lane) and a wave covers 32 rows.
// squaredNorm = predux(acc0 + acc1) with acc_a = sum over i of packet(2 i + a)^2, each component in index order.
__global__ void __launch_bounds__(256) cosine_prep_pair_kernel(const CosineParams p) {
    const int which = (int)blockIdx.y;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = tid >> 1, a = tid & 1;
    const int n = which ? p.n_cur : p.n_ref;
    const int n_pad = which ? p.n_cur_pad : p.n_ref_pad;
    if (row >= n_pad) {
        return;  // pairs leave together
    }
    const float4 *src4 = reinterpret_cast<const float4 *>((which ? p.cur : p.ref) + (size_t)(row < n ? row : 0) * p.dim);
    _Float16 *dst = (which ? p.cur_h : p.ref_h) + (size_t)row * p.dim_pad;
    const int packets = p.dim / 4;  // even
    float norm = 0.0f;
    bool regular = false;
    if (row < n) {  // uniform within the pair
        float4 x = src4[a];
        float4 acc = make_float4(x.x * x.x, x.y * x.y, x.z * x.z, x.w * x.w);
#pragma unroll 8
        for (int f = 2 + a; f < packets; f += 2) {  // loads of eight steps in flight; the adds stay in index order
            x = src4[f];
            acc.x = acc.x + x.x * x.x;
            acc.y = acc.y + x.y * x.y;
            acc.z = acc.z + x.z * x.z;
            acc.w = acc.w + x.w * x.w;
        }
        // acc0 + acc1 (lane a = 0 adds its partner's), then SSE2 predux (p0 + p2) + (p1 + p3)
        const float ox = __shfl_xor(acc.x, 1), oy = __shfl_xor(acc.y, 1), oz = __shfl_xor(acc.z, 1), ow = __shfl_xor(acc.w, 1);
        const float p0 = a ? ox + acc.x : acc.x + ox, p1 = a ? oy + acc.y : acc.y + oy;
        const float p2 = a ? oz + acc.z : acc.z + oz, p3 = a ? ow + acc.w : acc.w + ow;
        norm = sqrtf((p0 + p2) + (p1 + p3));
        regular = norm >= kNormLo && norm <= kNormHi;
    }
    const float inv = regular ? 1.0f / norm : 0.0f;
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    // eight packets per lane and round: all loads first (unconditional, from a valid packet — a load under a branch, or
    // behind a store the compiler cannot prove disjoint, is waited for one at a time), then the stores
    const int packets_pad = p.dim_pad / 4;
    for (int f0 = a; f0 < packets_pad; f0 += 16) {
        float4 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = f0 + 2 * j;
            x[j] = src4[f < packets ? f : packets - 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = f0 + 2 * j;
            const bool keep = regular && f < packets;
            half4 h;
            h[0] = (_Float16)(keep ? x[j].x * inv : 0.0f);
            h[1] = (_Float16)(keep ? x[j].y * inv : 0.0f);
            h[2] = (_Float16)(keep ? x[j].z * inv : 0.0f);
            h[3] = (_Float16)(keep ? x[j].w * inv : 0.0f);
            if (f < packets_pad) {
                *reinterpret_cast<half4 *>(dst + 4 * f) = h;
            }
        }
    }
    if (a == 0) {
        cosine_prep_row_outputs(p, which, row, n, norm, regular);
    }
    // NearbyMatch: the bounding boxes of the 64-row cur tiles (see cosine_tile_box_kernel) ride along — a block holds two
    // tiles, a wave half of one.  Waves past the padded end have left above, whole tiles at a time.
    if (which == 1 && p.tile_box != nullptr) {  // block-uniform
        __shared__ float box_part[4][5];
        const float pos_inf = __uint_as_float(0x7F800000u), neg_inf = __uint_as_float(0xFF800000u);
        float u0 = pos_inf, u1 = neg_inf, v0 = pos_inf, v1 = neg_inf;
        bool unordered = false;
        if (row < n) {
            const float u = p.cur_uv[2 * row], v = p.cur_uv[2 * row + 1];
            unordered = isnan(u) || isnan(v);
            if (!unordered) {
                u0 = u1 = u;
                v0 = v1 = v;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            u0 = fminf(u0, __shfl_xor(u0, off));
            u1 = fmaxf(u1, __shfl_xor(u1, off));
            v0 = fminf(v0, __shfl_xor(v0, off));
            v1 = fmaxf(v1, __shfl_xor(v1, off));
        }
        const int w = (int)threadIdx.x >> 6;
        const bool any_unordered = __ballot(unordered) != 0ull;
        if ((threadIdx.x & 63) == 0) {
            box_part[w][0] = u0;
            box_part[w][1] = u1;
            box_part[w][2] = v0;
            box_part[w][3] = v1;
            box_part[w][4] = any_unordered ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (threadIdx.x < 2 && ((int)blockIdx.x * 2 + (int)threadIdx.x) * 64 < n_pad) {
            const int t = (int)threadIdx.x, wa = 2 * t, wb = 2 * t + 1;
            float4 bx = make_float4(fminf(box_part[wa][0], box_part[wb][0]), fmaxf(box_part[wa][1], box_part[wb][1]),
                                    fminf(box_part[wa][2], box_part[wb][2]), fmaxf(box_part[wa][3], box_part[wb][3]));
            if (box_part[wa][4] != 0.0f || box_part[wb][4] != 0.0f) {
                bx = make_float4(neg_inf, pos_inf, neg_inf, pos_inf);
            }
            p.tile_box[(int)blockIdx.x * 2 + t] = bx;
        }
    }
}

// ---- 2 / 3. the contraction --------------------------------------------------------------------
// Output tile 128 (cur j, MFMA rows) x 128 (ref i, MFMA columns) per workgroup pass, 4 waves as
// 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32.  The ref row is the accumulator's lane
// (col = lane & 31) and the 16 accumulator registers are 16 different candidates j, so the
// per-row minimum is a chain of v_min in registers plus one cross-half shuffle at the very end.
template <bool kCollect, bool kNearby>
__global__ void __launch_bounds__(256) cosine_gemm_kernel(const CosineParams p) {
    __shared__ __attribute__((aligned(16))) _Float16 sX[kTile * kPitch];  // cur rows of the chunk
    __shared__ __attribute__((aligned(16))) _Float16 sY[kTile * kPitch];  // ref rows of the chunk
    __shared__ float4 sInfo[kTile];                                        // per cur row: {bias, u, v, -}

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.x * kTile;
    const int tiles_total = p.n_cur_pad / kTile;
    const int jt_begin = blockIdx.y * p.tiles_per_split;
    const int jt_end = min(jt_begin + p.tiles_per_split, tiles_total);
    if (jt_begin >= jt_end) {
        return;
    }
    const int n_chunks = p.dim_pad / kChunkK;
    constexpr bool nearby = kNearby;
    const float pos_inf = __uint_as_float(0x7F800000u), neg_inf = __uint_as_float(0xFF800000u);

    int row_i[2];
    bool live[2];
    float pu[2], pv[2], thr[2], best[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int i = i0 + wn * 64 + nt * 32 + (lane & 31);
        row_i[nt] = i;
        live[nt] = i < p.n_ref && p.ref_irregular[i] == 0;
        pu[nt] = (nearby && i < p.n_ref) ? p.pred_uv[2 * i] : 0.0f;
        pv[nt] = (nearby && i < p.n_ref) ? p.pred_uv[2 * i + 1] : 0.0f;
        best[nt] = neg_inf;
        thr[nt] = pos_inf;
        if (kCollect && live[nt]) {
            const uint32_t key = p.row_max[i];
            thr[nt] = (key == 0u) ? pos_inf : order_value(key) - 2.0f * kMargin;  // cosine domain: d = 0.5 - 0.5 c
        }
    }

    // staging: thread t carries 64 B (32 halfs) of row t / 2 of each operand chunk, in eight named
    // registers (an indexed array captured by a lambda ends up in scratch memory)
    const int srow = tid >> 1, scol = (tid & 1) * 32;
    uint4 rx0, rx1, rx2, rx3, ry0, ry1, ry2, ry3;
    const _Float16 *const gx_base = p.cur_h + (size_t)srow * p.dim_pad + scol;
    const _Float16 *const gy_base = p.ref_h + (size_t)(i0 + srow) * p.dim_pad + scol;
#define FTK_LOAD_CHUNK(jt_, kc_)                                                                                          \
    do {                                                                                                                  \
        const uint4 *gx = reinterpret_cast<const uint4 *>(gx_base + (size_t)(jt_) * kTile * p.dim_pad + (kc_) * kChunkK); \
        const uint4 *gy = reinterpret_cast<const uint4 *>(gy_base + (kc_) * kChunkK);                                     \
        rx0 = gx[0];                                                                                                      \
        rx1 = gx[1];                                                                                                      \
        rx2 = gx[2];                                                                                                      \
        rx3 = gx[3];                                                                                                      \
        ry0 = gy[0];                                                                                                      \
        ry1 = gy[1];                                                                                                      \
        ry2 = gy[2];                                                                                                      \
        ry3 = gy[3];                                                                                                      \
    } while (0)
    FTK_LOAD_CHUNK(jt_begin, 0);

    for (int jt = jt_begin; jt < jt_end; ++jt) {
        const int j0 = jt * kTile;
        float16v acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[mt][nt][r] = 0.0f;
                }
            }
        }
        for (int kc = 0; kc < n_chunks; ++kc) {
            __syncthreads();  // everyone is done reading the previous chunk (and the previous tile's sInfo)
            {
                uint4 *dx = reinterpret_cast<uint4 *>(&sX[srow * kPitch + scol]);
                uint4 *dy = reinterpret_cast<uint4 *>(&sY[srow * kPitch + scol]);
                dx[0] = rx0;
                dx[1] = rx1;
                dx[2] = rx2;
                dx[3] = rx3;
                dy[0] = ry0;
                dy[1] = ry1;
                dy[2] = ry2;
                dy[3] = ry3;
            }
            if (kc == 0 && tid < kTile) {
                const int j = j0 + tid;
                float4 info = make_float4(p.cur_bias[j], 0.0f, 0.0f, 0.0f);
                if (nearby && j < p.n_cur) {
                    info.y = p.cur_uv[2 * j];
                    info.z = p.cur_uv[2 * j + 1];
                }
                sInfo[tid] = info;
            }
            __syncthreads();
            if (kc + 1 < n_chunks) {
                FTK_LOAD_CHUNK(jt, kc + 1);
            } else if (jt + 1 < jt_end) {
                FTK_LOAD_CHUNK(jt + 1, 0);
            }
#pragma unroll
            for (int kk = 0; kk < kChunkK / 16; ++kk) {
                half8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = *reinterpret_cast<const half8 *>(&sX[(wm * 64 + t * 32 + (lane & 31)) * kPitch + kk * 16 + 8 * (lane >> 5)]);
                    b[t] = *reinterpret_cast<const half8 *>(&sY[(wn * 64 + t * 32 + (lane & 31)) * kPitch + kk * 16 + 8 * (lane >> 5)]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue: C/D map of the 32x32 MFMA — col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float4 info = sInfo[jl];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v = acc[mt][nt][r] + info.x;  // bias: 0, or -inf for padding / irregular candidates
                    if (kNearby) {
                        // window test of descriptor_matcher.h:108-111, branch-free
                        const bool out = (int)(fabsf(pu[nt] - info.y) > p.max_col) | (int)(fabsf(pv[nt] - info.z) > p.max_row);
                        v = out ? neg_inf : v;
                    }
                    if (!kCollect) {
                        best[nt] = fmaxf(best[nt], v);
                    } else if (v >= thr[nt]) {
                        const uint32_t slot = atomicAdd(&p.cand_count[row_i[nt]], 1u);
                        if (slot < (uint32_t)kCosineCandCap) {
                            p.cand[(size_t)row_i[nt] * kCosineCandCap + slot] = j0 + jl;
                        }
                    }
                }
            }
        }
    }
    if (!kCollect) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float other = __shfl_xor(best[nt], 32);
            const float m = fmaxf(best[nt], other);
            if (lane < 32 && live[nt] && m > neg_inf) {
                atomicMax(&p.row_max[row_i[nt]], order_key(m));
            }
        }
    }
}

#undef FTK_LOAD_CHUNK

// ---- 2 / 3 (dim_pad <= 256): ref-stationary contraction ----------------------------------------
// rocprofv3 on the kernel above (10 000^2 x 256): MFMA busy 25 %, 41 % of the wave time waiting on
// memory, a third of the L2 lookups missing — each workgroup re-fetches its ref chunk for every cur
// tile and has a prefetch distance of one 64-wide chunk (~500 cycles) against L2-miss latency.
// Here a workgroup of 8 waves keeps its 128 ref rows for the WHOLE K in LDS (loaded once), streams
// cur chunks of 256 rows x 64 K through a double-buffered LDS tile with the global loads issued
// TWO chunks ahead (two named register sets), and pays one barrier per chunk.  Wave (wm, wn) of
// the 4 x 2 grid computes 64 cur x 64 ref = 2 x 2 MFMA tiles; the epilogue is unchanged.
//
// kMode 0 / 1 are steps 2 / 3 of the header (row maximum, then collection against it).  kMode 2 does
// both in ONE walk: the workgroup keeps a running maximum per ref row in LDS (sMax) and collects every
// pair within 2 * margin of the maximum seen SO FAR — a superset of the final list, because the running
// maximum never exceeds the final one.  The first cur tile only feeds sMax and is walked a second time at
// the end, so collection never starts from an empty bound; after that a row adds entries only when a
// tile raises (or comes within the margin of) its maximum, ~ln(tiles) times per workgroup for any
// exchangeable order of the cur rows.  Each entry carries its approximate score; the recheck kernel
// drops those below the final row maximum - 2 * margin, so the exact work is that of the two-pass path.
// A row whose list overflows takes the exact scan, as before.
// Entries are staged in LDS (sStage: one ds_add for the slot, ~100 cycles) and appended to the per-row global lists
// after the walk, all at once: a returning global atomic per entry, waited for inside the epilogue, costs ~1.5 us each
// and a wave meets ~50 entries per walk — measured, that alone made the single walk slower than the two launches.
constexpr int kCurTile = 256;
constexpr int kStageCap = 1024;  // entries {ref row in the tile, cur row, score}; beyond it an entry goes to global directly

template <int kMode, bool kNearby>
__global__ void __launch_bounds__(512) cosine_gemm_rs_kernel(const CosineParams p) {
    constexpr bool kCollect = kMode == 1, kSingle = kMode == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    const int pitch_y = p.dim_pad + 8;                                         // halfs
    _Float16 *const sY = reinterpret_cast<_Float16 *>(rs_lds);                 // [128][pitch_y]
    _Float16 *const sX = sY + kTile * pitch_y;                                 // [2][256][kPitch]
    float4 *const sInfo = reinterpret_cast<float4 *>(sX + 2 * kCurTile * kPitch);  // [2][256]: {bias, u, v, -}
    uint32_t *const sMax = reinterpret_cast<uint32_t *>(sInfo + 2 * kCurTile);     // [128] running row maxima (kMode 2)
    uint32_t *const sStageCount = sMax + kTile;                                    // [4], first word used
    uint32_t *const sStage = sStageCount + 4;                                      // [kStageCap][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.x * kTile;
    const int tiles_total = p.n_cur_pad / kCurTile;
    const int jt_begin = blockIdx.y * p.tiles_per_split;
    const int jt_end = min(jt_begin + p.tiles_per_split, tiles_total);
    if (jt_begin >= jt_end) {
        return;
    }
    const int n_chunks = p.dim_pad / kChunkK;
    const int n_tiles = jt_end - jt_begin;
    const int n_steps = kSingle ? n_tiles + 1 : n_tiles;  // kMode 2: tile 0 once more at the end
    const int total_chunks = n_steps * n_chunks;
    if (kSingle && tid < kTile) {
        sMax[tid] = 0u;  // key 0 = nothing seen (visible after the first barrier of the walk)
        if (tid == 0) {
            sStageCount[0] = 0u;
        }
    }
    const float pos_inf = __uint_as_float(0x7F800000u), neg_inf = __uint_as_float(0xFF800000u);

    int row_i[2];
    bool live[2];
    float pu[2], pv[2], thr[2], best[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int i = i0 + wn * 64 + nt * 32 + (lane & 31);
        row_i[nt] = i;
        live[nt] = i < p.n_ref && p.ref_irregular[i] == 0;
        pu[nt] = (kNearby && i < p.n_ref) ? p.pred_uv[2 * i] : 0.0f;
        pv[nt] = (kNearby && i < p.n_ref) ? p.pred_uv[2 * i + 1] : 0.0f;
        best[nt] = neg_inf;
        thr[nt] = pos_inf;
        if (kCollect && live[nt]) {
            const uint32_t key = p.row_max[i];
            thr[nt] = (key == 0u) ? pos_inf : order_value(key) - 2.0f * kMargin;
        }
    }

    // ref rows, whole K, once: thread t copies dim_pad / 4 halfs of row t / 4
    {
        const int r = tid >> 2, part = tid & 3, span = p.dim_pad / 4;  // span is a multiple of 16 halfs (dim_pad % 64 == 0)
        const uint4 *src = reinterpret_cast<const uint4 *>(p.ref_h + (size_t)(i0 + r) * p.dim_pad + part * span);
        uint4 *dst = reinterpret_cast<uint4 *>(sY + r * pitch_y + part * span);
        for (int k = 0; k < span / 8; ++k) {
            dst[k] = src[k];
        }
    }

    // cur chunk staging: thread t carries 64 B (32 halfs) of row t / 2
    const int srow = tid >> 1, scol = (tid & 1) * 32;
    const _Float16 *const gx_base = p.cur_h + (size_t)srow * p.dim_pad + scol;
    // chunk index -> (tile, K chunk) without a runtime division (n_chunks = dim_pad / 64 <= 4; c < 2^16):
    // ceil(2^16 / d) is exact for these ranges
    const unsigned div_magic = (65536u + (unsigned)n_chunks - 1u) / (unsigned)n_chunks;
    auto tile_of = [&](int c) { return (int)(((unsigned)c * div_magic) >> 16); };
    auto chunk_src = [&](int c) {
        const int t_ = tile_of(c);
        const int jt = jt_begin + ((kSingle && t_ == n_tiles) ? 0 : t_), kc = c - t_ * n_chunks;
        return reinterpret_cast<const uint4 *>(gx_base + (size_t)jt * kCurTile * p.dim_pad + kc * kChunkK);
    };
    // two named register sets (an aggregate passed by reference into a lambda ends up in scratch memory)
    uint4 s0a, s0b, s0c, s0d, s1a, s1b, s1c, s1d;
#define FTK_RS_LOAD(A, B, C, D, c_)          \
    do {                                     \
        const uint4 *g_ = chunk_src(c_);     \
        A = g_[0];                           \
        B = g_[1];                           \
        C = g_[2];                           \
        D = g_[3];                           \
    } while (0)
#define FTK_RS_STORE(A, B, C, D, buf_)                                                                       \
    do {                                                                                                     \
        uint4 *d_ = reinterpret_cast<uint4 *>(sX + ((buf_) * kCurTile + srow) * kPitch + scol);              \
        d_[0] = A;                                                                                           \
        d_[1] = B;                                                                                           \
        d_[2] = C;                                                                                           \
        d_[3] = D;                                                                                           \
    } while (0)
    auto write_info = [&](int tile_index) {  // per-candidate data of the cur tile walked at step tile_index
        if (tid < kCurTile) {
            const int j = (jt_begin + ((kSingle && tile_index == n_tiles) ? 0 : tile_index)) * kCurTile + tid;
            float4 info = make_float4(p.cur_bias[j], 0.0f, 0.0f, 0.0f);
            if (kNearby && j < p.n_cur) {
                info.y = p.cur_uv[2 * j];
                info.z = p.cur_uv[2 * j + 1];
            }
            sInfo[(tile_index & 1) * kCurTile + tid] = info;
        }
    };

    // set (c & 1) carries chunk c between its load and its LDS store
    FTK_RS_LOAD(s0a, s0b, s0c, s0d, 0);
    FTK_RS_STORE(s0a, s0b, s0c, s0d, 0);
    write_info(0);
    if (total_chunks > 1) {
        FTK_RS_LOAD(s1a, s1b, s1c, s1d, 1);
    }
    if (total_chunks > 2) {
        FTK_RS_LOAD(s0a, s0b, s0c, s0d, 2);
    }

    float16v acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[mt][nt][r] = 0.0f;
                }
            }
        }
    };
    zero_acc();

    // one pipeline step: chunk c is in LDS buffer c & 1; the set named at the call site holds chunk c + 1 and is
    // refilled with chunk c + 3 (FTK_RS_STEP below); `compute` is the part that touches no staging register
    auto compute = [&](int c) {
        const int tile_index = tile_of(c), kc = c - tile_index * n_chunks;
        if (kc == 0 && tile_index + 1 < n_steps) {
            write_info(tile_index + 1);
        }
        const _Float16 *bx = sX + ((c & 1) * kCurTile) * kPitch;
        // fragments of K-step kk + 1 are read while the four MFMAs of step kk run (two register sets; the compiler
        // otherwise re-reads into the same registers right before each use and exposes the LDS latency four times a chunk)
        const _Float16 *ax0 = &bx[(wm * 64 + (lane & 31)) * kPitch + 8 * (lane >> 5)];
        const _Float16 *by0 = &sY[(wn * 64 + (lane & 31)) * pitch_y + kc * kChunkK + 8 * (lane >> 5)];
        half8 fa[2][2], fb[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[0][t] = *reinterpret_cast<const half8 *>(ax0 + t * 32 * kPitch);
            fb[0][t] = *reinterpret_cast<const half8 *>(by0 + t * 32 * pitch_y);
        }
#pragma unroll
        for (int kk = 0; kk < kChunkK / 16; ++kk) {
            const int cur_set = kk & 1, nxt_set = cur_set ^ 1;
            if (kk + 1 < kChunkK / 16) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[nxt_set][t] = *reinterpret_cast<const half8 *>(ax0 + t * 32 * kPitch + (kk + 1) * 16);
                    fb[nxt_set][t] = *reinterpret_cast<const half8 *>(by0 + t * 32 * pitch_y + (kk + 1) * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMAs: the scheduler sinks them otherwise
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_set][mt], fb[cur_set][nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        if (kc == n_chunks - 1) {
            // epilogue of this cur tile: C/D map of the 32x32 MFMA — col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            const int j0 = (jt_begin + ((kSingle && tile_index == n_tiles) ? 0 : tile_index)) * kCurTile;
            const float4 *info_tile = sInfo + (tile_index & 1) * kCurTile;
            // straight-line pass: bias, window, running maximum (per tile in collect mode, per launch otherwise)
            float tile_best[2] = {neg_inf, neg_inf};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float4 info = info_tile[jl];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        float v = acc[mt][nt][r] + info.x;
                        if (kNearby) {
                            const bool out = (int)(fabsf(pu[nt] - info.y) > p.max_col) | (int)(fabsf(pv[nt] - info.z) > p.max_row);
                            v = out ? neg_inf : v;
                        }
                        if (kCollect || kSingle) {
                            acc[mt][nt][r] = v;  // kept for the (rare) second look below
                        }
                        tile_best[nt] = fmaxf(tile_best[nt], v);
                    }
                }
            }
            if (kSingle) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int slot_row = wn * 64 + nt * 32 + (lane & 31);
                    const float mine = live[nt] ? tile_best[nt] : neg_inf;  // padding / irregular rows collect nothing
                    if (tile_index > 0 && mine > neg_inf) {
                        // bound = maximum over the tiles before this one (all waves, published before the last barrier)
                        // joined with this lane's share of the current tile
                        const uint32_t seen = sMax[slot_row];
                        const float bound = fmaxf(seen ? order_value(seen) : neg_inf, mine) - 2.0f * kMargin;
                        if (mine >= bound) {
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    if (acc[mt][nt][r] >= bound) {
                                        const int jl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                        const uint32_t at = atomicAdd(&sStageCount[0], 1u);
                                        if (at < (uint32_t)kStageCap) {
                                            sStage[3 * at] = (uint32_t)slot_row;
                                            sStage[3 * at + 1] = (uint32_t)(j0 + jl);
                                            sStage[3 * at + 2] = __float_as_uint(acc[mt][nt][r]);
                                        } else {
                                            const uint32_t slot = atomicAdd(&p.cand_count[row_i[nt]], 1u);
                                            if (slot < (uint32_t)kCosineCandCap) {
                                                p.cand[(size_t)row_i[nt] * kCosineCandCap + slot] = j0 + jl;
                                                p.cand_score[(size_t)row_i[nt] * kCosineCandCap + slot] = acc[mt][nt][r];
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                    if (tile_index < n_tiles && mine > neg_inf) {
                        atomicMax(&sMax[slot_row], order_key(mine));
                    }
                }
            } else if (!kCollect) {
                best[0] = fmaxf(best[0], tile_best[0]);
                best[1] = fmaxf(best[1], tile_best[1]);
            } else {
                // a tile holds a candidate for very few rows: one test per (lane, nt) instead of one branch per element
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (tile_best[nt] >= thr[nt]) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if (acc[mt][nt][r] >= thr[nt]) {
                                    const int jl = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                    const uint32_t slot = atomicAdd(&p.cand_count[row_i[nt]], 1u);
                                    if (slot < (uint32_t)kCosineCandCap) {
                                        p.cand[(size_t)row_i[nt] * kCosineCandCap + slot] = j0 + jl;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            zero_acc();
        }
    };

#define FTK_RS_STEP(c_, A, B, C, D)                                                                                       \
    do {                                                                                                                  \
        __syncthreads(); /* chunk c visible; everyone is done with chunk c - 1 (its buffer, the older sInfo half) */      \
        if ((c_) + 1 < total_chunks) {                                                                                    \
            FTK_RS_STORE(A, B, C, D, ((c_) + 1) & 1);                                                                     \
        }                                                                                                                 \
        if ((c_) + 3 < total_chunks) {                                                                                    \
            FTK_RS_LOAD(A, B, C, D, (c_) + 3);                                                                            \
        }                                                                                                                 \
        compute(c_);                                                                                                      \
    } while (0)
    for (int c = 0; c < total_chunks; c += 2) {
        FTK_RS_STEP(c, s1a, s1b, s1c, s1d);  // chunk c + 1 lives in set 1 (odd), refilled with chunk c + 3
        if (c + 1 < total_chunks) {
            FTK_RS_STEP(c + 1, s0a, s0b, s0c, s0d);  // chunk c + 2 lives in set 0 (even), refilled with chunk c + 4
        }
    }
#undef FTK_RS_STEP
#undef FTK_RS_LOAD
#undef FTK_RS_STORE
    if (kSingle) {
        __syncthreads();
        if (tid < kTile) {
            const int i = i0 + tid;
            const uint32_t key = sMax[tid];
            if (key != 0u && i < p.n_ref) {
                atomicMax(&p.row_max[i], key);
            }
        }
        const uint32_t staged = min(sStageCount[0], (uint32_t)kStageCap);
        for (uint32_t e = (uint32_t)tid; e < staged; e += 512u) {
            const int i = i0 + (int)sStage[3 * e];
            const uint32_t slot = atomicAdd(&p.cand_count[i], 1u);
            if (slot < (uint32_t)kCosineCandCap) {
                p.cand[(size_t)i * kCosineCandCap + slot] = (int32_t)sStage[3 * e + 1];
                p.cand_score[(size_t)i * kCosineCandCap + slot] = __uint_as_float(sStage[3 * e + 2]);
            }
        }
    } else if (!kCollect) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const float other = __shfl_xor(best[nt], 32);
            const float m = fmaxf(best[nt], other);
            if (lane < 32 && live[nt] && m > neg_inf) {
                atomicMax(&p.row_max[row_i[nt]], order_key(m));
            }
        }
    }
}

// ---- 2 + 3 (dim_pad <= 256), register-stationary: the ref fragments never leave the register file ----------
// rocprofv3 on the LDS-stationary kernel above (10 000^2 x 256, single walk): ~93 us, MFMA busy ~30 %.  Every wave
// reads one 1 KB fragment from LDS per MFMA (2 cur + 2 ref fragments for 2 x 2 tiles), which alone fills the LDS
// pipe at the MFMA peak rate, and every workgroup streams the whole cur slice through L2 -> LDS for only 128 ref rows.
// Here each of the 8 waves keeps ITS 64 ref rows for the whole K as MFMA B operands in registers (2 x dim_pad / 16
// fragments = 128 VGPRs at dim 256, loaded once), so a workgroup covers 512 ref rows; cur tiles of 64 rows x whole K
// stream through a double-buffered LDS tile that ALL waves read (2 fragment reads per 4 MFMAs — half the LDS traffic
// per flop, a quarter of the L2 -> LDS traffic per pair).  A ref row belongs to exactly one wave, so the running row
// maximum of the single walk is a register (exact, no cross-wave staleness).  Scored entries are staged in LDS and
// appended after the walk if they lie within the margin of the workgroup's own final maximum; the recheck kernel cuts
// against the global one.
#ifndef FTK_RR_DEBUG
#define FTK_RR_DEBUG 0  // timing experiments only (scripts/build_variant.sh): 1 no collection, 2 one epilogue row, 4 one MFMA step
#endif
constexpr int kRrTile = 64;        // cur rows per step
constexpr int kRrRows = 512;       // ref rows per workgroup
constexpr int kRrListCap = 1024;   // tiles of one workgroup's slice that the NearbyMatch tile list can hold (longer slices: no list)
constexpr int kRrStepEntriesMax = 384;  // most a wave can stage in one step: best, second best and a whole-share entry per lane and nt (64 x 2 x 3)
constexpr int kRrWaveStageCap = 768;  // staged entries per wave (64 rows; measured need ~3.3 per row)
constexpr int kRrStageCap = 8 * kRrWaveStageCap;
constexpr float kRrNone = -1.0e30f;   // scores at or below it are "no candidate": cur_info.x is -3e38 (finite, so the
                                      // index bits never turn it into a NaN) for padding / irregular cur rows

// The cur tile goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, which the 128 resident
// fragment registers leave no room for — with register staging the compiler spilled it to scratch).  One instruction
// writes 64 consecutive 16-byte chunks, so rows cannot be padded; instead chunk g of row r sits at position
// g ^ swizzle(r), which spreads the 16 rows a fragment read touches at once over all bank groups.  The two tile buffers
// are distinct static arrays: the compiler then knows a read of one does not wait for the transfer into the other (it
// orders LDS transfers against LDS reads by alias analysis) and completes the transfer at the next barrier.
template <int kKSteps>
__device__ __forceinline__ int rr_swizzle(int row) {
    return ((2 * kKSteps) % 16 == 0) ? (row & 15) : ((row >> 1) & 7);
}

// NearbyMatch only: bounding box {u min, u max, v min, v max} of the candidate pixels of every 64-row cur tile.  A
// candidate with a NaN coordinate passes every window test (fabs(NaN) > x is false, descriptor_matcher.h:108-111), so its
// tile gets the whole plane; a tile of padding rows only gets the empty box.
__global__ void __launch_bounds__(64) cosine_tile_box_kernel(const CosineParams p) {
    const int lane = threadIdx.x, j = (int)blockIdx.x * 64 + lane;
    const float pos_inf = __uint_as_float(0x7F800000u), neg_inf = __uint_as_float(0xFF800000u);
    float u0 = pos_inf, u1 = neg_inf, v0 = pos_inf, v1 = neg_inf;
    bool unordered = false;
    if (j < p.n_cur) {
        const float u = p.cur_uv[2 * j], v = p.cur_uv[2 * j + 1];
        unordered = isnan(u) || isnan(v);
        if (!unordered) {
            u0 = u1 = u;
            v0 = v1 = v;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        u0 = fminf(u0, __shfl_xor(u0, off));
        u1 = fmaxf(u1, __shfl_xor(u1, off));
        v0 = fminf(v0, __shfl_xor(v0, off));
        v1 = fmaxf(v1, __shfl_xor(v1, off));
    }
    if (__ballot(unordered) != 0ull) {
        u0 = v0 = neg_inf;
        u1 = v1 = pos_inf;
    }
    if (lane == 0) {
        p.tile_box[blockIdx.x] = make_float4(u0, u1, v0, v1);
    }
}

// cur row, relative to the first row of the lane's share, of accumulator element e = 16 * mt + r (32x32 MFMA C/D map)
__device__ __forceinline__ uint32_t rr_share_offset(uint32_t e) { return (e >> 4) * 32u + (e & 3u) + 8u * ((e & 15u) >> 2); }

template <int kKSteps, bool kNearby>
__global__ void __launch_bounds__(512) cosine_gemm_rr_kernel(const CosineParams p) {
    constexpr int kDimPad = kKSteps * 16, kChunks = 2 * kKSteps /* 16-byte chunks per row */, kStageVecs = kKSteps / 4;
    typedef __attribute__((address_space(3))) void *lds_ptr;
    __shared__ __attribute__((aligned(16))) _Float16 sXa[kRrTile * kDimPad];
    __shared__ __attribute__((aligned(16))) _Float16 sXb[kRrTile * kDimPad];
    __shared__ __attribute__((aligned(16))) float4 sInfo0[kRrTile];  // {bias, u, v, -} per cur row; four of them: the late
    __shared__ __attribute__((aligned(16))) float4 sInfo1[kRrTile];  // waves (see the walk) still read step s - 1's while
    __shared__ __attribute__((aligned(16))) float4 sInfo2[kRrTile];  // step s + 1's is in flight
    __shared__ __attribute__((aligned(16))) float4 sInfo3[kRrTile];
    extern __shared__ __attribute__((aligned(16))) unsigned char rr_lds[];
    uint32_t *const sRun = reinterpret_cast<uint32_t *>(rr_lds);  // [512] final row maxima of this walk
    uint32_t *const sStage = sRun + kRrRows;                      // [8 waves][kRrWaveStageCap][3]
    int *const sTiles = reinterpret_cast<int *>(sStage + 3 * kRrStageCap);  // [kRrListCap] tiles this workgroup walks (NearbyMatch)
    float *const sBox = reinterpret_cast<float *>(sTiles + kRrListCap);     // [8 waves][4] + [1] list length

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid of row_groups x splits workgroups.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2):
    // renumber so that the workgroups of ONE XCD are consecutive in (split, row group) order and share as few cur slices
    // as possible — otherwise every XCD streams all of cur through its 4 MB L2 (30 % misses measured).
    const int row_groups = p.n_ref_pad / kRrRows, n_splits = p.splits;
    const int n_wg = row_groups * n_splits;
    int wg = (int)blockIdx.x;
    {
        const int per_xcd = n_wg / 8, rem = n_wg % 8, xcd = wg % 8, k = wg / 8;
        wg = xcd * per_xcd + (xcd < rem ? xcd : rem) + k;  // XCD x owns a contiguous run of per_xcd (+1 for the first rem) ids
    }
    const int split = wg / row_groups, row_group = wg - split * row_groups;
    const int i0 = row_group * kRrRows + wave * 64;
    const int tiles_total = p.n_cur_pad / kRrTile;
    // even split of the cur tiles
    const int jt_begin = (int)(((long long)split * tiles_total) / n_splits);
    const int jt_end = (int)(((long long)(split + 1) * tiles_total) / n_splits);
    if (jt_begin >= jt_end) {
        return;
    }
    int n_tiles = jt_end - jt_begin, n_steps = n_tiles;  // no second visit of the first tile here: see the collection below
    bool use_list = false;
    const float neg_inf = __uint_as_float(0xFF800000u);
    uint32_t wcount = 0u;  // entries this wave has staged (wave-uniform)

    int row_i[2];
    bool live[2];
    float pu[2], pv[2], run[2];
    half8 bfrag[2][kKSteps];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int i = i0 + nt * 32 + (lane & 31);
        row_i[nt] = i;
        live[nt] = i < p.n_ref && p.ref_irregular[i] == 0;
        pu[nt] = (kNearby && i < p.n_ref) ? p.pred_uv[2 * i] : 0.0f;
        pv[nt] = (kNearby && i < p.n_ref) ? p.pred_uv[2 * i + 1] : 0.0f;
        run[nt] = neg_inf;
        const _Float16 *src = p.ref_h + (size_t)i * kDimPad + 8 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < kKSteps; ++kk) {
            bfrag[nt][kk] = *reinterpret_cast<const half8 *>(src + kk * 16);
        }
    }

    int list_lo = 0;
    if (kNearby && p.tile_box != nullptr && tiles_total <= kRrListCap) {
        // NearbyMatch: walk only the tiles whose candidates can lie in the window of SOME row of this workgroup.  Every
        // workgroup of a row group builds the same list over ALL cur tiles and takes its n_splits-th share of it, so the
        // listed tiles — not the raw tiles — are what is balanced over the workgroups.  Tile and
        // workgroup bounding boxes more than window + 1 px apart on an axis cannot hold a pair that passes
        // fabs(du) <= max_col && fabs(dv) <= max_row (the extra pixel covers the rounding of the fp32 difference; a NaN on
        // either side makes its box the whole plane).  Exact for any input; it pays when features arrive in spatial
        // order (detectors scan the image), where a workgroup's 512 rows see a band of the image.
        const float pos_inf = __uint_as_float(0x7F800000u);
        float u0 = pos_inf, u1 = neg_inf, v0 = pos_inf, v1 = neg_inf;
        bool unordered = false;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (live[nt]) {
                if (isnan(pu[nt]) || isnan(pv[nt])) {
                    unordered = true;
                } else {
                    u0 = fminf(u0, pu[nt]);
                    u1 = fmaxf(u1, pu[nt]);
                    v0 = fminf(v0, pv[nt]);
                    v1 = fmaxf(v1, pv[nt]);
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            u0 = fminf(u0, __shfl_xor(u0, off));
            u1 = fmaxf(u1, __shfl_xor(u1, off));
            v0 = fminf(v0, __shfl_xor(v0, off));
            v1 = fmaxf(v1, __shfl_xor(v1, off));
        }
        if (__ballot(unordered) != 0ull) {
            u0 = v0 = neg_inf;
            u1 = v1 = pos_inf;
        }
        if (lane == 0) {
            sBox[4 * wave] = u0;
            sBox[4 * wave + 1] = u1;
            sBox[4 * wave + 2] = v0;
            sBox[4 * wave + 3] = v1;
        }
        __syncthreads();
        if (wave == 0) {
            float wu0 = sBox[0], wu1 = sBox[1], wv0 = sBox[2], wv1 = sBox[3];
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                wu0 = fminf(wu0, sBox[4 * w]);
                wu1 = fmaxf(wu1, sBox[4 * w + 1]);
                wv0 = fminf(wv0, sBox[4 * w + 2]);
                wv1 = fmaxf(wv1, sBox[4 * w + 3]);
            }
            const float reach_u = p.max_col + 1.0f, reach_v = p.max_row + 1.0f;
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            int listed = 0;
            for (int t0 = 0; t0 < tiles_total; t0 += 64) {
                const int t = t0 + lane;
                bool hit = false;
                if (t < tiles_total) {
                    const float4 bx = p.tile_box[t];
                    hit = !(bx.x - wu1 > reach_u || wu0 - bx.y > reach_u || bx.z - wv1 > reach_v || wv0 - bx.w > reach_v);
                }
                const unsigned long long mask = __ballot(hit);
                if (hit) {
                    sTiles[listed + (int)__popcll(mask & ((1ull << below) - 1ull))] = t;
                }
                listed += (int)__popcll(mask);
            }
            if (lane == 0) {
                sBox[32] = __int_as_float(listed);
            }
        }
        __syncthreads();
        const int listed = __float_as_int(sBox[32]);
        list_lo = (int)(((long long)split * listed) / n_splits);
        n_tiles = (int)(((long long)(split + 1) * listed) / n_splits) - list_lo;
        if (n_tiles == 0) {
            return;  // nothing (left) for this workgroup: no candidate lies in any window of these rows, or fewer listed tiles than workgroups
        }
        n_steps = n_tiles;
        use_list = true;
    }

    // transfer q of this wave fills LDS chunks [(wave * kStageVecs + q) * 64, +64); this lane's chunk L = row * kChunks + pos
    // holds global chunk pos ^ swizzle(row) of that row
    int soff0 = 0, soff1 = 0, soff2 = 0, soff3 = 0;  // half offsets into a cur tile; kStageVecs of them in use
    {
        auto source_of = [&](int q) {
            const int L = (wave * kStageVecs + q) * 64 + lane;
            const int r = L / kChunks, pos = L % kChunks;
            return r * kDimPad + (pos ^ rr_swizzle<kKSteps>(r)) * 8;
        };
        soff0 = source_of(0);
        if (kStageVecs > 1) soff1 = source_of(1);
        if (kStageVecs > 2) soff2 = source_of(2);
        if (kStageVecs > 3) soff3 = source_of(3);
    }
    auto rr_tile_of = [&](int idx) { return (kNearby && use_list) ? sTiles[list_lo + idx] : jt_begin + idx; };
#define FTK_RR_TILE(step_) rr_tile_of(step_)
#define FTK_RR_FETCH(step_, SX, SINFO)                                                                                        \
    do {                                                                                                                      \
        const int jt_ = FTK_RR_TILE(step_);                                                                                   \
        const _Float16 *g_ = p.cur_h + (size_t)jt_ * kRrTile * kDimPad;                                                       \
        _Float16 *d_ = SX + (wave * kStageVecs) * 64 * 8;                                                                     \
        __builtin_amdgcn_global_load_lds(g_ + soff0, (lds_ptr)(d_), 16, 0, 0);                                                \
        if (kStageVecs > 1) __builtin_amdgcn_global_load_lds(g_ + soff1, (lds_ptr)(d_ + 512), 16, 0, 0);                      \
        if (kStageVecs > 2) __builtin_amdgcn_global_load_lds(g_ + soff2, (lds_ptr)(d_ + 1024), 16, 0, 0);                     \
        if (kStageVecs > 3) __builtin_amdgcn_global_load_lds(g_ + soff3, (lds_ptr)(d_ + 1536), 16, 0, 0);                     \
        if (wave == 0) {                                                                                                      \
            __builtin_amdgcn_global_load_lds(p.cur_info + (size_t)jt_ * kRrTile + lane, (lds_ptr)(SINFO), 16, 0, 0);          \
        }                                                                                                                     \
    } while (0)

    float16v acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[mt][nt][r] = 0.0f;
            }
        }
    }
    // fragment address of this lane: row (lane & 31) (+ 32 for the second tile row block), chunk 2 * kk + (lane >> 5)
    // -> byte offset row * kDimPad * 2 + ((32 * kk) ^ frag_flip)
    const int frag_row_bytes = (lane & 31) * kDimPad * 2;
    const int frag_flip = ((lane >> 5) ^ rr_swizzle<kKSteps>(lane & 31)) << 4;

    // Appends the wave's staged entries to the per-row global lists and empties the stage.  An entry survives if it lies
    // within the margin of the row's maximum SO FAR (never above the final one, so nothing that can decide is dropped; the
    // recheck cuts against the final global maximum).  Called at the end of the walk, and inside it whenever the stage
    // could not take another step's worth of entries — long walks (few workgroups per row group, many candidates) would
    // otherwise overflow it and send their rows to the exact scan.  The wave's 64 rows and its stage are its own: no
    // workgroup barrier.
    auto flush_stage = [&]() {
        if (lane < 32) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                sRun[wave * 64 + nt * 32 + lane] = run[nt] > neg_inf ? order_key(run[nt]) : 0u;
            }
        }
        const uint32_t staged = min(wcount, (uint32_t)kRrWaveStageCap);
        const uint32_t *const stage = sStage + (size_t)wave * kRrWaveStageCap * 3;
        const int g0 = row_group * kRrRows;
        for (uint32_t e = (uint32_t)lane; e < staged; e += 64u) {
            const uint32_t row_word = stage[3 * e], local = row_word & 0x7FFFFFFFu;
            const uint32_t cur_word = stage[3 * e + 1];
            const float score = __uint_as_float(stage[3 * e + 2]);
            if (score >= order_value(sRun[local]) - 2.0f * kMargin) {  // sRun[local] != 0: the row staged something
                const int i = g0 + (int)local;
                if (row_word & 0x80000000u) {
                    const uint32_t first = atomicAdd(&p.cand_count[i], 32u);
                    if (first + 32u <= (uint32_t)kCosineCandCap) {
                        const uint32_t j_best = cur_word + rr_share_offset(__float_as_uint(score) & 31u);
                        for (uint32_t k = 0; k < 32u; ++k) {
                            const uint32_t j = cur_word + rr_share_offset(k);
                            // padding rows of the last tile are not candidates: their slots repeat the lane's best
                            p.cand[(size_t)i * kCosineCandCap + first + k] = (int32_t)(j < (uint32_t)p.n_cur ? j : j_best);
                            p.cand_score[(size_t)i * kCosineCandCap + first + k] = score;
                        }
                    }
                } else {
                    const uint32_t slot = atomicAdd(&p.cand_count[i], 1u);
                    if (slot < (uint32_t)kCosineCandCap) {
                        p.cand[(size_t)i * kCosineCandCap + slot] = (int32_t)cur_word;
                        p.cand_score[(size_t)i * kCosineCandCap + slot] = score;
                    }
                }
            }
        }
        wcount = 0u;
    };

    auto rr_mfma = [&](const unsigned char *ax0) {
        // 64 cur rows x 64 ref rows x whole K: fragments of K step kk + 1 are read while the MFMAs of step kk run
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[mt][nt][r] = 0.0f;
                }
            }
        }
        half8 fa[2][2];
        int flip = frag_flip;
        asm volatile("" : "+v"(flip));  // per step: otherwise the 2 x kKSteps fragment addresses are all hoisted out of the walk
        fa[0][0] = *reinterpret_cast<const half8 *>(ax0 + flip);
        fa[0][1] = *reinterpret_cast<const half8 *>(ax0 + flip + 32 * kDimPad * 2);
#pragma unroll
        for (int kk = 0; kk < kKSteps; ++kk) {
            const int cur_set = kk & 1, nxt_set = cur_set ^ 1;
            if (kk + 1 < kKSteps) {
                const int off = (32 * (kk + 1)) ^ flip;
                fa[nxt_set][0] = *reinterpret_cast<const half8 *>(ax0 + off);
                fa[nxt_set][1] = *reinterpret_cast<const half8 *>(ax0 + off + 32 * kDimPad * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
#if FTK_RR_DEBUG & 4
                    if (kk == 0)
#endif
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur_set][mt], bfrag[nt][kk], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    };
    // epilogue: C/D map of the 32x32 MFMA — col (ref) = lane & 31, row (cur) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto rr_epilogue = [&](int s, const float4 *info_tile) {
        const int j0 = FTK_RR_TILE(s) * kRrTile;
        // Pass 1, branch-free: score = accumulator + bias (window applied), with the element's index e = 16 * mt + r written
        // into the five low mantissa bits (a perturbation below 2e-6, see the margin budget in the header), and the two
        // largest of the lane's 32 scores: m1 (which then names its own element) and m2.
        float m1[2] = {neg_inf, neg_inf}, m2[2] = {neg_inf, neg_inf};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if FTK_RR_DEBUG & 2
                if (r != 5) continue;
#endif
                const int jl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float4 info = info_tile[jl];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v = acc[mt][nt][r] + info.x;
                    v = __uint_as_float((__float_as_uint(v) & ~31u) | (uint32_t)(mt * 16 + r));
                    if (kNearby) {
                        const bool out = (int)(fabsf(pu[nt] - info.y) > p.max_col) | (int)(fabsf(pv[nt] - info.z) > p.max_row);
                        v = out ? neg_inf : v;
                    }
                    acc[mt][nt][r] = v;  // in place, for the rare look at every element below
                    m2[nt] = __builtin_amdgcn_fmed3f(m1[nt], m2[nt], v);
                    // the instruction itself: fmaxf() is preceded by a canonicalising v_max v, v, v per operand (the compiler cannot
                    // know that the bit-edited score is not a signalling NaN) — 1.5 extra VALU instructions per element
                    asm("v_max_f32 %0, %1, %2" : "=v"(m1[nt]) : "v"(m1[nt]), "v"(v));
                }
                if ((r & 3) == 3) {
                    // four rows of per-candidate data in flight, not all thirty-two: the running values are pinned here, or the
                    // compiler sinks the whole nt = 1 chain below the nt = 0 one and keeps every loaded value alive for it
                    asm volatile("" : "+v"(m1[0]), "+v"(m1[1]), "+v"(m2[0]), "+v"(m2[1]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // Collection.  A lane whose best score reaches the bound stages that one element without looking at the others (the
        // usual case); one whose second best does too stages both; the wave appends through a ballot, no atomic.  Only when
        // a lane's second best is in do its 32 scores get counted, and three or more above the bound name the whole share.
        const uint32_t lanes_below = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        uint32_t *const wave_stage = sStage + (size_t)wave * kRrWaveStageCap * 3;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const bool has = live[nt] && m1[nt] > kRrNone;  // padding / irregular rows and empty windows collect nothing
            float mine = has ? m1[nt] : neg_inf;
            {   // lanes l and l ^ 32 hold the two halves of the row's tile: one v_permlane32_swap instead of an LDS round trip
                const auto halves = __builtin_amdgcn_permlane32_swap(__float_as_uint(mine), __float_as_uint(mine), false, false);
                mine = fmaxf(__uint_as_float(halves[0]), __uint_as_float(halves[1]));
            }
#if !(FTK_RR_DEBUG & 1)
            {
                // the row's maximum over this tile is exact here (one wave owns the row; both halves were just joined), so the
                // first tile collects against its own maximum and needs no second visit (the LDS-stationary kernel above,
                // whose waves share a row through a stale LDS word, does need one)
                const float bound = fmaxf(run[nt], mine) - 2.0f * kMargin;
                const bool hit1 = has && m1[nt] >= bound;
                const bool hit2 = has && m2[nt] >= bound && m2[nt] > kRrNone;
                const uint32_t slot_row = (uint32_t)(wave * 64 + nt * 32 + (lane & 31));
                // wave-uniform call; lanes with `on` append (row word, cur index word, score).  Row word bit 31: the entry names the
                // lane's whole share of the tile (cur index word = its first row); the flush expands it if the score survives.
                auto append = [&](bool on, uint32_t row_word, uint32_t cur_word, float v) {
                    const unsigned long long mask = __ballot(on);
                    if (mask != 0ull) {
                        const uint32_t at = wcount + (uint32_t)__popcll(mask & ((1ull << lanes_below) - 1ull));
                        if (on) {
                            if (at < (uint32_t)kRrWaveStageCap) {
                                wave_stage[3 * at] = row_word;
                                wave_stage[3 * at + 1] = cur_word;
                                wave_stage[3 * at + 2] = __float_as_uint(v);
                            } else {
                                p.cand_count[row_i[nt]] = (uint32_t)kCosineCandCap + 1u;  // stage full (never seen): the row takes the exact scan
                            }
                        }
                        wcount += (uint32_t)__popcll(mask);
                    }
                };
                const uint32_t share_first = (uint32_t)(j0 + 4 * (lane >> 5));
                const uint32_t e1 = __float_as_uint(m1[nt]) & 31u, e2 = __float_as_uint(m2[nt]) & 31u;
                append(hit1, slot_row, share_first + rr_share_offset(e1), m1[nt]);
                if (__ballot(hit2) != 0ull) {  // wave-uniform, so the wave's count stays uniform
                    append(hit2, slot_row, share_first + rr_share_offset(e2), m2[nt]);  // med3 returns one of its inputs: m2 names its element too
                    // a third score above the bound?  count, branch-free
                    const float bound2 = hit2 ? bound : __uint_as_float(0x7F800000u);
                    int above = 0;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            above += (acc[mt][nt][r] >= bound2) ? 1 : 0;
                        }
                    }
                    // rare: name all 32 cur rows of this lane's share, under the lane's best score; the recheck decides
                    append(above > 2, slot_row | 0x80000000u, share_first, m1[nt]);
                }
            }
#endif
            run[nt] = fmaxf(run[nt], mine);
        }
        if (wcount > (uint32_t)(kRrWaveStageCap - kRrStepEntriesMax)) {  // wave-uniform, rare
            flush_stage();
        }
    };

    // The walk.  Waves 0-3 ("early") run MFMA(s) then epilogue(s) inside step s; waves 4-7 ("late") run epilogue(s - 1)
    // then MFMA(s), keeping their accumulators across the barrier.  Wave w and wave w + 4 share a SIMD: in lockstep both
    // would feed the matrix pipe together and then both run their epilogues on the vector ALU; staggered by half a step
    // one pipe works while the other does.  One barrier per step, before anybody reads tile s — it also says that everyone
    // is done reading tile s - 1, whose buffer the transfer of tile s + 1 overwrites.  The late waves still read the
    // per-candidate data of step s - 1 then, hence four rotating buffers for it and the walk unrolled by four (the buffers
    // are distinct static arrays so that the compiler waits for a transfer only where its target is read).
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
#ifdef FTK_RR_STAMPS  // timing build only: cycles per phase of one early and one late wave of one workgroup, printed at the end
    unsigned long long st_wait = 0, st_fetch = 0, st_mfma = 0, st_epi = 0, st_t = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_t;
#define FTK_RR_LAP(acc_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - st_t; st_t = n_; } while (0)
#else
#define FTK_RR_LAP(acc_) do { } while (0)
#endif
#define FTK_RR_STEP(s_, SX, SINFO, SXN, SINFON, SINFOP)                                                                     \
    {                                                                                                                         \
        if ((s_) >= n_steps) {                                                                                                \
            break; /* leaving (not skipping) keeps the accumulators out of a merge: a skipped MFMA block costs 64 copies */    \
        }                                                                                                                     \
        __syncthreads();                                                                                                      \
        FTK_RR_LAP(st_wait);                                                                                                  \
        if ((s_) + 1 < n_steps) {                                                                                             \
            FTK_RR_FETCH((s_) + 1, SXN, SINFON);                                                                              \
        }                                                                                                                     \
        FTK_RR_LAP(st_fetch);                                                                                                 \
        if (late && (s_) >= 1) {                                                                                              \
            rr_epilogue((s_) - 1, SINFOP);                                                                                    \
        }                                                                                                                     \
        FTK_RR_LAP(st_epi);                                                                                                   \
        rr_mfma(reinterpret_cast<const unsigned char *>(SX) + frag_row_bytes);                                               \
        FTK_RR_LAP(st_mfma);                                                                                                  \
        if (!late) {                                                                                                          \
            rr_epilogue(s_, SINFO);                                                                                           \
        }                                                                                                                     \
        FTK_RR_LAP(st_epi);                                                                                                   \
    }
    FTK_RR_FETCH(0, sXa, sInfo0);
    for (int s = 0;; s += 4) {
        FTK_RR_STEP(s, sXa, sInfo0, sXb, sInfo1, sInfo3)
        FTK_RR_STEP(s + 1, sXb, sInfo1, sXa, sInfo2, sInfo0)
        FTK_RR_STEP(s + 2, sXa, sInfo2, sXb, sInfo3, sInfo1)
        FTK_RR_STEP(s + 3, sXb, sInfo3, sXa, sInfo0, sInfo2)
    }
    if (late) {  // the last step's epilogue of the late waves (its per-candidate data landed before that step's barrier)
        const int last = n_steps - 1, slot = last & 3;
        rr_epilogue(last, slot == 0 ? sInfo0 : slot == 1 ? sInfo1 : slot == 2 ? sInfo2 : sInfo3);
    }
#undef FTK_RR_STEP
#undef FTK_RR_FETCH
#undef FTK_RR_TILE
    if (lane < 32) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (run[nt] > neg_inf) {  // implies live
                atomicMax(&p.row_max[row_i[nt]], order_key(run[nt]));
            }
        }
    }
    flush_stage();
#ifdef FTK_RR_STAMPS
    if (blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 4)) {
        const unsigned long long total = __builtin_amdgcn_s_memtime() - st_begin;
        printf("rr stamps wave %d steps %d: wait %llu fetch %llu mfma %llu epilogue %llu (in loop) total %llu cycles (s_memtime ticks)\n", wave, n_steps,
               st_wait, st_fetch, st_mfma, st_epi, total);
    }
#endif
}

// ---- 4. exact decision -------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cosine_recheck_kernel(const CosineParams p) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = tid >> 3, c = tid & 7, base = (threadIdx.x & 63) & ~7;
    if (row >= p.n_ref) {
        return;  // octets leave whole
    }
    const bool nearby = p.pred_uv != nullptr;
    const float *x = p.ref + (size_t)row * p.dim;
    const float na = p.ref_norm[row];
    const uint32_t n_irr = *p.irregular_count;
    const uint32_t cnt = p.cand_count[row];
    const bool scan_all = p.ref_irregular[row] != 0 || cnt > (uint32_t)kCosineCandCap || n_irr > (uint32_t)kCosineIrregularCap;
    // single-walk lists were cut against the running maximum: keep what the final maximum admits
    const uint32_t max_key = p.cand_score ? p.row_max[row] : 0u;
    const float admit = max_key ? order_value(max_key) - 2.0f * kMargin : __uint_as_float(0xFF800000u);
    const float pu = nearby ? p.pred_uv[2 * row] : 0.0f, pv = nearby ? p.pred_uv[2 * row + 1] : 0.0f;
    float best_d = __uint_as_float(0x7F800000u);
    int best_j = -1;
    int zero_j = INT_MAX;  // NearbyMatch: lowest in-window j whose distance is exactly 0 — where the reference's scan stops (:119)
    auto consider = [&](int j) {  // whole octet, same j
        const float dot = eigen_dot_octet(x, p.cur + (size_t)j * p.dim, p.dim, c, base);
        const float d = 0.5f - dot / na / p.cur_norm[j] * 0.5f;
        if (d < best_d || (d == best_d && j < best_j)) {
            best_d = d;
            best_j = j;
        }
        if (nearby && d == 0.0f && j < zero_j) {
            zero_j = j;
        }
    };
    auto outside = [&](int j) { return nearby && (fabsf(pu - p.cur_uv[2 * j]) > p.max_col || fabsf(pv - p.cur_uv[2 * j + 1]) > p.max_row); };
    // minimum over the row's candidates with j <= j_limit (octet-uniform)
    auto scan = [&](int j_limit) {
        best_d = __uint_as_float(0x7F800000u);
        best_j = -1;
        if (scan_all) {
            const int j_end = j_limit < p.n_cur - 1 ? j_limit + 1 : p.n_cur;
            for (int j = 0; j < j_end; ++j) {
                if (!outside(j)) {
                    consider(j);
                }
            }
            return;
        }
        // eight list entries per round, one per lane: the filters (score, window) run in parallel and only the survivors —
        // one or two per row — cost an exact distance; a serial walk pays the load latency once per entry instead
        const int n_cand = (int)cnt, n_all = (int)(cnt + n_irr);
        for (int t0 = 0; t0 < n_all; t0 += 8) {
            const int t = t0 + c;
            int j = -1;
            if (t < n_cand) {
                j = p.cand[(size_t)row * kCosineCandCap + t];
                if (p.cand_score && p.cand_score[(size_t)row * kCosineCandCap + t] < admit) {
                    j = -1;
                }
            } else if (t < n_all) {
                j = p.irregular_list[t - n_cand];
            }
            if (j >= 0 && (j > j_limit || outside(j))) {
                j = -1;
            }
            unsigned live_mask = (unsigned)((__ballot(j >= 0) >> base) & 0xFFull);
            while (live_mask) {
                const int k = __ffs(live_mask) - 1;
                live_mask &= live_mask - 1u;
                consider(__shfl(j, base + k));
            }
        }
    };
    scan(INT_MAX);
    // NearbyMatch stops scanning a row at the first in-window candidate at distance exactly 0 (descriptor_matcher.h:119), so
    // candidates behind it never compete.  Nothing beats a zero except a NEGATIVE distance (a cosine rounded above 1); only
    // when the global minimum is such a candidate and lies behind the stop is the minimum taken again over j <= stop.  Every
    // zero-distance candidate is in the list (its score is the row maximum to within the margin), so the stop is exact.
    if (nearby && best_d < 0.0f && best_j > zero_j) {
        scan(zero_j);
    }
    // strict '<' against a running minimum that starts at the threshold (descriptor_matcher.h:68-75, :114-117)
    if (c == 0 && best_j >= 0 && best_d < p.max_distance) {
        p.index_pairs[row] = best_j;
    }
}

// ---- small calls: the whole match in ONE launch, exact arithmetic only --------------------------
// The sizes the reference's own programs match (<= 300 features a side: test_descriptor_matcher_superpoint.cpp:52-58) are bound by
// the pipeline's launches (clear + prep + contraction + recheck: 42 us for 300 x 300 x 256 NearbyMatch, of which the pairs inside
// the windows are a few thousand exact distances).  Here a wave owns a reference row, kept in registers — lane c of every octet
// holds x[8 k + c], the elements its lane of Eigen's two packet accumulators multiplies — and its eight octets walk the
// candidates j = octet, octet + 8, ...: window test, then x . y and y . y by the octet in Eigen's order (eigen_dot_octet's
// operations on registers), the distance as the scalar code writes it, a running minimum per octet (strict '<': lowest j on
// ties), merged over the octets at the end.  No workspace, no fp16 copies, no candidate lists; dims 64 / 128 / 256.
template <int kSteps>
__device__ __forceinline__ float small_octet_sum(float acc, int q, int base) {
    // (acc of packet accumulator 0 + accumulator 1), then the SSE2 predux (p0 + p2) + (p1 + p3): Core/Redux.h
    const float p = acc + __shfl(acc, base + 4 + q);  // meaningful in the octet's lanes 0 - 3
    const float p0 = __shfl(p, base + 0), p1 = __shfl(p, base + 1), p2 = __shfl(p, base + 2), p3 = __shfl(p, base + 3);
    return (p0 + p2) + (p1 + p3);
}

template <int kSteps, bool kNearby>
__global__ void __launch_bounds__(256) cosine_match_small_kernel(const CosineParams p) {
    const int lane = (int)threadIdx.x & 63, c = lane & 7, q = c & 3, octet = lane >> 3, base = lane & ~7;
    const int row = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + ((int)threadIdx.x >> 6));
    if (row >= p.n_ref) {
        return;
    }
    constexpr int kDim = 8 * kSteps;
    float x[kSteps];
    {
        const float *xr = p.ref + (size_t)row * kDim + c;
#pragma unroll
        for (int k = 0; k < kSteps; ++k) {
            x[k] = xr[8 * k];
        }
    }
    float acc = x[0] * x[0];
#pragma unroll
    for (int k = 1; k < kSteps; ++k) {
        acc = acc + x[k] * x[k];
    }
    const float na = sqrtf(small_octet_sum<kSteps>(acc, q, base));  // ref.norm()
    float pu = 0.0f, pv = 0.0f;
    if (kNearby) {
        pu = p.pred_uv[2 * (size_t)row];
        pv = p.pred_uv[2 * (size_t)row + 1];
    }
    float best_d = __uint_as_float(0x7F800000u);
    int best_j = -1;
    int zero_j = INT_MAX;  // NearbyMatch: lowest in-window j whose distance is exactly 0 — where the reference's scan stops (:119)
    // minimum over the candidates j <= j_limit of this lane's octet
    auto scan = [&](int j_limit) {
        best_d = __uint_as_float(0x7F800000u);
        best_j = -1;
        const int j_last = j_limit < p.n_cur - 1 ? j_limit : p.n_cur - 1;
        for (int j0 = 0; j0 <= j_last; j0 += 64) {
            // 64 candidates at a time through the window test (one per lane); the survivors — a few per row under a real window —
            // are dealt to the octets eight at a time in ascending j, so that an octet meets its candidates in ascending order
            const int jt = j0 + lane;
            bool inside = jt <= j_last;
            if (kNearby && inside) {
                const float2 cu = reinterpret_cast<const float2 *>(p.cur_uv)[jt];
                inside = !(fabsf(pu - cu.x) > p.max_col || fabsf(pv - cu.y) > p.max_row);  // descriptor_matcher.h:108-111
            }
            unsigned long long mask = __ballot(inside);
            while (mask != 0ull) {  // wave-uniform
                int mine = -1;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const int bit = mask != 0ull ? (int)__builtin_ctzll(mask) : -1;
                    mine = octet == o ? bit : mine;
                    mask &= mask - 1ull;
                }
                const bool live = mine >= 0;
                const int j = j0 + (live ? mine : 0);
                const float *yr = p.cur + (size_t)j * kDim + c;
                float y[kSteps];
#pragma unroll
                for (int k = 0; k < kSteps; ++k) {
                    y[k] = yr[8 * k];  // every load of the row in flight before the first product
                }
                float dot = x[0] * y[0], nn = y[0] * y[0];
#pragma unroll
                for (int k = 1; k < kSteps; ++k) {
                    dot = dot + x[k] * y[k];
                    nn = nn + y[k] * y[k];
                }
                const float nb = sqrtf(small_octet_sum<kSteps>(nn, q, base));  // cur.norm()
                const float d = 0.5f - small_octet_sum<kSteps>(dot, q, base) / na / nb * 0.5f;
                if (live && d < best_d) {  // j ascends within an octet: a tie keeps the earlier candidate
                    best_d = d;
                    best_j = j;
                }
                if (kNearby && live && d == 0.0f && j < zero_j) {
                    zero_j = j;
                }
            }
        }
    };
    // the minimum over the wave's eight octets: smallest distance, lowest j among equals (NaN distances never enter)
    auto merge = [&]() {
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {
            const float od = __shfl_xor(best_d, off);
            const int oj = __shfl_xor(best_j, off);
            if (oj >= 0 && (best_j < 0 || od < best_d || (od == best_d && oj < best_j))) {
                best_d = od;
                best_j = oj;
            }
        }
    };
    scan(INT_MAX);
    merge();
    if (kNearby) {
        // the reference stops a row's scan at the first in-window candidate at distance exactly 0; only a NEGATIVE distance (a cosine
        // rounded above 1) behind that stop could differ: then the minimum is taken again over j <= stop (cosine_recheck_kernel)
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {
            zero_j = min(zero_j, __shfl_xor(zero_j, off));
        }
        if (best_j >= 0 && best_d < 0.0f && best_j > zero_j) {  // wave-uniform after the merges
            scan(zero_j);
            merge();
        }
    }
    // strict '<' against a running minimum that starts at the threshold (descriptor_matcher.h:68-75, :114-117)
    if (lane == 0 && best_j >= 0 && best_d < p.max_distance) {
        p.index_pairs[row] = best_j;
    }
}

template <int kSteps>
hipError_t cosine_launch_small(const CosineParams &p, hipStream_t stream) {
    const dim3 grid((unsigned)((p.n_ref + 3) / 4));
    if (p.pred_uv) {
        hipLaunchKernelGGL((cosine_match_small_kernel<kSteps, true>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((cosine_match_small_kernel<kSteps, false>), grid, dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace

bool cosine_small_form(int n_ref, int n_cur, int dim, bool nearby, bool small_off, bool small_any) {
    if (small_off) {  // FTK_COSINE_SMALL=0 (experiment switch of the context)
        return false;
    }
    // small_any: FTK_COSINE_SMALL_ANY=1, no size limit (scripts/cosine_small_ab.py)
    const bool fits = small_any || (n_ref <= kCosineSmallRefMax && n_cur <= (nearby ? kCosineSmallCurNearby : kCosineSmallCurForce));
    return (dim == 64 || dim == 128 || dim == 256) && fits;
}

size_t cosine_rs_lds_bytes(int dim_pad) {
    return sizeof(_Float16) * ((size_t)kTile * (dim_pad + 8) + (size_t)2 * kCurTile * kPitch) + sizeof(float4) * 2 * kCurTile +
           sizeof(uint32_t) * (kTile + 4 + 3 * kStageCap);
}

size_t cosine_rr_lds_bytes(int dim_pad) {
    (void)dim_pad;  // the tile buffers are static arrays of the kernel instance
    return sizeof(uint32_t) * (kRrRows + 3 * kRrStageCap + kRrListCap + 40);
}

hipError_t cosine_match_launch(const CosineParams &p, hipStream_t stream) {
    if (p.n_ref <= 0 || p.n_cur <= 0) {
        return hipSuccess;
    }
    if (cosine_small_form(p.n_ref, p.n_cur, p.dim, p.pred_uv != nullptr, p.small_off != 0, p.small_any != 0)) {
        switch (p.dim) {
            case 64: return cosine_launch_small<8>(p, stream);
            case 128: return cosine_launch_small<16>(p, stream);
            default: return cosine_launch_small<32>(p, stream);
        }
    }
    hipError_t e = hipMemsetAsync(p.clear_begin, 0, p.clear_bytes, stream);  // key 0 = "no candidate yet", counts 0
    if (e != hipSuccess) {
        return e;
    }
    // both operands in one launch: blockIdx.y selects ref / cur
    const bool packets = (p.dim % 8) == 0 && ((reinterpret_cast<uintptr_t>(p.ref) | reinterpret_cast<uintptr_t>(p.cur)) & 15u) == 0;
    {
        const int rows = p.n_ref_pad > p.n_cur_pad ? p.n_ref_pad : p.n_cur_pad;
        if (packets) {
            hipLaunchKernelGGL(cosine_prep_pair_kernel, dim3((unsigned)((rows * 2 + 255) / 256), 2u), dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL(cosine_prep_kernel, dim3((unsigned)((rows * 8 + 255) / 256), 2u), dim3(256), 0, stream, p);
        }
    }
    const int row_tiles = p.n_ref_pad / kTile;
    if (p.ref_stationary == 2) {
        if (p.pred_uv && p.tile_box && !packets) {  // the packet-wide prep kernel has written the boxes itself
            hipLaunchKernelGGL(cosine_tile_box_kernel, dim3((unsigned)(p.n_cur_pad / kRrTile)), dim3(64), 0, stream, p);
        }
        const dim3 grid((unsigned)((p.n_ref_pad / kRrRows) * p.splits));
        const size_t lds = cosine_rr_lds_bytes(p.dim_pad);
#define FTK_RR_LAUNCH(KSTEPS, NEARBY)                                                                                             \
    do {                                                                                                                            \
        auto kern = cosine_gemm_rr_kernel<KSTEPS, NEARBY>;                                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        if (e != hipSuccess) {                                                                                                      \
            return e;                                                                                                               \
        }                                                                                                                           \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);                                                                  \
    } while (0)
#define FTK_RR_DISPATCH(KSTEPS)            \
    do {                                   \
        if (p.pred_uv) {                   \
            FTK_RR_LAUNCH(KSTEPS, true);   \
        } else {                           \
            FTK_RR_LAUNCH(KSTEPS, false);  \
        }                                  \
    } while (0)
        switch (p.dim_pad / 16) {
            case 4: FTK_RR_DISPATCH(4); break;
            case 8: FTK_RR_DISPATCH(8); break;
            case 12: FTK_RR_DISPATCH(12); break;
            case 16: FTK_RR_DISPATCH(16); break;
            default: return hipErrorInvalidValue;
        }
#undef FTK_RR_DISPATCH
#undef FTK_RR_LAUNCH
    } else if (p.ref_stationary) {
        const int tiles_total = p.n_cur_pad / kCurTile;
        const int splits = (tiles_total + p.tiles_per_split - 1) / p.tiles_per_split;
        const dim3 grid((unsigned)row_tiles, (unsigned)splits);
        const size_t lds = cosine_rs_lds_bytes(p.dim_pad);
#define FTK_RS_LAUNCH(MODE, NEARBY)                                                                                               \
    do {                                                                                                                            \
        auto kern = cosine_gemm_rs_kernel<MODE, NEARBY>;                                                                            \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        if (e != hipSuccess) {                                                                                                      \
            return e;                                                                                                               \
        }                                                                                                                           \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);                                                                  \
    } while (0)
        if (p.cand_score) {  // single walk
            if (p.pred_uv) {
                FTK_RS_LAUNCH(2, true);
            } else {
                FTK_RS_LAUNCH(2, false);
            }
        } else if (p.pred_uv) {
            FTK_RS_LAUNCH(0, true);
            FTK_RS_LAUNCH(1, true);
        } else {
            FTK_RS_LAUNCH(0, false);
            FTK_RS_LAUNCH(1, false);
        }
#undef FTK_RS_LAUNCH
    } else {
        const int tiles_total = p.n_cur_pad / kTile;
        const int splits = (tiles_total + p.tiles_per_split - 1) / p.tiles_per_split;
        const dim3 grid((unsigned)row_tiles, (unsigned)splits);
        if (p.pred_uv) {
            hipLaunchKernelGGL((cosine_gemm_kernel<false, true>), grid, dim3(256), 0, stream, p);
            hipLaunchKernelGGL((cosine_gemm_kernel<true, true>), grid, dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL((cosine_gemm_kernel<false, false>), grid, dim3(256), 0, stream, p);
            hipLaunchKernelGGL((cosine_gemm_kernel<true, false>), grid, dim3(256), 0, stream, p);
        }
    }
    hipLaunchKernelGGL(cosine_recheck_kernel, dim3((unsigned)((p.n_ref * 8 + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// First-use cost out of the callers' timed regions (ftk_warmup): launching this empty kernel makes the runtime load this
// translation unit's code object onto the device, which otherwise happens inside the first real call.
__global__ void cosine_warm_kernel() {}
hipError_t cosine_warm(hipStream_t stream) {
    hipLaunchKernelGGL(cosine_warm_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace ftk
