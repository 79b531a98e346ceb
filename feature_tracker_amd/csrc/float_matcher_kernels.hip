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
        for (int index = 8; index < aligned_end2; index += 8) {
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

// ---- 1. norms + unit-length fp16 copies ------------------------------------------------------
// One octet per (padded) row of one operand.  which == 0: ref, 1: cur.
__global__ void __launch_bounds__(256) cosine_prep_kernel(const CosineParams p, int which) {
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
    for (int k = c; k < p.dim_pad; k += 8) {
        const float v = (regular && k < p.dim) ? src[k] * inv : 0.0f;
        dst[k] = (_Float16)v;
    }
    if (c == 0) {
        if (which) {
            p.cur_bias[row] = regular ? 0.0f : __uint_as_float(0xFF800000u);  // -inf keeps the column out of every maximum
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
    const int n_list = scan_all ? p.n_cur : (int)(cnt + n_irr);
    const float pu = nearby ? p.pred_uv[2 * row] : 0.0f, pv = nearby ? p.pred_uv[2 * row + 1] : 0.0f;
    float best_d = __uint_as_float(0x7F800000u);
    int best_j = -1;
    for (int t = 0; t < n_list; ++t) {
        int j;
        if (scan_all) {
            j = t;
        } else if (t < (int)cnt) {
            j = p.cand[(size_t)row * kCosineCandCap + t];
        } else {
            j = p.irregular_list[t - (int)cnt];
        }
        if (nearby && (fabsf(pu - p.cur_uv[2 * j]) > p.max_col || fabsf(pv - p.cur_uv[2 * j + 1]) > p.max_row)) {
            continue;
        }
        const float dot = eigen_dot_octet(x, p.cur + (size_t)j * p.dim, p.dim, c, base);
        const float d = 0.5f - dot / na / p.cur_norm[j] * 0.5f;
        if (d < best_d || (d == best_d && j < best_j)) {
            best_d = d;
            best_j = j;
        }
    }
    // strict '<' against a running minimum that starts at the threshold (descriptor_matcher.h:68-75, :114-117)
    if (c == 0 && best_j >= 0 && best_d < p.max_distance) {
        p.index_pairs[row] = best_j;
    }
}

}  // namespace

hipError_t cosine_match_launch(const CosineParams &p, hipStream_t stream) {
    if (p.n_ref <= 0 || p.n_cur <= 0) {
        return hipSuccess;
    }
    hipError_t e = hipMemsetAsync(p.row_max, 0, sizeof(uint32_t) * (size_t)p.n_ref_pad, stream);  // key 0 = "no candidate yet"
    if (e != hipSuccess) {
        return e;
    }
    e = hipMemsetAsync(p.cand_count, 0, sizeof(uint32_t) * (size_t)p.n_ref_pad, stream);
    if (e != hipSuccess) {
        return e;
    }
    e = hipMemsetAsync(p.irregular_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(cosine_prep_kernel, dim3((unsigned)((p.n_ref_pad * 8 + 255) / 256)), dim3(256), 0, stream, p, 0);
    hipLaunchKernelGGL(cosine_prep_kernel, dim3((unsigned)((p.n_cur_pad * 8 + 255) / 256)), dim3(256), 0, stream, p, 1);
    const int row_tiles = p.n_ref_pad / kTile;
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
    hipLaunchKernelGGL(cosine_recheck_kernel, dim3((unsigned)((p.n_ref * 8 + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ftk
