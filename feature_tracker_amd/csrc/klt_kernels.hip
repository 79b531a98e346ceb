// klt_kernels.hip — pyramidal Lucas-Kanade trackers for gfx950 (MI355X), hand-written HIP.
//
// One 64-lane wavefront (= one workgroup) owns one feature for the whole call: it walks the
// pyramid coarse -> fine and runs every Gauss-Newton iteration without leaving the CU, so a
// TrackFeatures call is ONE launch and the only HBM traffic is the patch footprints it samples
// (the pyramids of both frames total 0.8-5.5 MB and stay L2 / Infinity-Cache resident).
//
// Work split inside the wave, per iteration:
//   phase A (64-wide)  : lane l handles patch pixels l, l+64, ...: bilinear samples (byte loads,
//                        rows of a patch are contiguous so a wave touches <= 2 cache lines per
//                        patch row), gradients, residual, and the per-pixel PRODUCTS of every
//                        normal-equation entry, written to LDS as terms[k][pixel].
//   phase B (K lanes)  : lane k < K adds terms[k][0..P) strictly in row-major pixel order
//                        (ds_read_b128 + 4 dependent v_add_f32).  This is the reference's
//                        sequential accumulation order, so sums are bit-identical to the scalar
//                        CPU path; the Gauss-Newton convergence test (||v||^2 < 4e-2) amplifies
//                        any 1-ulp deviation into extra/missing iterations, which is why a
//                        shuffle-tree reduction is not used here.
//   phase C (uniform)  : results are broadcast (v_readlane), every lane solves the 2x2 / 3x3 /
//                        6x6 system with an Eigen-compatible pivoted LDLT held in registers and
//                        applies the update and the status logic redundantly (no divergence).
//
// Arithmetic contract: IEEE fp32, no FMA contraction (-ffp-contract=off), correctly rounded
// division / sqrt, bilinear weights and summation order exactly as the reference writes them.
//
// Reference behaviour implemented here (file:line relative to the reference repo):
//   basic_klt.cpp:7-181, basic_klt_fast.cpp:7-195, affine_klt.cpp:6-273, affine_klt_fast.cpp:7-188,
//   lssd_klt.cpp:7-250, lssd_klt_fast.cpp:7-229, optical_flow.cpp:49-102.
#include "ftk_device.h"

#include <limits.h>
#include <math.h>

namespace ftk {
namespace {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------

// static_cast<int32_t>(float) as x86-64 cvttss2si does it (out of range / NaN -> INT_MIN)
__device__ __forceinline__ int f2i(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : INT_MIN; }
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ float floor_from_trunc(float x, int t) {
    const float f = (float)t;
    return (f > x) ? f - 1.0f : f;
}
__device__ __forceinline__ float px(const DevImage &im, int row, int col) { return (float)im.data[(long long)row * im.cols + col]; }

// GrayImage::GetPixelValueNoCheck(float, float): bilinear, ((tl + tr) + bl) + br
__device__ __forceinline__ float bilinear(const DevImage &im, float row, float col) {
    int r0 = f2i(row);
    int c0 = f2i(col);
    const float sub_row = row - floor_from_trunc(row, r0);
    const float sub_col = col - floor_from_trunc(col, c0);
    r0 = r0 < 0 ? 0 : (r0 > im.rows - 1 ? im.rows - 1 : r0);
    c0 = c0 < 0 ? 0 : (c0 > im.cols - 1 ? im.cols - 1 : c0);
    const int r1 = (r0 + 1 < im.rows) ? r0 + 1 : r0;
    const int c1 = (c0 + 1 < im.cols) ? c0 + 1 : c0;
    const float inv_sub_row = 1.0f - sub_row;
    const float inv_sub_col = 1.0f - sub_col;
    const float w_tl = inv_sub_row * inv_sub_col;
    const float w_tr = inv_sub_row * sub_col;
    const float w_bl = sub_row * inv_sub_col;
    const float w_br = sub_row * sub_col;
    return w_tl * px(im, r0, c0) + w_tr * px(im, r0, c1) + w_bl * px(im, r1, c0) + w_br * px(im, r1, c1);
}

// GrayImage::GetPixelValue(row, col, *value): closed-rectangle validity, NaN invalid
__device__ __forceinline__ bool sample(const DevImage &im, float row, float col, float &value) {
    if (!(row >= 0.0f && col >= 0.0f && row <= (float)(im.rows - 1) && col <= (float)(im.cols - 1))) {
        return false;
    }
    value = bilinear(im, row, col);
    return true;
}

__device__ __forceinline__ bool uv_outside(float u, float v, const DevImage &im) {
    return u < 0.0f || u > (float)(im.cols - 1) || v < 0.0f || v > (float)(im.rows - 1);
}

__device__ __forceinline__ float bcast(float x, int src_lane) { return __shfl(x, src_lane, kWave); }

template <typename T>
__device__ __forceinline__ void swap_values(T &a, T &b) {
    const T t = a;
    a = b;
    b = t;
}

// ---------------------------------------------------------------------------------------------
// Eigen-compatible LDLT solve, N in {2, 3, 6}, everything in registers (all loops unrolled,
// pivot swaps predicated on compile-time indices so nothing is dynamically indexed).
// Mirrors the published Eigen 3.3.7+ algorithm; see oracle/oracle_substrate.c for the statement.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void ldlt_solve(float (&m)[N][N], const float (&b)[N], float (&x)[N]) {
    int tr[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        tr[k] = k;
    }
    bool degenerate = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (!degenerate) {
            int p = k;
            float biggest = fabsf(m[k][k]);
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const float cand = fabsf(m[i][i]);
                if (cand > biggest) {
                    biggest = cand;
                    p = i;
                }
            }
            tr[k] = p;
#pragma unroll
            for (int q = k + 1; q < N; ++q) {
                if (p == q) {
#pragma unroll
                    for (int j = 0; j < k; ++j) {
                        swap_values(m[k][j], m[q][j]);
                    }
#pragma unroll
                    for (int i = q + 1; i < N; ++i) {
                        swap_values(m[i][k], m[i][q]);
                    }
                    swap_values(m[k][k], m[q][q]);
#pragma unroll
                    for (int i = k + 1; i < q; ++i) {
                        swap_values(m[i][k], m[q][i]);
                    }
                }
            }
            if (k > 0) {
                float temp[N];
#pragma unroll
                for (int j = 0; j < k; ++j) {
                    temp[j] = m[j][j] * m[k][j];
                }
                float dot = m[k][0] * temp[0];
#pragma unroll
                for (int j = 1; j < k; ++j) {
                    dot += m[k][j] * temp[j];
                }
                m[k][k] -= dot;
#pragma unroll
                for (int i = k + 1; i < N; ++i) {
                    float s = m[i][0] * temp[0];
#pragma unroll
                    for (int j = 1; j < k; ++j) {
                        s += m[i][j] * temp[j];
                    }
                    m[i][k] -= s;
                }
            }
            const float akk = m[k][k];
            const bool pivot_valid = fabsf(akk) > 0.0f;
            if (k == 0 && !pivot_valid) {
                tr[0] = 0;
                degenerate = true;
            } else if (pivot_valid) {
#pragma unroll
                for (int i = k + 1; i < N; ++i) {
                    m[i][k] /= akk;
                }
            }
        }
    }

    float y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        y[i] = b[i];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int q = k + 1; q < N; ++q) {
            if (tr[k] == q) {
                swap_values(y[k], y[q]);
            }
        }
    }
#pragma unroll
    for (int i = 1; i < N; ++i) {
        float s = m[i][0] * y[0];
#pragma unroll
        for (int j = 1; j < i; ++j) {
            s += m[i][j] * y[j];
        }
        y[i] -= s;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (fabsf(m[i][i]) > 1.17549435e-38f) {
            y[i] /= m[i][i];
        } else {
            y[i] = 0.0f;
        }
    }
#pragma unroll
    for (int i = N - 2; i >= 0; --i) {
        float s = m[i + 1][i] * y[i + 1];
#pragma unroll
        for (int j = i + 2; j < N; ++j) {
            s += m[j][i] * y[j];
        }
        y[i] -= s;
    }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
#pragma unroll
        for (int q = k + 1; q < N; ++q) {
            if (tr[k] == q) {
                swap_values(y[k], y[q]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i] = y[i];
    }
}

// ---------------------------------------------------------------------------------------------
// LDS carve-up of one workgroup (dynamic shared memory)
// ---------------------------------------------------------------------------------------------
struct Carve {
    float *terms;    // [K][Ppad] per-pixel products, k-major
    float *a0;       // 4 float arrays of Epad entries each (meaning depends on the variant)
    float *a1;
    float *a2;
    float *a3;
    uint8_t *flagsE;  // E bytes
    uint8_t *flagsP;  // P bytes
};

__host__ __device__ inline int pad4(int x) { return (x + 3) & ~3; }

__device__ __forceinline__ Carve carve_lds(float *base, int K, const KltParams &p) {
    Carve c;
    const int epad = pad4(p.E);
    c.terms = base;
    c.a0 = c.terms + K * p.Ppad;
    c.a1 = c.a0 + epad;
    c.a2 = c.a1 + epad;
    c.a3 = c.a2 + epad;
    c.flagsE = reinterpret_cast<uint8_t *>(c.a3 + epad);
    c.flagsP = c.flagsE + epad;
    return c;
}

// Phase B: lane k < K returns sum_{px} terms[k][px] accumulated strictly left to right.
__device__ __forceinline__ float chain_sum(const float *terms, int K, int Ppad, int lane) {
    float acc = 0.0f;
    if (lane < K) {
        const float4 *t = reinterpret_cast<const float4 *>(terms + lane * Ppad);
        const int n4 = Ppad >> 2;
        for (int i = 0; i < n4; ++i) {
            const float4 v = t[i];
            acc += v.x;
            acc += v.y;
            acc += v.z;
            acc += v.w;
        }
    }
    return acc;
}

__device__ __forceinline__ void zero_term_padding(float *terms, int K, const KltParams &p, int lane) {
    const int extra = p.Ppad - p.P;
    for (int k = 0; k < K; ++k) {
        if (lane < extra) {
            terms[k * p.Ppad + p.P + lane] = 0.0f;
        }
    }
}

__device__ __forceinline__ void pixel_rc(const KltParams &p, int pxi, int &prow, int &pcol) {
    prow = (p.patch_cols == 1) ? pxi : (int)__umulhi((unsigned)pxi, p.magic_pc);
    pcol = pxi - prow * p.patch_cols;
}

// ---------------------------------------------------------------------------------------------
// Shared pieces of the non-fast variants (inverse / direct)
//   a0 = gx (right - left), a1 = gy (bottom - top), a2 = i_ref, flagsP = ref-side validity.
// For METHOD == inverse the five reference-image fetches of a pixel do not change within a
// level, so they are sampled once per level; the arithmetic per pixel is unchanged.
// ---------------------------------------------------------------------------------------------
template <int METHOD>
__device__ __forceinline__ void nonfast_level_setup(const KltParams &p, const DevImage &ref, float ref_u, float ref_v, Carve &c, int lane) {
    for (int base = 0; base < p.P; base += kWave) {
        const int pxi = base + lane;
        if (pxi < p.P) {
            int prow, pcol;
            pixel_rc(p, pxi, prow, pcol);
            const float row_i = (float)(prow - p.half_rows) + ref_v;
            const float col_i = (float)(pcol - p.half_cols) + ref_u;
            float left = 0.0f, right = 0.0f, top = 0.0f, bottom = 0.0f, i_ref = 0.0f;
            bool ok;
            if (METHOD == FTK_METHOD_INVERSE) {
                ok = sample(ref, row_i, col_i - 1.0f, left) && sample(ref, row_i, col_i + 1.0f, right) && sample(ref, row_i - 1.0f, col_i, top) &&
                     sample(ref, row_i + 1.0f, col_i, bottom) && sample(ref, row_i, col_i, i_ref);
                c.a0[pxi] = right - left;
                c.a1[pxi] = bottom - top;
            } else {
                ok = sample(ref, row_i, col_i, i_ref);
            }
            c.a2[pxi] = i_ref;
            c.flagsP[pxi] = ok ? 1 : 0;
        }
    }
    __syncthreads();
}

// Completes the six fetches of one patch pixel for the current iteration.
template <int METHOD>
__device__ __forceinline__ bool nonfast_gather(const DevImage &cur, const Carve &c, int pxi, float row_j, float col_j, float &gx, float &gy,
                                               float &i_ref, float &i_cur) {
    bool ok = c.flagsP[pxi] != 0;
    i_ref = c.a2[pxi];
    i_cur = 0.0f;
    if (METHOD == FTK_METHOD_INVERSE) {
        gx = c.a0[pxi];
        gy = c.a1[pxi];
        ok = ok && sample(cur, row_j, col_j, i_cur);
    } else {
        float left = 0.0f, right = 0.0f, top = 0.0f, bottom = 0.0f;
        const bool g = sample(cur, row_j, col_j - 1.0f, left) && sample(cur, row_j, col_j + 1.0f, right) && sample(cur, row_j - 1.0f, col_j, top) &&
                       sample(cur, row_j + 1.0f, col_j, bottom) && sample(cur, row_j, col_j, i_cur);
        gx = right - left;
        gy = bottom - top;
        ok = ok && g;
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------
// Shared pieces of the fast variants
//   a0 = extended reference patch (E), flagsE = its validity, a1 = dx (P), a2 = dy (P)
// ---------------------------------------------------------------------------------------------

// OpticalFlow::ExtractExtendPatchInReferenceImage (optical_flow.cpp:49-102); returns the valid count.
__device__ __forceinline__ uint32_t extract_extended_patch(const KltParams &p, const DevImage &ref, float ref_u, float ref_v, float *ex,
                                                           uint8_t *exv, int lane) {
    const float int_row = floorf(ref_v);
    const float int_col = floorf(ref_u);
    const float dec_row = ref_v - int_row;
    const float dec_col = ref_u - int_col;
    const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
    const float w_tr = (1.0f - dec_row) * dec_col;
    const float w_bl = dec_row * (1.0f - dec_col);
    const float w_br = dec_row * dec_col;
    const int min_row = wadd(f2i(int_row), -(p.ex_rows / 2));
    const int min_col = wadd(f2i(int_col), -(p.ex_cols / 2));
    uint32_t count = 0;
    for (int base = 0; base < p.E; base += kWave) {
        const int e = base + lane;
        bool valid = false;
        if (e < p.E) {
            const int erow = (int)__umulhi((unsigned)e, p.magic_exc);
            const int ecol = e - erow * p.ex_cols;
            const int row = wadd(min_row, erow);
            const int col = wadd(min_col, ecol);
            valid = !(row < 0 || row > ref.rows - 2 || col < 0 || col > ref.cols - 2);
            float value = 0.0f;
            if (valid) {
                value = w_tl * px(ref, row, col) + w_tr * px(ref, row, col + 1) + w_bl * px(ref, row + 1, col) + w_br * px(ref, row + 1, col + 1);
            }
            ex[e] = value;
            exv[e] = valid ? 1 : 0;
        }
        count += (uint32_t)__popcll(__ballot(valid));
    }
    __syncthreads();
    return count;
}

// Central differences on the extended patch: dx = dy = 0 where a 4-neighbour is invalid
// (basic_klt_fast.cpp:64-99, affine_klt_fast.cpp:71-138, lssd_klt_fast.cpp:116-143).
__device__ __forceinline__ bool ex_gradient(const KltParams &p, const float *ex, const uint8_t *exv, int prow, int pcol, float &dx, float &dy) {
    const int ei = (prow + 1) * p.ex_cols + pcol + 1;
    if (exv[ei - 1] && exv[ei + 1] && exv[ei - p.ex_cols] && exv[ei + p.ex_cols]) {
        dx = ex[ei + 1] - ex[ei - 1];
        dy = ex[ei + p.ex_cols] - ex[ei - p.ex_cols];
        return true;
    }
    dx = 0.0f;
    dy = 0.0f;
    return false;
}

// Large-step / convergence bookkeeping shared by the three fast variants
// (basic_klt_fast.cpp:49-60, affine_klt_fast.cpp:55-67, lssd_klt_fast.cpp:101-112).
// Returns true when the iteration loop has to stop.
__device__ __forceinline__ bool fast_step_logic(const KltParams &p, float squared_step, float &last_squared_step, uint32_t &large_step_cnt,
                                                uint8_t &status) {
    if (squared_step < last_squared_step) {
        last_squared_step = squared_step;
        large_step_cnt = 0;
    } else {
        ++large_step_cnt;
        if (large_step_cnt >= p.max_large_step) {
            return true;
        }
    }
    if (squared_step < p.converge) {
        status = FTK_TRACKED;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// Basic KLT (translation only, 2x2)
// ---------------------------------------------------------------------------------------------
struct BasicState {
    float cur_u, cur_v;
};

// TrackOneFeature, basic_klt.cpp:88-181.  Terms: 0 H00, 1 H11, 2 H01, 3 -fx*ft, 4 -fy*ft.
template <int METHOD>
__device__ __forceinline__ void basic_level(const KltParams &p, const DevImage &ref, const DevImage &cur, float ref_u, float ref_v, BasicState &s,
                                            uint8_t &status, uint32_t &iters, Carve &c, int lane) {
    nonfast_level_setup<METHOD>(p, ref, ref_u, ref_v, c, lane);
    for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
        ++iters;
        uint32_t n_valid = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float row_j = (float)(prow - p.half_rows) + s.cur_v;
                const float col_j = (float)(pcol - p.half_cols) + s.cur_u;
                float fx, fy, i_ref, i_cur;
                ok = nonfast_gather<METHOD>(cur, c, pxi, row_j, col_j, fx, fy, i_ref, i_cur);
                const float ft = i_cur - i_ref;
                c.terms[0 * p.Ppad + pxi] = ok ? fx * fx : 0.0f;
                c.terms[1 * p.Ppad + pxi] = ok ? fy * fy : 0.0f;
                c.terms[2 * p.Ppad + pxi] = ok ? fx * fy : 0.0f;
                c.terms[3 * p.Ppad + pxi] = ok ? -(fx * ft) : 0.0f;
                c.terms[4 * p.Ppad + pxi] = ok ? -(fy * ft) : 0.0f;
            }
            n_valid += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        if (n_valid == 0) {
            break;
        }
        const float acc = chain_sum(c.terms, 5, p.Ppad, lane);
        __syncthreads();
        float m[2][2];
        float b[2], v[2];
        m[0][0] = bcast(acc, 0);
        m[1][1] = bcast(acc, 1);
        m[0][1] = m[1][0] = bcast(acc, 2);
        b[0] = bcast(acc, 3);
        b[1] = bcast(acc, 4);
        ldlt_solve<2>(m, b, v);
        if (isnan(v[0]) || isnan(v[1])) {
            status = FTK_NUMERIC_ERROR;
            break;
        }
        s.cur_u += v[0];
        s.cur_v += v[1];
        if (uv_outside(s.cur_u, s.cur_v, cur)) {
            status = FTK_OUTSIDE;
            break;
        }
        if (v[0] * v[0] + v[1] * v[1] < p.converge) {
            status = FTK_TRACKED;
            break;
        }
    }
}

// TrackOneFeatureFast, basic_klt_fast.cpp:7-195.
__device__ __forceinline__ void basic_level_fast(const KltParams &p, const DevImage &ref, const DevImage &cur, float ref_u, float ref_v,
                                                 BasicState &s, uint8_t &status, uint32_t &iters, Carve &c, int lane) {
    float *ex = c.a0, *dxs = c.a1, *dys = c.a2;
    uint8_t *exv = c.flagsE;
    if (extract_extended_patch(p, ref, ref_u, ref_v, ex, exv, lane) == 0) {
        status = FTK_OUTSIDE;
        return;
    }
    // dx, dy and the fixed Hessian: terms 0 dx*dx, 1 dx*dy, 2 dy*dy
    for (int base = 0; base < p.P; base += kWave) {
        const int pxi = base + lane;
        if (pxi < p.P) {
            int prow, pcol;
            pixel_rc(p, pxi, prow, pcol);
            float dx, dy;
            ex_gradient(p, ex, exv, prow, pcol, dx, dy);
            dxs[pxi] = dx;
            dys[pxi] = dy;
            c.terms[0 * p.Ppad + pxi] = dx * dx;
            c.terms[1 * p.Ppad + pxi] = dx * dy;
            c.terms[2 * p.Ppad + pxi] = dy * dy;
        }
    }
    __syncthreads();
    const float hacc = chain_sum(c.terms, 3, p.Ppad, lane);
    __syncthreads();
    const float h00 = bcast(hacc, 0), h01 = bcast(hacc, 1), h11 = bcast(hacc, 2);

    status = FTK_LARGE_RESIDUAL;
    float last_squared_step = INFINITY;
    uint32_t large_step_cnt = 0;
    for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
        ++iters;
        // ComputeBias (:101-195): one weight set from frac(cur), integer lattice floor(cur) - patch/2
        const float int_row = floorf(s.cur_v);
        const float int_col = floorf(s.cur_u);
        const float dec_row = s.cur_v - int_row;
        const float dec_col = s.cur_u - int_col;
        const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
        const float w_tr = (1.0f - dec_row) * dec_col;
        const float w_bl = dec_row * (1.0f - dec_col);
        const float w_br = dec_row * dec_col;
        const int min_row = wadd(f2i(int_row), -(p.patch_rows / 2));
        const int min_col = wadd(f2i(int_col), -(p.patch_cols / 2));
        uint32_t n_valid = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const int row = wadd(min_row, prow);
                const int col = wadd(min_col, pcol);
                const int ei = (prow + 1) * p.ex_cols + pcol + 1;
                ok = !(row < 0 || row > cur.rows - 2 || col < 0 || col > cur.cols - 2) && exv[ei] != 0;
                float t0 = 0.0f, t1 = 0.0f;
                if (ok) {
                    const float i_cur =
                        w_tl * px(cur, row, col) + w_tr * px(cur, row, col + 1) + w_bl * px(cur, row + 1, col) + w_br * px(cur, row + 1, col + 1);
                    const float dt = i_cur - ex[ei];
                    t0 = -(dxs[pxi] * dt);
                    t1 = -(dys[pxi] * dt);
                }
                c.terms[0 * p.Ppad + pxi] = t0;
                c.terms[1 * p.Ppad + pxi] = t1;
            }
            n_valid += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        if (n_valid == 0) {
            break;
        }
        const float acc = chain_sum(c.terms, 2, p.Ppad, lane);
        __syncthreads();
        float m[2][2] = {{h00, h01}, {h01, h11}};
        float b[2] = {bcast(acc, 0), bcast(acc, 1)};
        float v[2];
        ldlt_solve<2>(m, b, v);
        if (isnan(v[0]) || isnan(v[1])) {
            status = FTK_NUMERIC_ERROR;
            break;
        }
        s.cur_u += v[0];
        s.cur_v += v[1];
        if (fast_step_logic(p, v[0] * v[0] + v[1] * v[1], last_squared_step, large_step_cnt, status)) {
            break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Affine KLT (6x6).  Chain indices of the 18 distinct Hessian product sequences + 6 bias ones.
// H(1,2) = H(0,3), H(1,4) = H(0,5) are the same products; H(3,4) is yy*dxdy as in the reference
// (affine_klt.cpp:245, sic), i.e. the same sequence as H(2,3).
// ---------------------------------------------------------------------------------------------
enum {
    A_XX_DXDX, A_XX_DXDY, A_XY_DXDX, A_XY_DXDY, A_X_DXDX, A_X_DXDY, A_XX_DYDY, A_XY_DYDY, A_X_DYDY, A_YY_DXDX, A_YY_DXDY, A_Y_DXDX, A_Y_DXDY,
    A_YY_DYDY, A_Y_DYDY, A_DXDX, A_DXDY, A_DYDY, A_B0, A_B1, A_B2, A_B3, A_B4, A_B5, A_COUNT
};

struct AffineState {
    float cur_u, cur_v;
    float a00, a01, a10, a11;
};

__device__ __forceinline__ void affine_hessian_terms(const KltParams &p, float *terms, int pxi, bool ok, float x, float y, float dx, float dy) {
    const float xx = x * x, yy = y * y, xy = x * y;
    const float dxdx = dx * dx, dydy = dy * dy, dxdy = dx * dy;
    terms[A_XX_DXDX * p.Ppad + pxi] = ok ? xx * dxdx : 0.0f;
    terms[A_XX_DXDY * p.Ppad + pxi] = ok ? xx * dxdy : 0.0f;
    terms[A_XY_DXDX * p.Ppad + pxi] = ok ? xy * dxdx : 0.0f;
    terms[A_XY_DXDY * p.Ppad + pxi] = ok ? xy * dxdy : 0.0f;
    terms[A_X_DXDX * p.Ppad + pxi] = ok ? x * dxdx : 0.0f;
    terms[A_X_DXDY * p.Ppad + pxi] = ok ? x * dxdy : 0.0f;
    terms[A_XX_DYDY * p.Ppad + pxi] = ok ? xx * dydy : 0.0f;
    terms[A_XY_DYDY * p.Ppad + pxi] = ok ? xy * dydy : 0.0f;
    terms[A_X_DYDY * p.Ppad + pxi] = ok ? x * dydy : 0.0f;
    terms[A_YY_DXDX * p.Ppad + pxi] = ok ? yy * dxdx : 0.0f;
    terms[A_YY_DXDY * p.Ppad + pxi] = ok ? yy * dxdy : 0.0f;
    terms[A_Y_DXDX * p.Ppad + pxi] = ok ? y * dxdx : 0.0f;
    terms[A_Y_DXDY * p.Ppad + pxi] = ok ? y * dxdy : 0.0f;
    terms[A_YY_DYDY * p.Ppad + pxi] = ok ? yy * dydy : 0.0f;
    terms[A_Y_DYDY * p.Ppad + pxi] = ok ? y * dydy : 0.0f;
    terms[A_DXDX * p.Ppad + pxi] = ok ? dxdx : 0.0f;
    terms[A_DXDY * p.Ppad + pxi] = ok ? dxdy : 0.0f;
    terms[A_DYDY * p.Ppad + pxi] = ok ? dydy : 0.0f;
}

// bias(0) -= dt * x * dx ... (affine_klt.cpp:251-256, affine_klt_fast.cpp:174-179); first_bias_chain
// is A_B0 in the non-fast layout and 0 in the fast one.
__device__ __forceinline__ void affine_bias_terms(const KltParams &p, float *terms, int first_bias_chain, int pxi, bool ok, float dt, float x,
                                                  float y, float dx, float dy) {
    terms[(first_bias_chain + 0) * p.Ppad + pxi] = ok ? -(dt * x * dx) : 0.0f;
    terms[(first_bias_chain + 1) * p.Ppad + pxi] = ok ? -(dt * x * dy) : 0.0f;
    terms[(first_bias_chain + 2) * p.Ppad + pxi] = ok ? -(dt * y * dx) : 0.0f;
    terms[(first_bias_chain + 3) * p.Ppad + pxi] = ok ? -(dt * y * dy) : 0.0f;
    terms[(first_bias_chain + 4) * p.Ppad + pxi] = ok ? -(dt * dx) : 0.0f;
    terms[(first_bias_chain + 5) * p.Ppad + pxi] = ok ? -(dt * dy) : 0.0f;
}

__device__ __forceinline__ void affine_fill_matrix(float acc, float (&m)[6][6]) {
    const float h00 = bcast(acc, A_XX_DXDX), h01 = bcast(acc, A_XX_DXDY), h02 = bcast(acc, A_XY_DXDX), h03 = bcast(acc, A_XY_DXDY);
    const float h04 = bcast(acc, A_X_DXDX), h05 = bcast(acc, A_X_DXDY), h11 = bcast(acc, A_XX_DYDY), h13 = bcast(acc, A_XY_DYDY);
    const float h15 = bcast(acc, A_X_DYDY), h22 = bcast(acc, A_YY_DXDX), h23 = bcast(acc, A_YY_DXDY), h24 = bcast(acc, A_Y_DXDX);
    const float h25 = bcast(acc, A_Y_DXDY), h33 = bcast(acc, A_YY_DYDY), h35 = bcast(acc, A_Y_DYDY), h44 = bcast(acc, A_DXDX);
    const float h45 = bcast(acc, A_DXDY), h55 = bcast(acc, A_DYDY);
    const float h12 = h03, h14 = h05, h34 = h23;
    const float u[6][6] = {{h00, h01, h02, h03, h04, h05}, {h01, h11, h12, h13, h14, h15}, {h02, h12, h22, h23, h24, h25},
                           {h03, h13, h23, h33, h34, h35}, {h04, h14, h24, h34, h44, h45}, {h05, h15, h25, h35, h45, h55}};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            m[i][j] = u[i][j];
        }
    }
}

__device__ __forceinline__ void affine_apply_step(AffineState &s, const float (&z)[6], const float (&v)[2]) {
    s.cur_u += v[0];
    s.cur_v += v[1];
    s.a00 += z[0];
    s.a10 += z[1];
    s.a01 += z[2];
    s.a11 += z[3];
}

// TrackOneFeature, affine_klt.cpp:93-273.
template <int METHOD>
__device__ __forceinline__ void affine_level(const KltParams &p, const DevImage &ref, const DevImage &cur, float ref_u, float ref_v,
                                             AffineState &s, uint8_t &status, uint32_t &iters, Carve &c, int lane) {
    nonfast_level_setup<METHOD>(p, ref, ref_u, ref_v, c, lane);
    for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
        ++iters;
        uint32_t n_valid = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float dcol = (float)(pcol - p.half_cols);
                const float drow = (float)(prow - p.half_rows);
                const float warped_x = s.a00 * dcol + s.a01 * drow;
                const float warped_y = s.a10 * dcol + s.a11 * drow;
                const float row_j = warped_y + s.cur_v;
                const float col_j = warped_x + s.cur_u;
                float dx, dy, i_ref, i_cur;
                ok = nonfast_gather<METHOD>(cur, c, pxi, row_j, col_j, dx, dy, i_ref, i_cur);
                const float dt = i_cur - i_ref;
                affine_hessian_terms(p, c.terms, pxi, ok, col_j, row_j, dx, dy);
                affine_bias_terms(p, c.terms, A_B0, pxi, ok, dt, col_j, row_j, dx, dy);
            }
            n_valid += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        if (n_valid == 0) {
            break;
        }
        const float acc = chain_sum(c.terms, A_COUNT, p.Ppad, lane);
        __syncthreads();
        float m[6][6];
        affine_fill_matrix(acc, m);
        const float b[6] = {bcast(acc, A_B0), bcast(acc, A_B1), bcast(acc, A_B2), bcast(acc, A_B3), bcast(acc, A_B4), bcast(acc, A_B5)};
        float z[6];
        ldlt_solve<6>(m, b, z);
        float v[2];
        v[0] = (z[0] * s.cur_u + z[2] * s.cur_v) + z[4];
        v[1] = (z[1] * s.cur_u + z[3] * s.cur_v) + z[5];
        if (isnan(v[0]) || isnan(v[1])) {
            status = FTK_NUMERIC_ERROR;
            break;
        }
        affine_apply_step(s, z, v);
        if (uv_outside(s.cur_u, s.cur_v, cur)) {
            status = FTK_OUTSIDE;
            break;
        }
        if (v[0] * v[0] + v[1] * v[1] < p.converge) {
            status = FTK_TRACKED;
            break;
        }
    }
}

// TrackOneFeatureFast, affine_klt_fast.cpp:7-188.
__device__ __forceinline__ void affine_level_fast(const KltParams &p, const DevImage &ref, const DevImage &cur, float ref_u, float ref_v,
                                                  AffineState &s, uint8_t &status, uint32_t &iters, Carve &c, int lane) {
    float *ex = c.a0, *dxs = c.a1, *dys = c.a2;
    uint8_t *exv = c.flagsE;
    if (extract_extended_patch(p, ref, ref_u, ref_v, ex, exv, lane) == 0) {
        status = FTK_OUTSIDE;
        return;
    }
    // H once per level, anchored at cur_uv on level entry (:95-96)
    for (int base = 0; base < p.P; base += kWave) {
        const int pxi = base + lane;
        if (pxi < p.P) {
            int prow, pcol;
            pixel_rc(p, pxi, prow, pcol);
            float dx, dy;
            const bool has_gradient = ex_gradient(p, ex, exv, prow, pcol, dx, dy);
            dxs[pxi] = dx;
            dys[pxi] = dy;
            const float x = (float)(pcol - p.half_cols) + s.cur_u;
            const float y = (float)(prow - p.half_rows) + s.cur_v;
            affine_hessian_terms(p, c.terms, pxi, has_gradient, x, y, dx, dy);
        }
    }
    __syncthreads();
    const float hacc = chain_sum(c.terms, A_B0, p.Ppad, lane);
    __syncthreads();
    float h[6][6];
    affine_fill_matrix(hacc, h);

    status = FTK_LARGE_RESIDUAL;
    float last_squared_step = INFINITY;
    uint32_t large_step_cnt = 0;
    for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
        ++iters;
        uint32_t n_valid = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float dcol = (float)(pcol - p.half_cols);
                const float drow = (float)(prow - p.half_rows);
                const float warped_x = s.a00 * dcol + s.a01 * drow;
                const float warped_y = s.a10 * dcol + s.a11 * drow;
                const float row_c = warped_y + s.cur_v;
                const float col_c = warped_x + s.cur_u;
                float i_cur = 0.0f;
                const int ei = (prow + 1) * p.ex_cols + pcol + 1;
                ok = sample(cur, row_c, col_c, i_cur) && exv[ei] != 0;
                const float dt = i_cur - ex[ei];
                affine_bias_terms(p, c.terms, 0, pxi, ok, dt, col_c, row_c, dxs[pxi], dys[pxi]);
            }
            n_valid += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        if (n_valid == 0) {
            break;
        }
        const float acc = chain_sum(c.terms, 6, p.Ppad, lane);
        __syncthreads();
        float m[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                m[i][j] = h[i][j];
            }
        }
        const float b[6] = {bcast(acc, 0), bcast(acc, 1), bcast(acc, 2), bcast(acc, 3), bcast(acc, 4), bcast(acc, 5)};
        float z[6];
        ldlt_solve<6>(m, b, z);
        if (isnan(z[0]) || isnan(z[1]) || isnan(z[2]) || isnan(z[3]) || isnan(z[4]) || isnan(z[5])) {
            status = FTK_NUMERIC_ERROR;
            break;
        }
        float v[2];
        v[0] = (z[0] * s.cur_u + z[2] * s.cur_v) + z[4];
        v[1] = (z[1] * s.cur_u + z[3] * s.cur_v) + z[5];
        affine_apply_step(s, z, v);
        if (fast_step_logic(p, v[0] * v[0] + v[1] * v[1], last_squared_step, large_step_cnt, status)) {
            break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LSSD KLT (SE(2), 3x3).  Chains: 0 j0j0, 1 j0j1, 2 j0j2, 3 j1j1, 4 j1j2, 5 j2j2, 6..8 -j_i*r.
// ---------------------------------------------------------------------------------------------
struct LssdState {
    float r00, r01, r10, r11;
    float t0, t1;
};

__device__ __forceinline__ void se2_apply(const LssdState &s, float x, float y, float &ox, float &oy) {
    ox = (s.r00 * x + s.r01 * y) + s.t0;
    oy = (s.r10 * x + s.r11 * y) + s.t1;
}

// delta_R << 1, -theta, theta, 1; R *= delta_R; R /= R.col(0).norm(); t += v.tail<2>()
__device__ __forceinline__ void se2_update(LssdState &s, const float (&v)[3]) {
    const float theta = v[0];
    const float d00 = 1.0f, d01 = -theta, d10 = theta, d11 = 1.0f;
    const float n00 = s.r00 * d00 + s.r01 * d10;
    const float n01 = s.r00 * d01 + s.r01 * d11;
    const float n10 = s.r10 * d00 + s.r11 * d10;
    const float n11 = s.r10 * d01 + s.r11 * d11;
    const float norm = sqrtf(n00 * n00 + n10 * n10);
    s.r00 = n00 / norm;
    s.r01 = n01 / norm;
    s.r10 = n10 / norm;
    s.r11 = n11 / norm;
    s.t0 += v[1];
    s.t1 += v[2];
}

// Vec3::squaredNorm(): Eigen's unrolled reduction of a fixed 3-vector is a0 + (a1 + a2)
__device__ __forceinline__ float vec3_squared_norm(const float (&v)[3]) { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }

__device__ __forceinline__ void lssd_terms(const KltParams &p, float *terms, int pxi, bool ok, float j0, float j1, float j2, float residual) {
    terms[0 * p.Ppad + pxi] = ok ? j0 * j0 : 0.0f;
    terms[1 * p.Ppad + pxi] = ok ? j0 * j1 : 0.0f;
    terms[2 * p.Ppad + pxi] = ok ? j0 * j2 : 0.0f;
    terms[3 * p.Ppad + pxi] = ok ? j1 * j1 : 0.0f;
    terms[4 * p.Ppad + pxi] = ok ? j1 * j2 : 0.0f;
    terms[5 * p.Ppad + pxi] = ok ? j2 * j2 : 0.0f;
    terms[6 * p.Ppad + pxi] = ok ? -(j0 * residual) : 0.0f;
    terms[7 * p.Ppad + pxi] = ok ? -(j1 * residual) : 0.0f;
    terms[8 * p.Ppad + pxi] = ok ? -(j2 * residual) : 0.0f;
}

// Solves the 3x3 system from the nine chain sums and applies the SE(2) update.
// Returns false (after setting status) on NaN.
__device__ __forceinline__ bool lssd_solve_and_update(float acc, LssdState &s, float (&v)[3], uint8_t &status) {
    const float h00 = bcast(acc, 0), h01 = bcast(acc, 1), h02 = bcast(acc, 2), h11 = bcast(acc, 3), h12 = bcast(acc, 4), h22 = bcast(acc, 5);
    float m[3][3] = {{h00, h01, h02}, {h01, h11, h12}, {h02, h12, h22}};
    const float b[3] = {bcast(acc, 6), bcast(acc, 7), bcast(acc, 8)};
    ldlt_solve<3>(m, b, v);
    if (isnan(v[0]) || isnan(v[1]) || isnan(v[2])) {
        status = FTK_NUMERIC_ERROR;
        return false;
    }
    se2_update(s, v);
    return true;
}

// TrackOneFeature, lssd_klt.cpp:96-250.  a3 = i_cur of the current iteration.
template <int METHOD>
__device__ __forceinline__ void lssd_level(const KltParams &p, const DevImage &ref, const DevImage &cur, float ref_u, float ref_v, LssdState &s,
                                           uint8_t &status, uint32_t &iters, Carve &c, int lane) {
    nonfast_level_setup<METHOD>(p, ref, ref_u, ref_v, c, lane);
    uint8_t *okflags = c.flagsE;  // per-iteration validity of a pixel (all six fetches)
    for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
        ++iters;
        // pass 1 (:140-184): validity mask and the two patch means (sequential sums)
        uint32_t n_valid = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float row_i = (float)(prow - p.half_rows) + ref_v;
                const float col_i = (float)(pcol - p.half_cols) + ref_u;
                float row_j, col_j;
                se2_apply(s, col_i, row_i, col_j, row_j);
                float gx, gy, i_ref, i_cur;
                ok = nonfast_gather<METHOD>(cur, c, pxi, row_j, col_j, gx, gy, i_ref, i_cur);
                if (METHOD != FTK_METHOD_INVERSE) {
                    c.a0[pxi] = gx;
                    c.a1[pxi] = gy;
                }
                c.a3[pxi] = i_cur;
                okflags[pxi] = ok ? 1 : 0;
                c.terms[0 * p.Ppad + pxi] = ok ? i_ref : 0.0f;
                c.terms[1 * p.Ppad + pxi] = ok ? i_cur : 0.0f;
            }
            n_valid += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        const float macc = chain_sum(c.terms, 2, p.Ppad, lane);
        __syncthreads();
        const float ref_average = bcast(macc, 0) / (float)n_valid;
        const float cur_average = bcast(macc, 1) / (float)n_valid;
        const float grad_average = (METHOD == FTK_METHOD_INVERSE) ? ref_average : cur_average;

        // pass 2 (:186-247): mean-normalised Jacobian and residual
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float row_i = (float)(prow - p.half_rows) + ref_v;
                const float col_i = (float)(pcol - p.half_cols) + ref_u;
                const bool ok = okflags[pxi] != 0;
                const float jp0 = c.a0[pxi] / grad_average;
                const float jp1 = c.a1[pxi] / grad_average;
                const float s0 = s.r00 * (-row_i) + s.r01 * col_i;
                const float s1 = s.r10 * (-row_i) + s.r11 * col_i;
                const float j0 = jp0 * s0 + jp1 * s1;
                const float j1 = jp0 * 1.0f + jp1 * 0.0f;
                const float j2 = jp0 * 0.0f + jp1 * 1.0f;
                const float residual = c.a3[pxi] / cur_average - c.a2[pxi] / ref_average;
                lssd_terms(p, c.terms, pxi, ok, j0, j1, j2, residual);
            }
        }
        __syncthreads();
        if (n_valid == 0) {
            break;
        }
        const float acc = chain_sum(c.terms, 9, p.Ppad, lane);
        __syncthreads();
        float v[3];
        if (!lssd_solve_and_update(acc, s, v, status)) {
            break;
        }
        if (vec3_squared_norm(v) < p.converge) {
            status = FTK_TRACKED;
            break;
        }
    }
}

// TrackOneFeatureFast, lssd_klt_fast.cpp:7-229.  a3 = current patch, flagsP = its validity.
__device__ __forceinline__ void lssd_level_fast(const KltParams &p, const DevImage &ref, const DevImage &cur, float ref_u, float ref_v,
                                                LssdState &s, uint8_t &status, uint32_t &iters, Carve &c, int lane) {
    float *ex = c.a0, *dxs = c.a1, *dys = c.a2, *curp = c.a3;
    uint8_t *exv = c.flagsE, *curv = c.flagsP;
    const uint32_t ref_valid_num = extract_extended_patch(p, ref, ref_u, ref_v, ex, exv, lane);
    if (ref_valid_num == 0) {
        status = FTK_OUTSIDE;
        return;
    }
    for (int base = 0; base < p.P; base += kWave) {
        const int pxi = base + lane;
        if (pxi < p.P) {
            int prow, pcol;
            pixel_rc(p, pxi, prow, pcol);
            float dx, dy;
            ex_gradient(p, ex, exv, prow, pcol, dx, dy);
            dxs[pxi] = dx;
            dys[pxi] = dy;
            // interior of the extended patch in row-major order == the P patch pixels
            c.terms[pxi] = ex[(prow + 1) * p.ex_cols + pcol + 1];
        }
    }
    __syncthreads();
    if (p.consider_luminance) {
        // :27-46 — numerator: interior of the extended patch; denominator: valid count of the WHOLE extended patch
        const float racc = chain_sum(c.terms, 1, p.Ppad, lane);
        __syncthreads();
        const float ref_average = bcast(racc, 0) / (float)ref_valid_num;
        for (int i = lane; i < p.P; i += kWave) {
            dxs[i] /= ref_average;
            dys[i] /= ref_average;
        }
        for (int i = lane; i < p.E; i += kWave) {
            ex[i] /= ref_average;
        }
        __syncthreads();
    }

    status = FTK_LARGE_RESIDUAL;
    float last_squared_step = INFINITY;
    uint32_t large_step_cnt = 0;
    for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
        ++iters;
        // ExtractPatchInCurrentImage (:145-195)
        float centre_u, centre_v;
        se2_apply(s, ref_u, ref_v, centre_u, centre_v);
        const int min_row = wadd(f2i(centre_v), -p.patch_rows);
        const int min_col = wadd(f2i(centre_u), -p.patch_cols);
        const int max_row = wadd(min_row, p.patch_rows * 2);
        const int max_col = wadd(min_col, p.patch_cols * 2);
        const bool partly_outside = (min_row < 0 || max_row > cur.rows - 2 || min_col < 0 || max_col > cur.cols - 2);
        uint32_t cur_valid_num = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float row_i = (float)(prow - p.half_rows) + ref_v;
                const float col_i = (float)(pcol - p.half_cols) + ref_u;
                float row_j, col_j;
                se2_apply(s, col_i, row_i, col_j, row_j);
                float value = 0.0f;
                if (partly_outside) {
                    ok = sample(cur, row_j, col_j, value);
                    if (!ok) {
                        value = 0.0f;
                    }
                } else {
                    value = bilinear(cur, row_j, col_j);
                    ok = true;
                }
                curp[pxi] = value;
                curv[pxi] = ok ? 1 : 0;
                // :65-71 — the mean numerator only covers patch rows / cols 1 .. size-2
                const bool interior = prow >= 1 && prow < p.patch_rows - 1 && pcol >= 1 && pcol < p.patch_cols - 1;
                c.terms[pxi] = interior ? value : 0.0f;
            }
            cur_valid_num += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        if (cur_valid_num == 0) {
            break;
        }
        if (p.consider_luminance) {
            const float cacc = chain_sum(c.terms, 1, p.Ppad, lane);
            __syncthreads();
            const float cur_average = bcast(cacc, 0) / (float)cur_valid_num;
            for (int i = lane; i < p.P; i += kWave) {
                curp[i] /= cur_average;
            }
            __syncthreads();
        }

        // ComputeHessianAndBias (:197-229)
        uint32_t n_valid = 0;
        for (int base = 0; base < p.P; base += kWave) {
            const int pxi = base + lane;
            bool ok = false;
            if (pxi < p.P) {
                int prow, pcol;
                pixel_rc(p, pxi, prow, pcol);
                const float row_i = (float)(prow - p.half_rows) + ref_v;
                const float col_i = (float)(pcol - p.half_cols) + ref_u;
                const int ei = (prow + 1) * p.ex_cols + pcol + 1;
                ok = exv[ei] != 0 && curv[pxi] != 0;
                const float s0 = s.r00 * (-row_i) + s.r01 * col_i;
                const float s1 = s.r10 * (-row_i) + s.r11 * col_i;
                const float dx = dxs[pxi], dy = dys[pxi];
                const float j0 = dx * s0 + dy * s1;
                const float residual = curp[pxi] - ex[ei];
                lssd_terms(p, c.terms, pxi, ok, j0, dx, dy, residual);
            }
            n_valid += (uint32_t)__popcll(__ballot(ok));
        }
        __syncthreads();
        if (n_valid == 0) {
            break;
        }
        const float acc = chain_sum(c.terms, 9, p.Ppad, lane);
        __syncthreads();
        float v[3];
        if (!lssd_solve_and_update(acc, s, v, status)) {
            break;
        }
        if (fast_step_logic(p, vec3_squared_norm(v), last_squared_step, large_step_cnt, status)) {
            break;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-feature driver: TrackMultipleLevel / TrackSingleLevel of the three trackers
// (basic_klt.cpp:7-86, affine_klt.cpp:6-91, lssd_klt.cpp:7-94).
// ---------------------------------------------------------------------------------------------
template <int MODEL>
struct ChainCount;
template <>
struct ChainCount<FTK_MODEL_BASIC> {
    static constexpr int value = 5;
};
template <>
struct ChainCount<FTK_MODEL_AFFINE> {
    static constexpr int value = A_COUNT;
};
template <>
struct ChainCount<FTK_MODEL_LSSD> {
    static constexpr int value = 9;
};

template <int MODEL, int METHOD>
__global__ void __launch_bounds__(kWave) klt_track_kernel(const KltParams p) {
    extern __shared__ float4 lds_raw[];
    const int lane = threadIdx.x;
    const uint32_t id = blockIdx.x;
    if (id >= (uint32_t)p.n) {
        return;
    }
    const float in_u = p.cur_uv_in[2 * id], in_v = p.cur_uv_in[2 * id + 1];
    uint8_t status = p.status_in[id];
    // features beyond kMaxTrackPointsNumber and features that already failed are passed through
    if (id >= p.n_track || status > FTK_TRACKED) {
        if (lane == 0) {
            p.cur_uv_out[2 * id] = in_u;
            p.cur_uv_out[2 * id + 1] = in_v;
            p.status_out[id] = status;
            if (p.iters) {
                p.iters[id] = 0;
            }
        }
        return;
    }

    constexpr int K = ChainCount<MODEL>::value;
    Carve c = carve_lds(reinterpret_cast<float *>(lds_raw), K, p);
    zero_term_padding(c.terms, K, p, lane);

    const float full_ref_u = p.ref_uv[2 * id], full_ref_v = p.ref_uv[2 * id + 1];
    const float scale = p.single_level ? 1.0f : (float)(1 << (p.n_levels - 1));
    float ref_u = p.single_level ? full_ref_u : full_ref_u / scale;
    float ref_v = p.single_level ? full_ref_v : full_ref_v / scale;
    const float scur_u = p.single_level ? in_u : in_u / scale;
    const float scur_v = p.single_level ? in_v : in_v / scale;

    BasicState bs = {scur_u, scur_v};
    AffineState as = {scur_u, scur_v, 1.0f, 0.0f, 0.0f, 1.0f};
    if (p.single_level) {
        as.a00 = p.prior[0];  // affine_klt.cpp:70 — the prediction is only honoured on the single-level path
        as.a01 = p.prior[1];
        as.a10 = p.prior[2];
        as.a11 = p.prior[3];
    }
    LssdState ls;
    ls.r00 = p.prior[0];
    ls.r01 = p.prior[1];
    ls.r10 = p.prior[2];
    ls.r11 = p.prior[3];
    ls.t0 = scur_u - (p.prior[0] * ref_u + p.prior[1] * ref_v);  // lssd_klt.cpp:23
    ls.t1 = scur_v - (p.prior[2] * ref_u + p.prior[3] * ref_v);

    uint32_t iters = 0;
    float out_u = in_u, out_v = in_v;
    for (int level = p.n_levels - 1; level > -1; --level) {
        const DevImage ref = p.ref[level];
        const DevImage cur = p.cur[level];
        if (MODEL == FTK_MODEL_BASIC) {
            if (METHOD == FTK_METHOD_FAST) {
                basic_level_fast(p, ref, cur, ref_u, ref_v, bs, status, iters, c, lane);
            } else {
                basic_level<METHOD>(p, ref, cur, ref_u, ref_v, bs, status, iters, c, lane);
            }
        } else if (MODEL == FTK_MODEL_AFFINE) {
            if (METHOD == FTK_METHOD_FAST) {
                affine_level_fast(p, ref, cur, ref_u, ref_v, as, status, iters, c, lane);
            } else {
                affine_level<METHOD>(p, ref, cur, ref_u, ref_v, as, status, iters, c, lane);
            }
        } else {
            if (METHOD == FTK_METHOD_FAST) {
                lssd_level_fast(p, ref, cur, ref_u, ref_v, ls, status, iters, c, lane);
            } else {
                lssd_level<METHOD>(p, ref, cur, ref_u, ref_v, ls, status, iters, c, lane);
            }
        }
        __syncthreads();

        if (level == 0) {
            if (MODEL == FTK_MODEL_BASIC) {
                out_u = bs.cur_u;
                out_v = bs.cur_v;
            } else if (MODEL == FTK_MODEL_AFFINE) {
                out_u = as.cur_u;
                out_v = as.cur_v;
            } else if (!p.single_level) {
                // lssd_klt.cpp:43 — written back with the UNSCALED ref; the single-level path
                // never writes cur_pixel_uv (lssd_klt.cpp:72-89, sic)
                out_u = (ls.r00 * full_ref_u + ls.r01 * full_ref_v) + ls.t0;
                out_v = (ls.r10 * full_ref_u + ls.r11 * full_ref_v) + ls.t1;
            }
            break;
        }
        ref_u *= 2.0f;
        ref_v *= 2.0f;
        if (MODEL == FTK_MODEL_BASIC) {
            bs.cur_u *= 2.0f;
            bs.cur_v *= 2.0f;
        } else if (MODEL == FTK_MODEL_AFFINE) {
            as.cur_u *= 2.0f;
            as.cur_v *= 2.0f;
        } else {
            ls.t0 *= 2.0f;
            ls.t1 *= 2.0f;
        }
    }

    if (uv_outside(out_u, out_v, p.cur[0])) {
        status = FTK_OUTSIDE;
    }
    if (lane == 0) {
        p.cur_uv_out[2 * id] = out_u;
        p.cur_uv_out[2 * id + 1] = out_v;
        p.status_out[id] = status;
        if (p.iters) {
            p.iters[id] = iters;
        }
    }
}

template <int MODEL, int METHOD>
hipError_t launch_variant(const KltParams &p, size_t lds_bytes, hipStream_t stream) {
    auto kernel = klt_track_kernel<MODEL, METHOD>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) {
            return e;
        }
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)p.n), dim3(kWave), lds_bytes, stream, p);
    return hipGetLastError();
}

int chain_count(int model) {
    switch (model) {
        case FTK_MODEL_BASIC: return ChainCount<FTK_MODEL_BASIC>::value;
        case FTK_MODEL_AFFINE: return ChainCount<FTK_MODEL_AFFINE>::value;
        case FTK_MODEL_LSSD: return ChainCount<FTK_MODEL_LSSD>::value;
        default: return 0;
    }
}

}  // namespace

size_t klt_lds_bytes(int model, int method, const KltParams &p) {
    (void)method;
    const int k = chain_count(model);
    if (k == 0) {
        return 0;
    }
    const size_t epad = (size_t)pad4(p.E);
    return sizeof(float) * ((size_t)k * p.Ppad + 4 * epad) + 2 * epad;
}

hipError_t klt_launch(int model, int method, const KltParams &p, hipStream_t stream) {
    const size_t lds = klt_lds_bytes(model, method, p);
    const int m = (method == FTK_METHOD_INVERSE || method == FTK_METHOD_DIRECT) ? method : FTK_METHOD_FAST;
#define FTK_DISPATCH(MODEL)                                                               \
    switch (m) {                                                                          \
        case FTK_METHOD_INVERSE: return launch_variant<MODEL, FTK_METHOD_INVERSE>(p, lds, stream); \
        case FTK_METHOD_DIRECT: return launch_variant<MODEL, FTK_METHOD_DIRECT>(p, lds, stream);   \
        default: return launch_variant<MODEL, FTK_METHOD_FAST>(p, lds, stream);                    \
    }
    switch (model) {
        case FTK_MODEL_BASIC: FTK_DISPATCH(FTK_MODEL_BASIC)
        case FTK_MODEL_AFFINE: FTK_DISPATCH(FTK_MODEL_AFFINE)
        case FTK_MODEL_LSSD: FTK_DISPATCH(FTK_MODEL_LSSD)
        default: return hipErrorInvalidValue;
    }
#undef FTK_DISPATCH
}

}  // namespace ftk
