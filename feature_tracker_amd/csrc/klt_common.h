// klt_common.h — device helpers shared by the tracker kernels (klt_kernels.hip: all nine variants on
// the generic path; klt_basic_kernels.hip: the pipelined Basic-KLT inverse kernel): scalar
// semantics of the reference (x86 float->int, bilinear samplers), LDS image windows and their
// staging, the Eigen-compatible LDLT, and the exact-order chain primitives.
#pragma once

#include "ftk_device.h"

#include <limits.h>
#include <math.h>
#include <stddef.h>

#include <type_traits>

namespace ftk {
namespace {

constexpr int kWave = 64;

// Diagnostic build only (-DFTK_STAMPS): per-phase cycle totals of every workgroup, written to a
// side buffer that no other code reads.  The production build contains none of this.
#ifdef FTK_STAMPS
#define FTK_STAMP_BEGIN(b) (b).stamp_t0 = __builtin_amdgcn_s_memtime()
#define FTK_STAMP_END(b, k)                                           \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        (b).stamp_acc[k] += now_ - (b).stamp_t0;                      \
        (b).stamp_t0 = now_;                                          \
    } while (0)
#else
#define FTK_STAMP_BEGIN(b)
#define FTK_STAMP_END(b, k)
#endif

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------

// static_cast<int32_t>(float) as x86-64 cvttss2si does it (out of range / NaN -> INT_MIN)
__device__ __forceinline__ int f2i(float x) { return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : INT_MIN; }
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
// Product of two SMALL integers (patch / window geometry, pixel and row indices inside them: |x| < 2^23, product < 2^31): the
// 24-bit multiplier issues at full rate, v_mul_lo_u32 at a quarter of it — and the kernels are bound by vector issue.
__device__ __forceinline__ int imul(int a, int b) { return __mul24(a, b); }
// __ballot() takes an int: a bool predicate goes through v_cndmask 0/1 + v_cmp_ne before it becomes the lane mask it already was.
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
__device__ __forceinline__ float floor_from_trunc(float x, int t) {
    const float f = (float)t;
    return (f > x) ? f - 1.0f : f;
}
// 32-bit offset on the 24-bit multiplier (levels are < 2^24 on a side and < 2^32 pixels: checked where a pyramid is made), so the
// load is base + zero-extended offset instead of a quarter-rate 64-bit multiply-add per tap
__device__ __forceinline__ float px(const DevImage &im, int row, int col) { return (float)im.data[__umul24((unsigned)row, (unsigned)im.cols) + (unsigned)col]; }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Thread coordinates inside the feature's workgroup.
struct Blk {
    int tid, nt, lane, wave, nwaves;
    bool solo = false;  // compile-time true in the one-wave instantiations: no workgroup barrier anywhere, cross-wave exchanges fold away
    bool tree = false;  // throughput mode (KltParams::tree): sums by per-lane partials + a butterfly instead of the exact-order chain
#ifdef FTK_STAMPS
    mutable unsigned long long stamp_t0 = 0;
    mutable unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
};

// Workgroup barrier; for a one-wave feature only a compiler fence (LDS operations of one wave execute in program order).
__device__ __forceinline__ void blk_sync(const Blk &b) {
    if (b.solo) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// Returns x behind an optimisation barrier: per-thread index math derived from it is recomputed where
// it is used instead of being hoisted out of the level loop and held in VGPRs for the whole kernel.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ Blk opaque_blk(const Blk &b) {
    Blk o = b;
    o.tid = opaque(b.tid);
    o.lane = opaque(b.lane);
    return o;
}

// The argument block is ~0.7 KB = a dozen 64-byte lines of the scalar cache, cold at the start of a launch, and the compiler loads a
// field where it is first needed — a dependent miss (an L2 round trip) every few dozen instructions of the prologue, for every
// wave of the launch at once.  One dword of every line is requested up front instead: the misses overlap into one round trip and
// the later loads hit.
template <size_t kBytes>
__device__ __forceinline__ void klt_touch_kernarg() {
    const __attribute__((address_space(4))) uint32_t *ka = (const __attribute__((address_space(4))) uint32_t *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t sink = 0;
#pragma unroll
    for (size_t off = 0; off < kBytes; off += 64) {
        sink |= ka[off / 4];
    }
    asm volatile("" ::"s"(sink));
}

// All features of a call are resident at once and the hardware arbitrates oldest-wave-first, which
// lets the first-dispatched features finish early and leaves the youngest ones to run the tail
// alone at single-wave issue rate.  Waves therefore raise their own priority while they are
// behind (coarse pyramid level) and lower it as they advance: the co-resident features progress
// evenly and the last one finishes ~10 % earlier (measured: 53.9 -> 49.3 us span at 2000 features).
// `younger`: the workgroup sits in the later-dispatched half of the launch.  With 2 000 identical features the finishing time
// still correlates 0.87 with the launch slot (31 us for the first 250 workgroups, 36 for the last 250, all started within
// 0.4 us: scripts/stamps_placement.py), so the younger half runs one step above the older half of the same level:
// config 2 40.4 -> 39.0 us per step, 300 / 1 000-feature launches -3.5 %, config 5 shard -1.7 %, config 1 +-0.  (Rotating or
// purely age-based priorities, and a finer split, all lose: docs/LAB_NOTES.md.)
__device__ __forceinline__ void set_level_priority(int level, bool younger = false) {
#ifdef FTK_NO_LEVEL_PRIO
    (void)level;
    (void)younger;
    return;
#endif
    const int pr = (level >= 3 ? 3 : level) + (younger ? 1 : 0);
    if (pr >= 3) {
        __builtin_amdgcn_s_setprio(3);
    } else if (pr == 2) {
        __builtin_amdgcn_s_setprio(2);
    } else if (pr == 1) {
        __builtin_amdgcn_s_setprio(1);
    } else {
        __builtin_amdgcn_s_setprio(0);
    }
}

// An image window resident in LDS.  Element (r, c) packs the pixel pair
// (img[clamp(r_lo + r)][clamp(c_lo + c)], img[clamp(r_lo + r)][clamp(c_lo + c + 1)]) into 16 bits,
// clamp = clamp-to-edge, so the 2x2 neighbourhood of any in-image pixel is two ds_read_u16.
struct Win {
    const uint16_t *data;
    int r_lo, c_lo;
    int rows, cols;  // wave-uniform
    // floor(v) in [cover[0], cover[1]] and floor(u) in [cover[2], cover[3]]: the footprint of a patch centred at (u, v) lies inside
    // this window (win_set_cover; an empty interval when the window's origin is too large for the test to be exact in fp32)
    float cover[4];
};

__device__ __forceinline__ uint16_t window_element(const DevImage &im, int r_lo, int c_lo, int idx, int wcols, uint32_t magic_cols);

// The 2x2 neighbourhood of the in-image pixel (r0, c0), +1 neighbours clamped to the image:
// from the LDS window when it covers the pixel, from global memory otherwise.
__device__ __forceinline__ void fetch4(const DevImage &im, const Win &w, int r0, int c0, float &p00, float &p01, float &p10, float &p11) {
    const int lr = (int)((unsigned)r0 - (unsigned)w.r_lo);
    const int lc = (int)((unsigned)c0 - (unsigned)w.c_lo);
    if ((unsigned)lr < (unsigned)(w.rows - 1) && (unsigned)lc < (unsigned)w.cols) {
        const int top = imul(lr, w.cols) + lc;  // the row below is one pitch further (not a second multiply of lr + 1)
        const unsigned a = w.data[top];
        const unsigned bb = w.data[top + w.cols];
        p00 = (float)(a & 0xFFu);
        p01 = (float)(a >> 8);
        p10 = (float)(bb & 0xFFu);
        p11 = (float)(bb >> 8);
    } else {
        const int r1 = (r0 + 1 < im.rows) ? r0 + 1 : r0;
        const int c1 = (c0 + 1 < im.cols) ? c0 + 1 : c0;
        p00 = px(im, r0, c0);
        p01 = px(im, r0, c1);
        p10 = px(im, r1, c0);
        p11 = px(im, r1, c1);
    }
}

// GrayImage::GetPixelValueNoCheck(float, float): bilinear, ((tl + tr) + bl) + br, any coordinates
__device__ __forceinline__ float bilinear(const DevImage &im, const Win &w, float row, float col) {
    int r0 = f2i(row);
    int c0 = f2i(col);
    const float sub_row = row - floor_from_trunc(row, r0);
    const float sub_col = col - floor_from_trunc(col, c0);
    r0 = clampi(r0, 0, im.rows - 1);
    c0 = clampi(c0, 0, im.cols - 1);
    const float inv_sub_row = 1.0f - sub_row;
    const float inv_sub_col = 1.0f - sub_col;
    const float w_tl = inv_sub_row * inv_sub_col;
    const float w_tr = inv_sub_row * sub_col;
    const float w_bl = sub_row * inv_sub_col;
    const float w_br = sub_row * sub_col;
    float p00, p01, p10, p11;
    fetch4(im, w, r0, c0, p00, p01, p10, p11);
    return w_tl * p00 + w_tr * p01 + w_bl * p10 + w_br * p11;
}

// bilinear() for coordinates the caller knows to be inside the image with room for the +1 neighbours (0 <= row <= rows - 2,
// 0 <= col <= cols - 2, finite): truncation is floor there and no index needs clamping, so the same values come out of fewer
// instructions.
__device__ __forceinline__ float bilinear_inside(const DevImage &im, const Win &w, float row, float col) {
    const int r0 = (int)row;
    const int c0 = (int)col;
    // x - (float)(int)x for 0 <= x < 2^31 is x - floor(x), exact in fp32, and that is what v_fract_f32 returns (its clamp below
    // 1.0 only matters for tiny negative x): one instruction instead of a conversion and a subtraction, same bits
    const float sub_row = __builtin_amdgcn_fractf(row);
    const float sub_col = __builtin_amdgcn_fractf(col);
    const float inv_sub_row = 1.0f - sub_row;
    const float inv_sub_col = 1.0f - sub_col;
    const float w_tl = inv_sub_row * inv_sub_col;
    const float w_tr = inv_sub_row * sub_col;
    const float w_bl = sub_row * inv_sub_col;
    const float w_br = sub_row * sub_col;
    float p00, p01, p10, p11;
    fetch4(im, w, r0, c0, p00, p01, p10, p11);
    return w_tl * p00 + w_tr * p01 + w_bl * p10 + w_br * p11;
}

// GrayImage::GetPixelValue(row, col, *value): closed-rectangle validity, NaN invalid.  Inside the
// rectangle truncation equals floor, so the fractions need no floor fix-up.
__device__ __forceinline__ bool sample(const DevImage &im, const Win &w, float row, float col, float &value) {
    if (!(row >= 0.0f && col >= 0.0f && row <= (float)(im.rows - 1) && col <= (float)(im.cols - 1))) {
        return false;
    }
    const int r0 = (int)row;
    const int c0 = (int)col;
    const float sub_row = __builtin_amdgcn_fractf(row);  // == row - (float)r0 for row >= 0, see bilinear_inside
    const float sub_col = __builtin_amdgcn_fractf(col);
    const float inv_sub_row = 1.0f - sub_row;
    const float inv_sub_col = 1.0f - sub_col;
    const float w_tl = inv_sub_row * inv_sub_col;
    const float w_tr = inv_sub_row * sub_col;
    const float w_bl = sub_row * inv_sub_col;
    const float w_br = sub_row * sub_col;
    float p00, p01, p10, p11;
    fetch4(im, w, r0, c0, p00, p01, p10, p11);
    value = w_tl * p00 + w_tr * p01 + w_bl * p10 + w_br * p11;
    return true;
}

// --- straight-line sampling for the hot loops ---------------------------------------------------
// One image axis of a bilinear tap: validity on the closed interval [0, limit], base index, fraction
// and its complement — the same quantities sample() derives, computed without branches.
struct Axis {
    int i0;
    float sub, inv;
    bool valid;
};

__device__ __forceinline__ Axis make_axis(float x, int limit) {
    Axis a;
    a.valid = (x >= 0.0f && x <= (float)limit);
    a.i0 = (int)x;
    a.sub = x - (float)a.i0;
    a.inv = 1.0f - a.sub;
    return a;
}

// Bilinear value of (row axis, col axis) read from the LDS window with NO branch: the LDS index is
// clamped into the window so the read is always safe, and `hit` is cleared when the tap was not
// really covered (the caller then redoes the pixel through sample(), which can reach global memory).
// Same weight products and summation order as sample().
__device__ __forceinline__ float tap(const Win &w, const Axis &ar, const Axis &ac, bool &hit) {
    const int lr = (int)((unsigned)ar.i0 - (unsigned)w.r_lo);
    const int lc = (int)((unsigned)ac.i0 - (unsigned)w.c_lo);
    const bool in = (unsigned)lr < (unsigned)(w.rows - 1) && (unsigned)lc < (unsigned)w.cols;
    hit = hit && in;
    const int idx = in ? imul(lr, w.cols) + lc : 0;
    const unsigned a = w.data[idx];
    const unsigned bb = w.data[idx + w.cols];
    const float w_tl = ar.inv * ac.inv;
    const float w_tr = ar.inv * ac.sub;
    const float w_bl = ar.sub * ac.inv;
    const float w_br = ar.sub * ac.sub;
    return w_tl * (float)(a & 0xFFu) + w_tr * (float)(a >> 8) + w_bl * (float)(bb & 0xFFu) + w_br * (float)(bb >> 8);
}

__device__ __forceinline__ bool uv_outside(float u, float v, const DevImage &im) {
    return u < 0.0f || u > (float)(im.cols - 1) || v < 0.0f || v > (float)(im.rows - 1);
}

template <typename T>
__device__ __forceinline__ void swap_values(T &a, T &b) {
    const T t = a;
    a = b;
    b = t;
}

// Axis tables: the five reference taps of the inverse methods use, per patch pixel, the row axes of
// (row, row-1, row+1) and the column axes of (col, col-1, col+1).  Those depend on the patch row
// (resp. column) only, so they are computed once per level into LDS — 3*(rows+cols) entries instead
// of 6 axes per pixel — with exactly the expressions of the per-pixel form.  Entry layout (float4):
// x = window-relative base index (int bits, 0 when not covered), y = fraction, z = 1 - fraction,
// w = flags (int bits: 1 = inside the image, 2 = covered by the staged window).
__device__ __forceinline__ void build_axis_tables(const Blk &b, const KltParams &p, const DevImage &im, const Win &w, float u, float v,
                                                  int variants, float4 *tab) {
    const int nr = variants * p.patch_rows, total = nr + variants * p.patch_cols;
    for (int t = b.tid; t < total; t += b.nt) {
        const bool is_row = t < nr;
        const int k = is_row ? t : t - nr;
        const int len = is_row ? p.patch_rows : p.patch_cols;
        const int var = (k >= 2 * len) ? 2 : (k >= len ? 1 : 0);
        const int d = k - var * len;
        float x = (float)(d - (is_row ? p.half_rows : p.half_cols)) + (is_row ? v : u);
        if (var == 1) {
            x = x - 1.0f;
        } else if (var == 2) {
            x = x + 1.0f;
        }
        const Axis a = make_axis(x, (is_row ? im.rows : im.cols) - 1);
        const int rel = (int)((unsigned)a.i0 - (unsigned)(is_row ? w.r_lo : w.c_lo));
        const bool hit = is_row ? (unsigned)rel < (unsigned)(w.rows - 1) : (unsigned)rel < (unsigned)w.cols;
        tab[t] = make_float4(__int_as_float(hit ? rel : 0), a.sub, a.inv, __int_as_float((a.valid ? 1 : 0) | (hit ? 2 : 0)));
    }
}

// tap() on two table entries.
__device__ __forceinline__ float tap_table(const Win &w, const float4 &ar, const float4 &ac) {
    const int idx = imul(__float_as_int(ar.x), w.cols) + __float_as_int(ac.x);
    const unsigned a = w.data[idx];
    const unsigned bb = w.data[idx + w.cols];
    const float w_tl = ar.z * ac.z;
    const float w_tr = ar.z * ac.y;
    const float w_bl = ar.y * ac.z;
    const float w_br = ar.y * ac.y;
    return w_tl * (float)(a & 0xFFu) + w_tr * (float)(a >> 8) + w_bl * (float)(bb & 0xFFu) + w_br * (float)(bb >> 8);
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

// ldlt_solve<3> written out for the symmetric 3x3 system of the LSSD trackers (lssd_klt.cpp:107, lssd_klt_fast.cpp:87): the
// same operations in the same order — Eigen's LDLT is left-looking, so the pivot search of step k sees ORIGINAL diagonal
// entries (moved by the swaps) and the two transpositions can be decided up front; only the lower triangle is ever read.
// ~40 % of the instructions of the generic unrolled form (no 3x3 array with predicated swaps of every element).
// Independent divisions of WAVE-UNIFORM values side by side: lane k divides numerator k (the lanes beyond the last repeat it),
// the quotients come back through v_readlane.  One correctly rounded division sequence (~10 instructions) instead of one per
// quotient, and each quotient is the same instruction sequence on the same operands as before — bit-identical.
__device__ __forceinline__ float uniform_lane(float v, int lane_index) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_index)); }
__device__ __forceinline__ void lane_div2(int lane, float n0, float n1, float d, float &q0, float &q1) {
    const float q = (lane == 0 ? n0 : n1) / d;
    q0 = uniform_lane(q, 0);
    q1 = uniform_lane(q, 1);
}

// kLanes: the inputs are wave-uniform and all 64 lanes are executing (the one-wave trackers): independent divisions run on
// separate lanes (lane_div2 and its three-way form below); otherwise every lane divides for itself as before.
template <bool kLanes = false>
__device__ __forceinline__ void ldlt3_solve(float a00, float a10, float a20, float a11, float a21, float a22, const float (&b)[3], float (&x)[3],
                                            int lane = 0) {
    // step 0: pivot = first maximum of |a00|, |a11|, |a22|
    float m00 = a00, m10 = a10, m20 = a20, m11 = a11, m21 = a21, m22 = a22;
    const float d0 = fabsf(m00), d1 = fabsf(m11), d2 = fabsf(m22);
    int p0 = 0;
    float biggest = d0;
    if (d1 > biggest) {
        biggest = d1;
        p0 = 1;
    }
    if (d2 > biggest) {
        p0 = 2;
    }
    if (p0 == 1) {  // rows / columns 0 <-> 1 of the lower triangle: m[2][0] <-> m[2][1], the diagonals
        swap_values(m20, m21);
        swap_values(m00, m11);
    } else if (p0 == 2) {  // 0 <-> 2: the diagonals, m[1][0] <-> m[2][1]
        swap_values(m00, m22);
        swap_values(m10, m21);
    }
    const bool valid0 = fabsf(m00) > 0.0f;
    const bool degenerate = !valid0;  // a zero first pivot: Eigen stops, the transpositions become the identity
    int p1 = 1;
    if (!degenerate) {
        if (kLanes) {
            lane_div2(lane, m10, m20, m00, m10, m20);
        } else {
            m10 /= m00;
            m20 /= m00;
        }
        // step 1: pivot among the remaining diagonals
        if (fabsf(m22) > fabsf(m11)) {
            p1 = 2;
            swap_values(m10, m20);
            swap_values(m11, m22);
        }
        {
            const float temp0 = m00 * m10;
            const float dot = m10 * temp0;
            m11 -= dot;
            const float s = m20 * temp0;
            m21 -= s;
        }
        if (fabsf(m11) > 0.0f) {
            m21 /= m11;
        }
        // step 2
        {
            const float temp0 = m00 * m20, temp1 = m11 * m21;
            float dot = m20 * temp0;
            dot += m21 * temp1;
            m22 -= dot;
        }
    } else {
        p0 = 0;
    }
    // y = P b
    float y0 = b[0], y1 = b[1], y2 = b[2];
    if (p0 == 1) {
        swap_values(y0, y1);
    } else if (p0 == 2) {
        swap_values(y0, y2);
    }
    if (p1 == 2) {
        swap_values(y1, y2);
    }
    // L^-1
    y1 -= m10 * y0;
    {
        float s = m20 * y0;
        s += m21 * y1;
        y2 -= s;
    }
    // D^+
    if (kLanes) {
        const float num = lane == 0 ? y0 : (lane == 1 ? y1 : y2);
        const float den = lane == 0 ? m00 : (lane == 1 ? m11 : m22);
        const float q = (fabsf(den) > 1.17549435e-38f) ? num / den : 0.0f;
        y0 = uniform_lane(q, 0);
        y1 = uniform_lane(q, 1);
        y2 = uniform_lane(q, 2);
    } else {
        y0 = (fabsf(m00) > 1.17549435e-38f) ? y0 / m00 : 0.0f;
        y1 = (fabsf(m11) > 1.17549435e-38f) ? y1 / m11 : 0.0f;
        y2 = (fabsf(m22) > 1.17549435e-38f) ? y2 / m22 : 0.0f;
    }
    // L^-T
    y1 -= m21 * y2;
    {
        float s = m10 * y1;
        s += m20 * y2;
        y0 -= s;
    }
    // P^T
    if (p1 == 2) {
        swap_values(y1, y2);
    }
    if (p0 == 1) {
        swap_values(y0, y1);
    } else if (p0 == 2) {
        swap_values(y0, y2);
    }
    x[0] = y0;
    x[1] = y1;
    x[2] = y2;
}

// ---------------------------------------------------------------------------------------------
// The same 6x6 LDLT, rows spread over lanes (affine trackers, direct method).
//
// ldlt_solve<6> above is ~1 000 dependent instructions on one lane per Gauss-Newton iteration — the
// longest serial stretch of an affine iteration.  Two facts make it parallel without changing a
// single rounding:
//   * Eigen's in-place LDLT is LEFT-looking: step k touches column k only, so the diagonal entries
//     the pivot search looks at are still the ORIGINAL ones (moved by the swaps).  The whole pivot
//     order can therefore be replayed up front on the six diagonal magnitudes alone, with the same
//     strict '>' first-maximum rule and the same position swaps.
//   * with the permutation known, row i of B = P A P^T lives on lane i: column k of L is one
//     multiply-add sweep over all lanes at once (same products, same left-to-right sums per
//     element), the divisions of a column happen side by side, and the triangular solves
//     broadcast one y_j per step.
// a_lds: 36 floats, row-major, symmetric (bitwise).  All 64 lanes of ONE wave call; lanes 0..5 work.
// ---------------------------------------------------------------------------------------------
struct Ldlt6 {
    float l[6];    // lane i: L(i, j) for j < i (unit diagonal implied), of the permuted matrix
    float d_mine;  // lane i: D(i)
    int perm;      // lane i: original index of position i
};

__device__ __forceinline__ float bcast_lane(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }

// Element (i, j) of the symmetric matrix: a row-major 6 x 6 in LDS (the non-fast affine trackers' chain lanes store their 18
// distinct Hessian sums straight into it, klt_kernels.hip affine_dense_slots).
struct Ldlt6Dense {
    const float *a;
    __device__ __forceinline__ float operator()(int i, int j) const { return a[i * 6 + j]; }
};

template <typename Elem>
__device__ __forceinline__ Ldlt6 ldlt6_factor_of(const Elem &elem, int lane);

__device__ __forceinline__ Ldlt6 ldlt6_factor(const float *a_lds, int lane) { return ldlt6_factor_of(Ldlt6Dense{a_lds}, lane); }

// my_ad: |A(i, i)| on lane i < 6 (anything elsewhere); ad_all[j]: the same six magnitudes, wave-uniform.  Callers that hold the
// diagonal in registers (the non-fast affine trackers: the chain lanes' accumulators) pass it in and save two LDS round trips.
template <typename Elem>
__device__ __forceinline__ Ldlt6 ldlt6_factor_diag(float my_ad, const float (&ad_all)[6], const Elem &elem, int lane);

template <typename Elem>
__device__ __forceinline__ Ldlt6 ldlt6_factor_of(const Elem &elem, int lane) {
    const int me0 = lane < 6 ? lane : 5;
    const float my_ad = fabsf(elem(me0, me0));
    float ad_all[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        ad_all[j] = bcast_lane(my_ad, j);
    }
    return ldlt6_factor_diag(my_ad, ad_all, elem, lane);
}

template <typename Elem>
__device__ __forceinline__ Ldlt6 ldlt6_factor_diag(float my_ad, const float (&ad_all)[6], const Elem &elem, int lane) {
    // ---- pivot order, replayed on the diagonal ----
    // Distinct, non-NaN magnitudes (the normal case): selection with swaps is then simply the descending order,
    // and lane i finds its own position as the number of larger magnitudes — six broadcasts instead of a serial
    // selection sort.  Ties, NaN or an all-zero diagonal take the literal replay (first maximum, position swaps).
    int rank = 0, equal = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        rank += (ad_all[j] > my_ad) ? 1 : 0;
        equal += (ad_all[j] == my_ad) ? 1 : 0;
    }
    rank = lane < 6 ? rank : 99;  // lanes that hold no row never match a position
    equal = lane < 6 ? equal : 1;
    const bool irregular = wave_ballot(equal != 1) != 0ull;  // equal == 0: NaN; > 1: a tie
    int pos[6];
    bool degenerate = false;
    if (!irregular) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            pos[k] = __builtin_ctzll(wave_ballot(rank == k));  // exactly one lane holds each rank here: v_cmp + s_ff1
        }
        // the largest magnitude is positive here (six distinct non-negative numbers), so the first pivot is valid
    } else {
        float ad[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ad[i] = ad_all[i];
            pos[i] = i;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int p = k;
            float biggest = ad[k];
#pragma unroll
            for (int i = k + 1; i < 6; ++i) {
                if (ad[i] > biggest) {
                    biggest = ad[i];
                    p = i;
                }
            }
            if (k == 0) {
                // the k == 0 pivot is its (unmodified) diagonal entry: |akk| > 0 fails for an all-zero / NaN diagonal,
                // and the reference then stops factorising (identity order from here on)
                degenerate = !(biggest > 0.0f);
            }
#pragma unroll
            for (int q = k + 1; q < 6; ++q) {
                if (p == q && !(degenerate && k > 0)) {
                    swap_values(ad[k], ad[q]);
                    swap_values(pos[k], pos[q]);
                }
            }
        }
    }
    // ---- row `lane` of the permuted matrix ----
    const int me = lane < 6 ? lane : 5;
    int pi = pos[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) {
        pi = (me == i) ? pos[i] : pi;
    }
    float bmat[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        bmat[j] = elem(pi, pos[j]);
    }
    Ldlt6 f;
    f.perm = pi;
    f.d_mine = 0.0f;
    float dl[6];  // lane i: D(j) * L(i, j) — the product is formed on every lane BEFORE the broadcast (one multiply by a uniform,
                  // off the critical path as soon as column j is known) instead of after it (uniform x uniform: a copy + a multiply)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float m_ik = bmat[k];
        if (k > 0) {
            // temp[j] = D(j) * L(k, j); s = sum_j L(i, j) * temp[j], j ascending — the reference's order
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < k; ++j) {
                const float temp_j = bcast_lane(dl[j], k);
                s = (j == 0) ? f.l[0] * temp_j : s + f.l[j] * temp_j;
            }
            m_ik = degenerate ? m_ik : m_ik - s;
        }
        const float akk = bcast_lane(m_ik, k);
        f.d_mine = (me == k) ? akk : f.d_mine;
        const bool pivot_valid = fabsf(akk) > 0.0f;
        f.l[k] = (pivot_valid && !degenerate) ? m_ik / akk : m_ik;  // meaningful on lanes > k
        dl[k] = akk * f.l[k];
    }
    return f;
}

__device__ __forceinline__ void ldlt6_solve(const Ldlt6 &f, const float *b_lds, float *x_lds, int lane) {
    const int me = lane < 6 ? lane : 5;
    float y = b_lds[f.perm];  // transpositions applied to the right-hand side
    // forward: y_i -= sum_{j<i} L(i, j) y_j, j ascending
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float yj = bcast_lane(y, j);  // final: lane j committed its sum at the end of step j - 1
        s = (j == 0) ? f.l[0] * yj : s + f.l[j] * yj;
        y = (me == j + 1) ? y - s : y;
    }
    // D^+: components with |d| <= FLT_MIN become 0
    y = (fabsf(f.d_mine) > 1.17549435e-38f) ? y / f.d_mine : 0.0f;
    // backward: y_i -= sum_{j>i} L(j, i) y_j, j ascending; the terms of column i live on lanes j > i
#pragma unroll
    for (int i = 4; i >= 0; --i) {
        const float term = f.l[i] * y;  // lane j: L(j, i) * y_j
        float t = bcast_lane(term, i + 1);
#pragma unroll
        for (int j = i + 2; j < 6; ++j) {
            t = t + bcast_lane(term, j);
        }
        y = (me == i) ? y - t : y;
    }
    if (lane < 6) {
        x_lds[f.perm] = y;  // transpositions undone
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

// Where chain lane s < 18 stores its sum in the dense row-major 6 x 6 Hessian (H(1,2) = H(0,3), H(1,4) = H(0,5), H(3,4) = H(2,3): the
// aliases of affine_klt.cpp:264-270 and the (sic) of :245): up to four entries e = 6 i + j per sum, one byte each (sums with fewer
// entries repeat one).  Row i of chain indices: 0 1 2 3 4 5 | 1 6 3 7 5 8 | 2 3 9 10 11 12 | 3 7 10 13 10 14 | 4 5 11 10 15 16 | 5 8 12 14 16 17.
__device__ __forceinline__ uint32_t affine_dense_slots(int sum) {
    constexpr uint32_t kSlots[18] = {
        0x00000000u | 0u * 0x01010101u,                           //  0: H00
        1u | 6u << 8 | 1u << 16 | 6u << 24,                       //  1: H01 H10
        2u | 12u << 8 | 2u << 16 | 12u << 24,                     //  2: H02 H20
        3u | 18u << 8 | 8u << 16 | 13u << 24,                     //  3: H03 H30 H12 H21
        4u | 24u << 8 | 4u << 16 | 24u << 24,                     //  4: H04 H40
        5u | 30u << 8 | 10u << 16 | 25u << 24,                    //  5: H05 H50 H14 H41
        7u * 0x01010101u,                                         //  6: H11
        9u | 19u << 8 | 9u << 16 | 19u << 24,                     //  7: H13 H31
        11u | 31u << 8 | 11u << 16 | 31u << 24,                   //  8: H15 H51
        14u * 0x01010101u,                                        //  9: H22
        15u | 20u << 8 | 22u << 16 | 27u << 24,                   // 10: H23 H32 H34 H43
        16u | 26u << 8 | 16u << 16 | 26u << 24,                   // 11: H24 H42
        17u | 32u << 8 | 17u << 16 | 32u << 24,                   // 12: H25 H52
        21u * 0x01010101u,                                        // 13: H33
        23u | 33u << 8 | 23u << 16 | 33u << 24,                   // 14: H35 H53
        28u * 0x01010101u,                                        // 15: H44
        29u | 34u << 8 | 29u << 16 | 34u << 24,                   // 16: H45 H54
        35u * 0x01010101u,                                        // 17: H55
    };
    return kSlots[sum < 18 ? sum : 17];
}
// chain lanes that hold the diagonal H(j, j), j = 0..5
constexpr int kAffineDiagLane[6] = {A_XX_DXDX, A_XX_DYDY, A_YY_DXDX, A_YY_DYDY, A_DXDX, A_DYDY};

// Phase B: lane k < K of wave 0 adds terms[k][0..Ppad) strictly left to right and publishes the sum.
// The adds form one dependent chain (that IS the reference's order); the LDS reads are software
// pipelined one round (8 x ds_read_b128 = 32 terms) ahead in two ping-pong register sets, pinned
// in place with sched_barrier, so that the chain of v_add_f32 — not the ds_read latency — sets the
// pace.  The prefetch may run up to 16 float4 past the end of a row: it stays inside the
// workgroup's LDS carve (the arrays behind `terms` are larger than that) and is never consumed.
#ifndef FTK_CHAIN_ROUND
#define FTK_CHAIN_ROUND 8
#endif
constexpr int kChainRound = FTK_CHAIN_ROUND;

// The reads of a round are issued LAST-CONSUMED FIRST: LDS data returns in issue order, so the wait in front of the round's first
// add (its float4 was issued last) covers the whole round — one s_waitcnt per round instead of one per float4.  A lone wave issues
// one instruction of ANY kind per 4 cycles (scripts/microbench/dep_add_latency.hip: a dependent v_add_f32 costs 4, an s_nop or
// s_waitcnt in between 4 more), so the waits were a ninth of the chain loop's instructions.
#ifdef FTK_CHAIN_FORWARD_ISSUE
constexpr bool kChainReversedIssue = false;
#else
constexpr bool kChainReversedIssue = true;
#endif
__device__ __forceinline__ void chain_load(float4 (&q)[kChainRound], const float4 *t) {
#pragma unroll
    for (int d = 0; d < kChainRound; ++d) {
        const int e = kChainReversedIssue ? kChainRound - 1 - d : d;
        q[e] = t[e];
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ float chain_consume(float acc, const float4 (&q)[kChainRound], int count) {
#pragma unroll
    for (int d = 0; d < kChainRound; ++d) {
        if (d < count) {
            acc += q[d].x;
            acc += q[d].y;
            acc += q[d].z;
            acc += q[d].w;
        }
    }
    return acc;
}

__device__ __forceinline__ float chain_consume_all(float acc, const float4 (&q)[kChainRound]) {
#pragma unroll
    for (int d = 0; d < kChainRound; ++d) {
        acc += q[d].x;
        acc += q[d].y;
        acc += q[d].z;
        acc += q[d].w;
    }
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

__device__ __forceinline__ float chain_lane(const float *row, int Ppad, float acc = 0.0f) {
    const float4 *t = reinterpret_cast<const float4 *>(row);
    const int n4 = Ppad >> 2;
    float4 qa[kChainRound], qb[kChainRound];
    int i = 0;
    chain_load(qa, t);
    for (; i + 2 * kChainRound <= n4; i += 2 * kChainRound) {
        chain_load(qb, t + i + kChainRound);
        acc = chain_consume_all(acc, qa);
        chain_load(qa, t + i + 2 * kChainRound);
        acc = chain_consume_all(acc, qb);
    }
    chain_load(qb, t + i + kChainRound);
    const int rem = n4 - i;  // 0 .. 2*kChainRound-1 float4 left, the first kChainRound already in qa
    acc = chain_consume(acc, qa, rem);
    acc = chain_consume(acc, qb, rem - kChainRound);
    return acc;
}

// Throughput mode (ftk_set_reduction_mode): the sums of K term rows by ALL 64 lanes of one wave — lane l adds the terms
// l, l + 64, ... of a row, a butterfly adds the lanes; four rows at a time so that the shuffles of one cover the latency of the
// others.  A fixed order, not the reference's: the results are NOT bit-identical to the CPU path (reported, never asserted).
__device__ __forceinline__ void tree_sums(const float *terms, int K, int Ppad, float *sums, int lane) {
    for (int k0 = 0; k0 < K; k0 += 4) {
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = lane; i < Ppad; i += kWave) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[j] += (k0 + j < K) ? terms[(k0 + j) * Ppad + i] : 0.0f;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[j] += __shfl_xor(part[j], off, kWave);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k0 + j < K) {
                    sums[k0 + j] = part[j];
                }
            }
        }
    }
}

// The same chain over a GROUPED layout: the terms of four consecutive pixels of one sum sit in one float4 and consecutive
// groups lie kStride4 float4 apart ([group][sum][4 pixels], written by the pixel lanes with immediate offsets).  Same adds in the
// same order as chain_lane.  `rounds` rounds of kChainRound groups: the caller pads the group count to a multiple of kChainRound
// with zero terms (x + 0 == x for every value a sum can hold), so no load is guarded and none runs past the last group — a
// stride's worth of overrun per prefetched group would leave the carve.
template <int kStride4>
__device__ __forceinline__ void chain_load_groups(float4 (&q)[kChainRound], const float4 *t) {
#pragma unroll
    for (int d = 0; d < kChainRound; ++d) {
        const int e = kChainReversedIssue ? kChainRound - 1 - d : d;
        q[e] = t[e * kStride4];
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int kStride4>
__device__ __forceinline__ float chain_groups(const float4 *t, int rounds, float acc) {
    // Branch-free body (a conditional load of a loop-carried register set costs a copy of the set per trip): the prefetch two
    // rounds ahead always happens, from round 0 again once it would pass the end — in bounds, never consumed.
    float4 qa[kChainRound], qb[kChainRound];
    chain_load_groups<kStride4>(qa, t);
    int r = 0;
#pragma nounroll  // with a compile-time round count (13 x 13 instantiations) full unrolling takes the kernel from 81 to 128 VGPRs
    for (; r + 2 <= rounds; r += 2) {
        chain_load_groups<kStride4>(qb, t + (r + 1) * (kChainRound * kStride4));
        acc = chain_consume_all(acc, qa);
        const int ahead = (r + 2 < rounds) ? r + 2 : 0;  // wave-uniform: a scalar select
        chain_load_groups<kStride4>(qa, t + ahead * (kChainRound * kStride4));
        acc = chain_consume_all(acc, qb);
    }
    if (r < rounds) {
        acc = chain_consume_all(acc, qa);  // odd count: round r was loaded as `ahead`
    }
    return acc;
}

// 64 strictly ordered adds of one chunk row (lane k < kTerms of the consumer wave), 16 terms per batch in two
// ping-pong register sets.  The compiler hoists every ds_read of the unrolled chunk to its top if it may (64 VGPRs
// of terms live at once — the difference between 4 and 5-6 waves per SIMD for this kernel); sched_barrier does not
// stop that, a data dependency does: the address of batch k + 2 is tied to the accumulator after batch k, so its
// reads issue exactly when their registers are free, one batch (16 dependent adds) ahead of their use.
__device__ __forceinline__ int chain_tie(float acc) {
    int zero = 0;
    asm volatile("" : "+v"(zero) : "v"(acc));
    return zero;
}

#if FTK_CHAIN_ROUND == 4  // written for 4 batches of 16 terms (a 64-pixel chunk); translation units with another round do not use it
__device__ __forceinline__ float chain_chunk(float acc, const float *row) {
    const float4 *t = reinterpret_cast<const float4 *>(row);
    float4 qa[kChainRound], qb[kChainRound];
    chain_load(qa, t);
    chain_load(qb, t + kChainRound);
    acc = chain_consume_all(acc, qa);
    chain_load(qa, t + 2 * kChainRound + chain_tie(acc));
    acc = chain_consume_all(acc, qb);
    chain_load(qb, t + 3 * kChainRound + chain_tie(acc));
    acc = chain_consume_all(acc, qa);
    acc = chain_consume_all(acc, qb);
    return acc;
}

// The chunk a patch ENDS in: its lanes [left, 64) wrote exact zeros, and x + (+-0) == x for every value a sum can hold (the sums start
// at +0 and -0 never arises: +0 + (-0) == +0), so only the 16-term batches that hold a patch pixel are read and added — at 13 x 13
// (41 pixels in the third chunk) 48 adds and 12 reads instead of 64 and 16, in every chain pass of every iteration.  Same adds in the
// same order up to the last patch pixel: bit-identical.  `left` = patch pixels in this chunk (wave-uniform; >= 64 for a full chunk).
__device__ __forceinline__ float chain_chunk_left(float acc, const float *row, int left) {
    if (left > 3 * 4 * kChainRound) {
        return chain_chunk(acc, row);
    }
    const float4 *t = reinterpret_cast<const float4 *>(row);
    float4 qa[kChainRound], qb[kChainRound];
    chain_load(qa, t);
    if (left > 2 * 4 * kChainRound) {  // 33 .. 48 pixels: three batches
        chain_load(qb, t + kChainRound);
        acc = chain_consume_all(acc, qa);
        chain_load(qa, t + 2 * kChainRound + chain_tie(acc));
        acc = chain_consume_all(acc, qb);
        return chain_consume_all(acc, qa);
    }
    if (left > 4 * kChainRound) {  // 17 .. 32: two
        chain_load(qb, t + kChainRound);
        acc = chain_consume_all(acc, qa);
        return chain_consume_all(acc, qb);
    }
    return chain_consume_all(acc, qa);  // 1 .. 16: one
}
#endif


// ---------------------------------------------------------------------------------------------
// The exact-order chain through the DPP network ("quad chain", round 5).
//
// chain_chunk above gives every sum ONE lane, which reads 4 of its terms per ds_read_b128 and adds them: one read instruction per
// 4 terms — and a lone wave issues one instruction of ANY kind per 4 cycles and pays 9 - 14 for an LDS read, so a term costs 8.5 - 10.7
// cycles although the dependent add itself needs 4 (scripts/microbench/dpp_quad_chain.hip, V0).  Here every sum has a QUAD of lanes,
// all four holding the same running value: lane i of quad q reads the float4 at terms 16 j + 4 i of sum q, so ONE ds_read_b128 brings
// 16 consecutive terms of up to 16 sums into the wave, and sixteen `v_add_f32_dpp acc, T, acc quad_perm:[i,i,i,i]` add them in
// order — quad lane 0's x, y, z, w, then lane 1's, ... — the DPP network broadcasting lane i's register to its quad.  The same
// IEEE adds in the same order (the DPP operand is the TERM; the accumulator is the plain src1, forwarded from the add before:
// no wait state, 64 000 random sums of 448 terms bit-identical), 4.84 cycles per term from registers, 6.6 with a chunk's four reads
// issued up front (V2 / V4 of the microbenchmark) against 10.7.  The loads are ordinary C++ loads (the compiler places their waits);
// the adds are one asm block per 16 terms.  Every lane of the wave must be active (EXEC all ones: no VALU write of EXEC in front of
// a DPP instruction); quads beyond the last sum follow any valid row and their results are ignored.
// ---------------------------------------------------------------------------------------------
#define FTK_QADD(i, reg) "v_add_f32_dpp %0, " reg ", %0 quad_perm:[" #i "," #i "," #i "," #i "] row_mask:0xf bank_mask:0xf\n"
#define FTK_QADD4(i) FTK_QADD(i, "%1") FTK_QADD(i, "%2") FTK_QADD(i, "%3") FTK_QADD(i, "%4")
// acc += the quad's 16 terms in order.  `s_nop 1`: should the compiler have MOVED a term register with a VALU instruction right in
// front of this block, the two wait states a DPP operand needs after a VALU write are there (it cannot see the DPP in the asm).
__device__ __forceinline__ float chain_quad_step16(float acc, const float4 q) {
    asm volatile("s_nop 1\n" FTK_QADD4(0) FTK_QADD4(1) FTK_QADD4(2) FTK_QADD4(3) : "+v"(acc) : "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
#ifdef FTK_QUAD_EXTRA_ZERO_ADDS  // experiment: what do 16 more adds per 16 terms cost (acc + 0 == acc)?
    {
        const float z = 0.0f;
        asm volatile("s_nop 1\n" FTK_QADD4(0) FTK_QADD4(1) FTK_QADD4(2) FTK_QADD4(3) : "+v"(acc) : "v"(z), "v"(z), "v"(z), "v"(z));
    }
#endif
    return acc;
}

// 64 terms (four reads) in one block: one `s_nop` and one wait for the reads per 64 adds instead of per 16 — a lone wave issues an
// instruction of any kind every 4 - 5 cycles, so each instruction that is not an add costs as much as a term.
#ifndef FTK_QUAD_STEP64
#define FTK_QUAD_STEP64 1  // 0: 16-add blocks everywhere (A / B, scripts/build_variant.sh)
#endif
#define FTK_QADD4R(i, a, b, c, d) FTK_QADD(i, a) FTK_QADD(i, b) FTK_QADD(i, c) FTK_QADD(i, d)
#define FTK_QADD16R(a, b, c, d) FTK_QADD4R(0, a, b, c, d) FTK_QADD4R(1, a, b, c, d) FTK_QADD4R(2, a, b, c, d) FTK_QADD4R(3, a, b, c, d)
__device__ __forceinline__ float chain_quad_step64(float acc, const float4 q0, const float4 q1, const float4 q2, const float4 q3) {
    asm volatile("s_nop 1\n" FTK_QADD16R("%1", "%2", "%3", "%4") FTK_QADD16R("%5", "%6", "%7", "%8") FTK_QADD16R("%9", "%10", "%11", "%12")
                     FTK_QADD16R("%13", "%14", "%15", "%16")
                 : "+v"(acc)
                 : "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w), "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w),
                   "v"(q3.x), "v"(q3.y), "v"(q3.z), "v"(q3.w));
    return acc;
}

// One chunk row of up to 64 terms: `quad_row` = the row of THIS lane's quad's sum + 4 * (lane & 3) floats (16-byte aligned);
// `left` = terms that count (wave-uniform, >= 1; anything >= 64 is a full chunk); terms up to the next multiple of 16 are read and
// added, so they must be exact zeros (x + (+-0) == x for every value a sum can hold: the sums start at +0).  All four reads of the
// chunk are issued first; each 16-add block waits only for its own.
__device__ __forceinline__ float chain_quads_left(float acc, const float *quad_row, int left) {
    const float4 *t = reinterpret_cast<const float4 *>(quad_row);
    const float4 q0 = t[0];
    if (left > 48) {
        const float4 q1 = t[4], q2 = t[8], q3 = t[12];
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q0);  // (16-add blocks here: the reads were issued just now, and the first block can start on the
        acc = chain_quad_step16(acc, q1);  // first read; one 64-add block measured 1.6 - 2.3 % slower in the pipelined Basic kernel)
        acc = chain_quad_step16(acc, q2);
        return chain_quad_step16(acc, q3);
    }
    if (left > 32) {
        const float4 q1 = t[4], q2 = t[8];
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q0);
        acc = chain_quad_step16(acc, q1);
        return chain_quad_step16(acc, q2);
    }
    if (left > 16) {
        const float4 q1 = t[4];
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q0);
        return chain_quad_step16(acc, q1);
    }
    return chain_quad_step16(acc, q0);
}

// `count` FULL chunks (count >= 1, wave-uniform), `step4` float4 from one chunk's row to the next.  Four term registers as in
// chain_quads_left, but each is read again — for the NEXT chunk — as soon as its 16 adds are through, so a read has the other 48 adds
// to arrive: no LDS latency between chunks, no registers beyond chain_quads_left's (a second set of four cost the pipelined Basic
// kernel a wave per SIMD), one address per chunk.  (Behind the last chunk the reads repeat that chunk: valid memory, never used.)
__device__ __forceinline__ float chain_quads_chunks(float acc, const float4 *t, int count, int step4) {
    float4 q0 = t[0], q1 = t[4], q2 = t[8], q3 = t[12];
#pragma nounroll
    for (int w = 0; w < count; ++w) {
        t += w + 1 < count ? step4 : 0;
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q0);
        q0 = t[0];
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q1);
        q1 = t[4];
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q2);
        q2 = t[8];
        __builtin_amdgcn_sched_barrier(0);
        acc = chain_quad_step16(acc, q3);
        q3 = t[12];
    }
    return acc;
}

// `n16` 16-term steps (n16 >= 1, wave-uniform) from `t_lane` = this lane's first float4, consecutive steps `step4` float4 apart: two
// register pairs, the reads two steps ahead of the adds.
// kBlocks64 (the one-wave-per-feature kernels, where the chain wave is alone on its SIMD and every issued instruction counts: Basic /
// affine fast on the real pair -4 .. -5 %): 64-add blocks; otherwise (the multi-wave kernels: no consistent difference) 16-add blocks.
template <bool kBlocks64 = false>
__device__ __forceinline__ float chain_quads_strided(float acc, const float4 *t, int n16, int step4) {
    if constexpr (kBlocks64 && FTK_QUAD_STEP64) {
        // blocks of four steps (64 adds in one asm block), the reads of the next block issued before the adds of this one; the up to
        // three steps behind the last block are read up front and kept in registers (no LDS latency at the end of the chain)
        const int rem = n16 & 3, tail = n16 - rem;
        const float4 r0 = t[(rem > 0 ? tail : 0) * step4], r1 = t[(rem > 1 ? tail + 1 : 0) * step4], r2 = t[(rem > 2 ? tail + 2 : 0) * step4];
        if (tail > 0) {
            float4 a0 = t[0], a1 = t[step4], a2 = t[2 * step4], a3 = t[3 * step4], b0, b1, b2, b3;
            int j = 0;
#pragma nounroll
            for (;;) {
                const bool more_b = j + 8 <= tail;
                if (more_b) {
                    b0 = t[(j + 4) * step4];
                    b1 = t[(j + 5) * step4];
                    b2 = t[(j + 6) * step4];
                    b3 = t[(j + 7) * step4];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = chain_quad_step64(acc, a0, a1, a2, a3);
                j += 4;
                if (!more_b) {
                    break;
                }
                const bool more_a = j + 8 <= tail;
                if (more_a) {
                    a0 = t[(j + 4) * step4];
                    a1 = t[(j + 5) * step4];
                    a2 = t[(j + 6) * step4];
                    a3 = t[(j + 7) * step4];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc = chain_quad_step64(acc, b0, b1, b2, b3);
                j += 4;
                if (!more_a) {
                    break;
                }
            }
        }
        if (rem > 0) {
            acc = chain_quad_step16(acc, r0);
        }
        if (rem > 1) {
            acc = chain_quad_step16(acc, r1);
        }
        if (rem > 2) {
            acc = chain_quad_step16(acc, r2);
        }
        return acc;
    } else {
        float4 qa = t[0], qb = t[n16 > 1 ? step4 : 0], qc, qd;
        int j = 0;
#pragma nounroll
        for (; j + 4 <= n16; j += 4) {
            qc = t[(j + 2) * step4];
            qd = t[(j + 3) * step4];
            __builtin_amdgcn_sched_barrier(0);
            acc = chain_quad_step16(acc, qa);
            acc = chain_quad_step16(acc, qb);
            qa = t[(j + 4 < n16 ? j + 4 : 0) * step4];  // (wave-uniform selects; a read past the end is never made)
            qb = t[(j + 5 < n16 ? j + 5 : 0) * step4];
            __builtin_amdgcn_sched_barrier(0);
            acc = chain_quad_step16(acc, qc);
            acc = chain_quad_step16(acc, qd);
        }
        const int rem = n16 - j;  // 0 .. 3 steps left; qa / qb hold the first two
        if (rem > 2) {
            qc = t[(j + 2) * step4];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (rem > 0) {
            acc = chain_quad_step16(acc, qa);
        }
        if (rem > 1) {
            acc = chain_quad_step16(acc, qb);
        }
        if (rem > 2) {
            acc = chain_quad_step16(acc, qc);
        }
        return acc;
    }
}

// A whole row [n16 * 16 terms] of one sum: `quad_row` = the row + 4 * (lane & 3) floats.
template <bool kBlocks64 = false>
__device__ __forceinline__ float chain_quads_row(float acc, const float *quad_row, int n16) {
    return chain_quads_strided<kBlocks64>(acc, reinterpret_cast<const float4 *>(quad_row), n16, 4);
}

constexpr int kChunkPixels = 64;              // pixels per chunk of the chunked sweep / chain loops = one wave round
constexpr int kChunkRow = kChunkPixels + 4;   // ring row pitch in floats: the chain lanes' 16-byte reads of different rows hit different banks

__device__ __forceinline__ void pixel_rc(const KltParams &p, int pxi, int &prow, int &pcol) {
    if (p.magic_pc20 != 0) {  // wave-uniform; v_mul_u32_u24 is full rate, v_mul_hi_u32 a quarter of it
        prow = (int)(__umul24((unsigned)pxi, p.magic_pc20) >> 20);
    } else {
        prow = (p.patch_cols == 1) ? pxi : (int)__umulhi((unsigned)pxi, p.magic_pc);
    }
    pcol = pxi - imul(prow, p.patch_cols);
}

// ---------------------------------------------------------------------------------------------
// Window management
// ---------------------------------------------------------------------------------------------

// Top-left corner of the (2h+4)^2 footprint of a patch centred at (u, v): bilinear bases of the
// patch pixels and of their +-1 neighbours lie in [floor - h - 1, floor + h + 1], plus one for the
// +1 bilinear neighbour.
// (u, v) are the feature's coordinates: the same in every lane of the wave (a wave never serves two features), so the corner is
// handed on as a SCALAR — everything integer that follows from it (window tests, clamps, restaging decisions, address bases)
// then runs on the scalar unit instead of taking vector issue slots.
__device__ __forceinline__ void footprint_origin(const KltParams &p, float u, float v, int &r_lo, int &c_lo) {
    r_lo = __builtin_amdgcn_readfirstlane(wadd(f2i(floorf(v)), -(p.half_rows + 1)));
    c_lo = __builtin_amdgcn_readfirstlane(wadd(f2i(floorf(u)), -(p.half_cols + 1)));
}

// The per-iteration form of "does the current window still cover the patch footprint" as four float compares: with
// need = floor(x) - (half + 1) the integer test need >= lo && need + 2 half + 4 <= lo + extent (+ 1 for columns, whose last pair
// reaches one pixel further) is floor(x) in [lo + half + 1, lo + extent - half - 3 (+ 1)] — exact in fp32 while the bounds are small
// integers; for a window whose origin is not (a feature far outside the image) the interval is empty and the integer test decides.
__device__ __forceinline__ void win_set_cover(const KltParams &p, Win &w) {
    const bool small = (unsigned)(w.r_lo + (1 << 22)) < (1u << 23) && (unsigned)(w.c_lo + (1 << 22)) < (1u << 23);
    w.cover[0] = small ? (float)(w.r_lo + p.half_rows + 1) : 1.0f;
    w.cover[1] = small ? (float)(w.r_lo + w.rows - p.half_rows - 3) : 0.0f;
    w.cover[2] = small ? (float)(w.c_lo + p.half_cols + 1) : 1.0f;
    w.cover[3] = small ? (float)(w.c_lo + w.cols - p.half_cols - 2) : 0.0f;
}

__device__ __forceinline__ bool win_covers(const Win &w, float u, float v) {
    const float fu = floorf(u), fv = floorf(v);
    return fv >= w.cover[0] && fv <= w.cover[1] && fu >= w.cover[2] && fu <= w.cover[3];  // NaN and huge coordinates fail
}

// One pixel-pair element of a window: (img[clamp(r)][clamp(c)], img[clamp(r)][clamp(c + 1)]).
__device__ __forceinline__ uint16_t window_element(const DevImage &im, int r_lo, int c_lo, int idx, int wcols, uint32_t magic_cols) {
    const int r = (int)__umulhi((unsigned)idx, magic_cols);
    const int c = idx - imul(r, wcols);
    const int ir = clampi(wadd(r_lo, r), 0, im.rows - 1);
    const int ic = wadd(c_lo, c);
    const int ic0 = clampi(ic, 0, im.cols - 1);
    const int ic1 = clampi(wadd(ic, 1), 0, im.cols - 1);
    const unsigned row_off = __umul24((unsigned)ir, (unsigned)im.cols);  // 32-bit offsets, as in px()
    return (uint16_t)((unsigned)im.data[row_off + (unsigned)ic0] | ((unsigned)im.data[row_off + (unsigned)ic1] << 8));
}

// Loads up to kStageBatch window elements per thread with every global load in flight before the
// first LDS store (the loop body is branch-free: out-of-range slots read element 0 and are not stored).
constexpr int kStageBatch = 4;

__device__ __forceinline__ void stage_elements(const Blk &b, const DevImage &im, uint16_t *dst, int r_lo, int c_lo, int wcols, uint32_t magic_cols,
                                               int total) {
    for (int base = 0; base < total; base += b.nt * kStageBatch) {
        uint16_t v[kStageBatch];
#pragma unroll
        for (int k = 0; k < kStageBatch; ++k) {
            const int idx = base + k * b.nt + b.tid;
            v[k] = window_element(im, r_lo, c_lo, idx < total ? idx : 0, wcols, magic_cols);
        }
#pragma unroll
        for (int k = 0; k < kStageBatch; ++k) {
            const int idx = base + k * b.nt + b.tid;
            if (idx < total) {
                dst[idx] = v[k];
            }
        }
    }
}

// Fast staging for windows that lie completely inside the image: thread (row, quad) fetches 8
// consecutive bytes with one unaligned global_load_dwordx2, forms the four pixel pairs
// (b0,b1) (b1,b2) (b2,b3) (b3,b4) with v_alignbyte / v_alignbit and stores them with one
// ds_write_b64 (wcols is a multiple of 4).  ~1/4 of the instructions of the per-element path.
__device__ __forceinline__ bool window_inside(const DevImage &im, int r_lo, int c_lo, int wrows, int wcols) {
    return r_lo >= 0 && c_lo >= 0 && (long long)r_lo + wrows <= im.rows && (long long)c_lo + wcols + 4 <= im.cols;
}

__device__ __forceinline__ void stage_rows_inside(const Blk &b, const DevImage &im, uint16_t *dst, int r_lo, int c_lo, int wrows, int wcols,
                                                  uint32_t magic_quads) {
    const int quads = wcols >> 2;
    const int total = wrows * quads;
    for (int idx = b.tid; idx < total; idx += b.nt) {
        const int r = (quads == 1) ? idx : (int)__umulhi((unsigned)idx, magic_quads);
        const int q = idx - imul(r, quads);
        const uint8_t *src = im.data + (__umul24((unsigned)(r_lo + r), (unsigned)im.cols) + (unsigned)(c_lo + 4 * q));
        uint32_t x, y;
        __builtin_memcpy(&x, src, 4);
        __builtin_memcpy(&y, src + 4, 4);
        const uint32_t p0 = x & 0xFFFFu;
        const uint32_t p1 = (x >> 8) & 0xFFFFu;
        const uint32_t p2 = x >> 16;
        const uint32_t p3 = __builtin_amdgcn_alignbyte(y, x, 3) & 0xFFFFu;
        *reinterpret_cast<uint2 *>(dst + imul(r, wcols) + 4 * q) = make_uint2(p0 | (p1 << 16), p2 | (p3 << 16));
    }
}

__device__ __forceinline__ void stage_any(const Blk &b, const DevImage &im, uint16_t *dst, int r_lo, int c_lo, int wrows, int wcols,
                                          uint32_t magic_cols, uint32_t magic_quads) {
    if (window_inside(im, r_lo, c_lo, wrows, wcols)) {
        stage_rows_inside(b, im, dst, r_lo, c_lo, wrows, wcols, magic_quads);
    } else {
        stage_elements(b, im, dst, r_lo, c_lo, wcols, magic_cols, wrows * wcols);
    }
}

// One quad of a window that lies inside the image: 8 bytes -> four pixel pairs (see stage_rows_inside).
__device__ __forceinline__ uint2 load_quad_pairs(const DevImage &im, int r_lo, int c_lo, int idx, int quads, uint32_t magic_quads, int &lds_off,
                                                 int wcols) {
    const int r = (quads == 1) ? idx : (int)__umulhi((unsigned)idx, magic_quads);
    const int q = idx - imul(r, quads);
    const uint8_t *src = im.data + (__umul24((unsigned)(r_lo + r), (unsigned)im.cols) + (unsigned)(c_lo + 4 * q));
    uint32_t x, y;
    __builtin_memcpy(&x, src, 4);
    __builtin_memcpy(&y, src + 4, 4);
    const uint32_t p0 = x & 0xFFFFu;
    const uint32_t p1 = (x >> 8) & 0xFFFFu;
    const uint32_t p2 = x >> 16;
    const uint32_t p3 = __builtin_amdgcn_alignbyte(y, x, 3) & 0xFFFFu;
    lds_off = imul(r, wcols) + 4 * q;
    return make_uint2(p0 | (p1 << 16), p2 | (p3 << 16));
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
// Window prefetch: raw 8-byte loads issued early, turned into pixel pairs and stored late.
// ---------------------------------------------------------------------------------------------
template <int N>
struct RawQuads {
    uint32_t x[N], y[N];
};

template <int N>
__device__ __forceinline__ void issue_quads(RawQuads<N> &q, const Blk &b, const DevImage &im, int r_lo, int c_lo, int wrows, int wcols,
                                            uint32_t magic_quads) {
    const int quads = wcols >> 2;
    const int total = wrows * quads;
    const int tid = opaque(b.tid);
    const uint8_t *base = im.data + (long long)r_lo * im.cols + c_lo;  // wave-uniform: scalar base + 32-bit lane offset
#pragma unroll
    for (int k = 0; k < N; ++k) {
        int idx = tid + k * b.nt;
        idx = idx < total ? idx : 0;
        const int r = (quads == 1) ? idx : (int)__umulhi((unsigned)idx, magic_quads);
        const int qq = idx - imul(r, quads);
        const uint8_t *src = base + (size_t)(unsigned)(r * im.cols + 4 * qq);
        __builtin_memcpy(&q.x[k], src, 4);
        __builtin_memcpy(&q.y[k], src + 4, 4);
    }
}

template <int N>
__device__ __forceinline__ void store_quads(const RawQuads<N> &q, const Blk &b, uint16_t *dst, int wrows, int wcols, uint32_t magic_quads) {
    const int quads = wcols >> 2;
    const int total = wrows * quads;
    const int tid = opaque(b.tid);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int idx = tid + k * b.nt;
        if (idx < total) {
            const int r = (quads == 1) ? idx : (int)__umulhi((unsigned)idx, magic_quads);
            const int qq = idx - imul(r, quads);
            const uint32_t x = q.x[k], y = q.y[k];
            const uint32_t p0 = x & 0xFFFFu;
            const uint32_t p1 = (x >> 8) & 0xFFFFu;
            const uint32_t p2 = x >> 16;
            const uint32_t p3 = __builtin_amdgcn_alignbyte(y, x, 3) & 0xFFFFu;
            *reinterpret_cast<uint2 *>(dst + imul(r, wcols) + 4 * qq) = make_uint2(p0 | (p1 << 16), p2 | (p3 << 16));
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Position-keyed slot swaps.  The launch order above is a permutation of LIST INDICES made during an earlier call: a caller whose
// feature list changes between frames (features re-detected, reshuffled) gets an arbitrary order, and a feature that needs dozens
// of iterations starts wherever it happens to sit — the launch then ends that much later (config 3: 146 us with a fitting order,
// 208 with a stale one).  Iteration counts are therefore ALSO remembered by POSITION: every tracked feature leaves
// (call number, count) in a hash table keyed by its level-0 position (atomicMax: the newest call wins, then the largest count;
// one table is written by this call while the other, last call's, is only read).  At the start of a launch the first kSchedHeadSlots slots — resident from the first microsecond — each look through
// their share of the LATE slots (>= kSchedLateSlot: the ones that may have to wait for a free CU), pick the one whose position
// predicts the most iterations, and, if that is kSchedSwapMargin more than their own feature's prediction, trade places with
// it: the head runs the long feature now, the late slot runs the head's feature when its turn comes.  A claim word per late
// slot, taken with a compare-and-swap by whichever of the two gets there first, guarantees that every feature is processed
// exactly once whatever the timing; nobody waits for anybody.  Which slot runs a feature changes nothing in its arithmetic.
// ---------------------------------------------------------------------------------------------
constexpr int kSchedTableBits = 16;
constexpr int kSchedTableSize = 1 << kSchedTableBits;  // entries per table; two tables (written by this call / read from the last)
constexpr int kSchedHeadFirst = 256;   // the heads are the slots [kSchedHeadFirst, kSchedHeadFirst + kSchedHeadSlots): resident from the
constexpr int kSchedHeadSlots = 256;   // first microsecond, but NOT the very first ones — with a fitting launch order those hold the
constexpr int kSchedLateSlot = 1024;   // longest features, which must not start a scan's 2 - 3 us later
constexpr uint32_t kSchedLongCount = 12;   // a late slot is a candidate from this predicted count on ...
constexpr uint32_t kSchedSwapMargin = 8;   // ... and a head trades with it if that is this much above its own feature's prediction
constexpr uint32_t kSchedSelf = 0x1FFu;    // claim code "the slot runs its own feature"; 0 .. kSchedHeadSlots - 1: the head (by number) it trades with

// positions are remembered at 4-pixel resolution (a feature that crosses such a boundary between two frames is simply not predicted)
__device__ __forceinline__ uint32_t sched_table_slot(float u, float v) {
    const uint32_t qx = (uint32_t)(int)fminf(fmaxf(u * 0.25f, 0.0f), 1048575.0f), qy = (uint32_t)(int)fminf(fmaxf(v * 0.25f, 0.0f), 1048575.0f);  // NaN -> 0
    return ((qx * 73856093u) ^ (qy * 19349663u)) & (uint32_t)(kSchedTableSize - 1);
}

// What the LAST call left at this position (the table this call only reads: every wave of a launch sees the same predictions)
__device__ __forceinline__ uint32_t sched_prediction(const KltParams &p, float u, float v) {
    const uint32_t word = p.sched_grid[(((p.sched_call - 1u) & 1u) << kSchedTableBits) + sched_table_slot(u, v)];
    return (word >> 8) == ((p.sched_call - 1u) & 0xFFFFFFu) ? (word & 0xFFu) : 0u;
}

// ... at the feature's reference position (a caller that tracks the same list again asks there) AND at the position it was tracked to
// (a caller that tracks frame after frame asks there: the next call's reference positions are this call's results)
__device__ __forceinline__ void sched_grid_record(const KltParams &p, float ref_u, float ref_v, float out_u, float out_v, uint32_t iters) {
    if (p.sched_grid != nullptr) {
        const uint32_t word = ((p.sched_call & 0xFFFFFFu) << 8) | (iters < 255u ? iters : 255u);
        uint32_t *table = p.sched_grid + ((p.sched_call & 1u) << kSchedTableBits);
        const uint32_t at_ref = sched_table_slot(ref_u, ref_v), at_out = sched_table_slot(out_u, out_v);
        atomicMax(&table[at_ref], word);
        if (at_out != at_ref) {
            atomicMax(&table[at_out], word);
        }
    }
}

// The iteration count of a call's longest feature, for the NEXT call's wave policy (ftk_api.cpp "tail-aware"): features that ran at
// least kTailReportFrom iterations raise a device word with atomicMax — an L2 load for most, an atomic for the few that raise it —
// and whoever raised it forwards the new value to a device-visible host word with one system-scope store.  A lower value may land
// after a higher one (two raisers racing over PCIe): the host treats the word as a hint.  One lane per feature calls this.
constexpr uint32_t kTailReportFrom = 12;
__device__ __forceinline__ void tail_report(const KltParams &p, uint32_t iters, uint32_t id) {
    if (p.tail_dev != nullptr && (iters >= kTailReportFrom || id == 0u)) {  // (feature 0 always: every launch refreshes the variant's word)
        const uint32_t word = (p.tail_call << 8) | (iters < 255u ? iters : 255u);
        if (word > __hip_atomic_load(p.tail_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (atomicMax(p.tail_dev, word) < word) {
                __hip_atomic_store(p.tail_host, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Called by ONE wave on behalf of a launch slot (every lane with the same arguments): the list slot whose feature this launch
// slot runs — `slot` itself, or the slot it traded places with.  `swapped_in` tells a head that it now runs a predicted-long feature.
__device__ __forceinline__ uint32_t sched_resolve_slot(const KltParams &p, uint32_t slot, bool &swapped_in) {
    swapped_in = false;
    const uint32_t n = (uint32_t)p.n, call = p.sched_call & 0x7FFFFFu;
    const int lane = (int)(threadIdx.x & 63);
    // The sort block of the LAST launch found no tail in the counts it sorted: nobody trades, and nobody pays for looking (one
    // word, the same for every slot of this launch: this launch's own sort block writes the other one).
    if (p.sched_flags[(p.sched_call - 1u) & 1u] == ((((p.sched_call - 1u) & 0x7FFFFFFFu) << 1) | 1u)) {
        return slot;
    }
    if (slot >= (uint32_t)kSchedLateSlot) {
        // A late slot.  Heads only ever claim slots whose prediction reaches kSchedLongCount, and predictions come from a table
        // nobody writes during this launch: a slot below that knows, without asking, that it runs its own feature.
        const uint32_t f = p.order ? (uint32_t)p.order[slot] : slot;
        if (sched_prediction(p, p.ref_uv[2 * f], p.ref_uv[2 * f + 1]) < kSchedLongCount) {
            return slot;
        }
        uint32_t word = 0;
        if (lane == 0) {
            uint32_t seen = __hip_atomic_load(&p.sched_claim[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((seen >> 9) != call) {
                const uint32_t mine = (call << 9) | kSchedSelf;
                const uint32_t prev = atomicCAS(&p.sched_claim[slot], seen, mine);
                seen = prev == seen ? mine : prev;  // lost the race: only a head of THIS call writes here
            }
            word = seen;
        }
        word = (uint32_t)__builtin_amdgcn_readfirstlane((int)word);
        const uint32_t code = word & 0x1FFu;
        return ((word >> 9) == call && code != kSchedSelf) ? (uint32_t)kSchedHeadFirst + code : slot;
    }
    if (slot < (uint32_t)kSchedHeadFirst || slot >= (uint32_t)(kSchedHeadFirst + kSchedHeadSlots) || n <= (uint32_t)kSchedLateSlot) {
        return slot;
    }
    // a head: my share of the late slots, one per lane and pass
    const uint32_t head = slot - (uint32_t)kSchedHeadFirst;
    const uint32_t late = n - (uint32_t)kSchedLateSlot, share = (late + kSchedHeadSlots - 1) / kSchedHeadSlots;
    const uint32_t first = (uint32_t)kSchedLateSlot + head * share, last = min(first + share, n);
    const uint32_t own_feature = p.order ? (uint32_t)p.order[slot] : slot;
    const uint32_t own = sched_prediction(p, p.ref_uv[2 * own_feature], p.ref_uv[2 * own_feature + 1]);
    uint32_t best = 0;  // (prediction << 20) | slot: the most iterations, then the highest slot
    for (uint32_t s = first + (uint32_t)lane; s < last; s += 64u) {
        const uint32_t f = p.order ? (uint32_t)p.order[s] : s;
        best = max(best, (sched_prediction(p, p.ref_uv[2 * f], p.ref_uv[2 * f + 1]) << 20) | s);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        best = max(best, (uint32_t)__shfl_xor((int)best, off));
    }
    const uint32_t pred = best >> 20, target = best & 0xFFFFFu;
    if (pred < kSchedLongCount || pred < own + kSchedSwapMargin || target < first) {
        return slot;
    }
    uint32_t won = 0;
    if (lane == 0) {
        const uint32_t seen = __hip_atomic_load(&p.sched_claim[target], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((seen >> 9) != call) {
            won = atomicCAS(&p.sched_claim[target], seen, (call << 9) | head) == seen ? 1u : 0u;
        }
    }
    won = (uint32_t)__builtin_amdgcn_readfirstlane((int)won);
    swapped_in = won != 0u;
    return won != 0u ? target : slot;
}

// ---------------------------------------------------------------------------------------------
// Launch order of a later call, computed by ONE extra workgroup of the tracker launch itself (block 0: it starts first and
// runs beside the feature workgroups, so the sort costs no launch and no time of its own): the feature indices sorted by the
// iteration counts of the PREVIOUS call, longest first — a counting sort over min(count, 255) with the workgroup's dynamic LDS
// as its bins.  When the counts have no tail (largest <= 1.5 x the mean of the tracked features) the same counting sort runs over
// image tiles instead (klt_order_block).  Any permutation yields the same tracking results; only the schedule differs.
// ---------------------------------------------------------------------------------------------

// Counting only (nothing returns, so nothing waits): ONE LDS atomic for all the lanes that share the lowest active lane's bin, one
// each for the rest.
__device__ __forceinline__ void wave_bin_count(int *bins, int bin, bool active) {
    const unsigned long long todo = wave_ballot(active);
    if (todo == 0ull) {
        return;
    }
    const int leader = __builtin_ctzll(todo);
    const int leader_bin = __builtin_amdgcn_readlane(bin, leader);
    const bool with_leader = active && bin == leader_bin;
    const unsigned long long same = wave_ballot(with_leader);
    if ((int)(threadIdx.x & 63) == leader) {
        (void)__hip_atomic_fetch_add(&bins[leader_bin], __popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (active && !with_leader) {
        (void)__hip_atomic_fetch_add(&bins[bin], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// A slot in its bin for every active lane, for U independent elements per lane.  Per element one round of "the lowest active
// lane's bin: ONE LDS atomic for all the lanes that share it" — iteration counts cluster, and a call whose features all took 5
// iterations would otherwise serialise thousands of atomics on one address — then one atomic per remaining lane (counts spread
// over many bins: little contention).  The U leader atomics are issued back to back, then the U atomics of the remaining lanes:
// two waits instead of 2 U.  Slots are unique (LDS atomics of a wave execute in order; other waves interleave atomically); which
// of two equal-bin features gets the earlier slot is not defined.
template <int U>
__device__ __forceinline__ void wave_bin_claim_batch(int *bins, const int (&bin)[U], const bool (&active)[U], int (&slot)[U]) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned long long below = (1ull << lane) - 1ull;
    int leader[U], leader_bin[U], base[U];
    unsigned long long same[U];
    bool with_leader[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned long long todo = wave_ballot(active[u]);
        leader[u] = todo != 0ull ? __builtin_ctzll(todo) : 0;
        leader_bin[u] = __builtin_amdgcn_readlane(bin[u], leader[u]);
        with_leader[u] = active[u] && bin[u] == leader_bin[u];
        same[u] = wave_ballot(with_leader[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        base[u] = 0;
        if (same[u] != 0ull && lane == leader[u]) {
            base[u] = atomicAdd(&bins[leader_bin[u]], __popcll(same[u]));
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        base[u] = __builtin_amdgcn_readlane(base[u], leader[u]);
        slot[u] = with_leader[u] ? base[u] + __popcll(same[u] & below) : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (active[u] && !with_leader[u]) {
            slot[u] = atomicAdd(&bins[bin[u]], 1);
        }
    }
}

constexpr int kOrderLdsBytes = (4 * 256 + 4) * 4;  // two histograms (iteration counts, image tiles) with their scans + a flag

// Rank in tile order -> launch slot.  Workgroups go to the eight XCDs round robin (workgroup w -> XCD w mod 8, each with its own
// L2), so runs of kOrderRunGroups consecutive workgroup ranks (one small image region) are dealt to ONE XCD, run after run round
// robin: neighbours in the image share an L2 and a moment in time, and every XCD still gets a mix of all regions (whole eighths
// of the image per XCD fetch least — 5.5 MB instead of 32.8 MB per 25 000-feature launch at 1080p — but run 6 % SLOWER than list
// order; runs of 4 / 16 / 32 / 64 / 128 / 256 workgroups: +2.5 / +1.3 / -0.9 / -1.6 / -1.6 / -1.3 % with 21.8 / 15.8 / - / 9.8 MB
// fetched).  Only whole blocks of 8 runs take part; the ranks behind them stay put.
#ifndef FTK_ORDER_RUN_GROUPS
#define FTK_ORDER_RUN_GROUPS 64
#endif
constexpr int kOrderRunGroups = FTK_ORDER_RUN_GROUPS;

__device__ __forceinline__ int xcd_major_slot(int rank, int n, int group) {
    const int whole = n / group;  // workgroups with `group` features
    const int dealt = whole / (8 * kOrderRunGroups) * (8 * kOrderRunGroups);
    const int m = rank / group;  // workgroup rank
    if (m >= dealt) {
        return rank;
    }
    const int sub = rank - m * group;
    const int run = m / kOrderRunGroups, in_run = m - run * kOrderRunGroups;
    const int xcd = run & 7, j = (run >> 3) * kOrderRunGroups + in_run;  // the j-th workgroup of that XCD
    return (xcd + 8 * j) * group + sub;
}

// 16 x 16 tiles of the level-0 image, row by row: 256 bins.  (A run of the launch order — kOrderRunGroups workgroups — holds two
// or three tiles' features, so a squarer Morton walk over the tiles would buy nothing, and this key is three instructions.)
__device__ __forceinline__ int image_tile(float u, float v, float inv_tile_u, float inv_tile_v) {
    const int tx = (int)fminf(fmaxf(u * inv_tile_u, 0.0f), 15.0f), ty = (int)fminf(fmaxf(v * inv_tile_v, 0.0f), 15.0f);  // NaN -> 0
    return ty * 16 + tx;
}

// exclusive scan of the 256 bin counts into bin_start by one wave, four bins per lane
__device__ __forceinline__ void order_scan_bins(const int *bin_count, int *bin_start, int l) {
    const int c0 = bin_count[4 * l], c1 = bin_count[4 * l + 1], c2 = bin_count[4 * l + 2], c3 = bin_count[4 * l + 3];
    int run = c0 + c1 + c2 + c3;
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
}

// The launch order of a later call (order[slot] = feature).  Iteration counts with a tail: longest first.  Without one (every
// feature takes about as long): in space — features of one image region next to each other AND on one XCD, so that the window
// loads of a level entry find their lines in that XCD's L2 instead of every L2 fetching the whole pyramid (`ref_uv`: this call's
// reference pixels at level 0, `cols` x `rows` that level; `group`: features per workgroup of the launch the order is for).
//
// ONE workgroup does this beside the feature workgroups of a launch, so it must not outlast them: two passes over the list, both
// histograms (256 iteration bins, 256 image tiles) counted in the first, and the global loads of kOrderBatch elements per thread
// in flight together (one load per trip had made each pass a chain of ~100 dependent L2 round trips: 137 us for 25 000 features
// and 1.16 ms for 200 000 — longer than the 200 000-feature launch itself).
constexpr int kOrderBatch = 8;

__device__ __forceinline__ void klt_order_block(const uint32_t *iters, int32_t *order, int n, int *lds, const float *ref_uv, int cols, int rows, int group,
                                                uint32_t *sched_flags, uint32_t sched_call) {
    int *bin_count = lds, *bin_start = lds + 256, *tile_count = lds + 512, *tile_start = lds + 768, *flat = lds + 1024;
    const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
    const bool spatial = ref_uv != nullptr && cols >= 16 && rows >= 16;
    const float inv_tile_u = 16.0f / (float)(cols > 0 ? cols : 1), inv_tile_v = 16.0f / (float)(rows > 0 ? rows : 1);
    const float2 *uv2 = reinterpret_cast<const float2 *>(ref_uv);
    for (int k = tid; k < 256; k += nt) {
        bin_count[k] = 0;
        tile_count[k] = 0;
    }
    __syncthreads();
    for (int base = 0; base < n; base += nt * kOrderBatch) {
        uint32_t it[kOrderBatch];
        float2 uv[kOrderBatch];
#pragma unroll
        for (int u = 0; u < kOrderBatch; ++u) {
            // unconditional loads of a clamped index through 32-bit byte offsets from the uniform bases: two instructions of address
            const uint32_t i = (uint32_t)min(base + u * nt + tid, n - 1);
            it[u] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(iters) + i * 4u);
            uv[u] = spatial ? *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(uv2) + i * 8u) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < kOrderBatch; ++u) {
            const bool active = base + u * nt + tid < n;
            wave_bin_count(bin_count, active ? 255 - (int)min(it[u], 255u) : 0, active);
            if (spatial && active) {
                // tile keys are spread over the bins: one plain LDS atomic per lane, nothing returned
                (void)__hip_atomic_fetch_add(&tile_count[image_tile(uv[u].x, uv[u].y, inv_tile_u, inv_tile_v)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    if (tid < 64) {
        // exclusive scans of both histograms by one wave: four bins per lane; and the no-tail test on the iteration counts
        const int l = tid;
        const int c0 = bin_count[4 * l], c1 = bin_count[4 * l + 1], c2 = bin_count[4 * l + 2], c3 = bin_count[4 * l + 3];
        // bin k holds count 255 - k: weighted sum and largest count over the tracked features (count > 0: bins 0..254)
        long long weighted = (long long)c0 * (255 - 4 * l) + (long long)c1 * (254 - 4 * l) + (long long)c2 * (253 - 4 * l) + (long long)c3 * (252 - 4 * l);
        int tracked = c0 + c1 + c2 + c3 - (l == 63 ? c3 : 0);
        int largest = c0 ? 255 - 4 * l : (c1 ? 254 - 4 * l : (c2 ? 253 - 4 * l : (c3 ? 252 - 4 * l : 0)));
        for (int off = 32; off >= 1; off >>= 1) {
            weighted += __shfl_xor(weighted, off);
            tracked += __shfl_xor(tracked, off);
            largest = max(largest, __shfl_xor(largest, off));
        }
        order_scan_bins(bin_count, bin_start, l);
        order_scan_bins(tile_count, tile_start, l);
        if (l == 0) {
            *flat = (tracked == 0 || 2ll * largest * tracked <= 3ll * weighted) ? 1 : 0;  // largest <= 1.5 x mean
        }
    }
    __syncthreads();
    if (tid == 0 && sched_flags != nullptr) {
        sched_flags[sched_call & 1u] = ((sched_call & 0x7FFFFFFFu) << 1) | (*flat != 0 ? 1u : 0u);  // for the NEXT launch's slots (sched_resolve_slot)
    }
    const bool by_tile = *flat != 0 && spatial;
    if (*flat != 0 && !spatial) {
        for (int i = tid; i < n; i += nt) {
            order[i] = i;
        }
        return;
    }
    for (int base = 0; base < n; base += nt * kOrderBatch) {
        uint32_t it[kOrderBatch];
        float2 uv[kOrderBatch];
#pragma unroll
        for (int u = 0; u < kOrderBatch; ++u) {
            const uint32_t i = (uint32_t)min(base + u * nt + tid, n - 1);
            it[u] = by_tile ? 0u : *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(iters) + i * 4u);
            uv[u] = by_tile ? *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(uv2) + i * 8u) : make_float2(0.0f, 0.0f);
        }
        int bin[kOrderBatch], slot[kOrderBatch];
        bool active[kOrderBatch];
#pragma unroll
        for (int u = 0; u < kOrderBatch; ++u) {
            active[u] = base + u * nt + tid < n;
            bin[u] = !active[u] ? 0 : (by_tile ? image_tile(uv[u].x, uv[u].y, inv_tile_u, inv_tile_v) : 255 - (int)min(it[u], 255u));
        }
        if (by_tile) {
#pragma unroll
            for (int u = 0; u < kOrderBatch; ++u) {
                slot[u] = active[u] ? atomicAdd(&tile_start[bin[u]], 1) : 0;  // all in flight before the first is used
            }
        } else {
            wave_bin_claim_batch<kOrderBatch>(bin_start, bin, active, slot);
        }
#pragma unroll
        for (int u = 0; u < kOrderBatch; ++u) {
            if (active[u]) {
                order[by_tile ? xcd_major_slot(slot[u], n, group) : slot[u]] = base + u * nt + tid;
            }
        }
    }
}

}  // namespace ftk
