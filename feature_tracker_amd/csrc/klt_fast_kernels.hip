// klt_fast_kernels.hip — the `fast` method of the trackers (the reference's DEFAULT: OpticalFlowOptions::kMethod = kFast,
// optical_flow.h:27), one wavefront per feature, no workgroup barrier anywhere.
//
// What sets the fast variants apart (basic_klt_fast.cpp:7-195): everything on the reference side of a level — the extended patch
// with ONE shared bilinear weight set (optical_flow.cpp:49-102), the central differences dx / dy, the Hessian — depends on the
// reference position alone, and an iteration samples the current image on an INTEGER lattice with one weight set and forms two
// sums.  With the test scenes' 1 - 2 iterations per level a feature's life is its four LEVEL ENTRIES, not its iterations, and in the
// generic kernel (klt_kernels.hip, 2 - 4 waves per feature) a level entry is five barriers, two dependent global round trips and a
// three-lane Hessian chain before the first iteration starts.  Here:
//
//   * one wave owns the feature: LDS operations of one wave execute in program order, so nothing ever waits at a barrier;
//   * the windows of the NEXT level (reference footprint; current footprint around twice the position at this level's entry) are
//     requested at this level's entry and arrive while it iterates — the current one is used when it still covers the patch at the
//     position the level ended with, otherwise staged again;
//   * the Hessian's three sums ride on the first iteration's chain: lanes 0 / 1 add the bias products, lanes 2 - 4 the Hessian
//     products, in one pass of the same dependent adds (the sums are formed in the reference's row-major order either way);
//   * the per-pixel record {dx, dy, reference value, usable} is one 16-byte LDS read per pixel and iteration.
//
// The affine tracker's fast method (affine_klt_fast.cpp:7-188) runs on the same skeleton: the level entry is the same extended patch;
// an iteration samples the current image at the WARPED patch positions (per-pixel bilinear weights, LDS window with a global-memory
// path for taps that left it), forms six bias products per pixel, and the 18 Hessian products (anchored at the level-entry position,
// :95-96) ride on the first iteration.  Its 24 sums do not fit a feature's LDS share as whole rows (24 x P floats), so sweep and
// chain alternate over 64-pixel CHUNKS through a one-slot ring [24][64 + 4] — the same products in the same order on the same
// lanes; the 6 x 6 LDLT is factored once per level on lanes 0 - 5 (ldlt6_factor_diag) and every iteration runs the substitutions.
//
// Arithmetic contract as in klt_kernels.hip: IEEE fp32, no contraction, correctly rounded division, every sum strictly in
// row-major pixel order on one lane.  Results are bit-identical to the generic kernel and to the oracle (tests/test_klt_gpu.py).
#define FTK_CHAIN_ROUND 4
#ifndef FTK_FK_QUAD_CHAIN
#define FTK_FK_QUAD_CHAIN 1  // the whole-row exact-order chains through the DPP network (klt_common.h "quad chain"); 0: one lane per sum (round 4)
#endif
#include "klt_common.h"

#include <stdlib.h>

namespace ftk {
namespace {

// An element of the extended patch is a convex combination of pixel values (weights from fractions in [0, 1)): never negative.
// A negative value therefore marks "outside the image" — validity travels WITH the value, one LDS read instead of two dependent ones.
constexpr float kFkInvalid = -1.0f;
constexpr int kFkTerms = 5;  // rows of `terms`: 0 -(dx * dt), 1 -(dy * dt) (every iteration); 2 dx * dx, 3 dx * dy, 4 dy * dy (once per level)

__host__ __device__ inline int fk_pad4(int x) { return (x + 3) & ~3; }
// Row pitch of `terms` in floats: congruent 4 mod 8, so that the 16-byte reads of the five chain lanes (one row each, same column)
// fall on five different groups of four banks
// (round 5, quad chains: the four lanes of a quad read 64 consecutive bytes of ONE row and the two quads of an 8-lane group two
// different rows — a pitch congruent 16 mod 32 floats puts those on the two halves of the banks)
__host__ __device__ inline int fk_term_pitch(const KltParams &p) {
#if FTK_FK_QUAD_CHAIN
    return (p.Ppad & 31) == 16 ? p.Ppad : p.Ppad + 16;
#else
    return (p.Ppad & 7) == 4 ? p.Ppad : p.Ppad + 4;
#endif
}

constexpr int kAfRows = 24;     // rows of the affine ring: 0 - 5 the bias products (every iteration), 6 + A_* the 18 Hessian products (first iteration)
constexpr int kAfSumFloats = 56;  // [0, 36) the dense 6 x 6 Hessian the factorisation reads, [40, 46) the bias sums, [48, 54) the solution

struct FkLds {
    float4 *rec;        // [Ppad] {dx, dy, extended patch at the pixel, flags (int bits; Basic: -1 / 0 = usable; affine: bit 0 usable, bit 1 has a gradient)}
    float *terms;       // Basic: [kFkTerms][pitch]; affine: the ring [kAfRows][kChunkRow]
    float *sums;        // affine only: kAfSumFloats
    float *ex;          // [Epad] extended reference patch (level entry only); an element outside the image holds kFkInvalid
    uint16_t *ref_win;  // rwin_rows x rwin_cols pixel pairs
    uint16_t *cur_win;  // cwin_rows x cwin_cols pixel pairs
};

__host__ __device__ inline size_t fk_terms_floats(int model, const KltParams &p) {
    // + what the chain's prefetch may read past the last row
    if (model == FTK_MODEL_AFFINE) {
        // the ring of the first iteration, or the six whole bias rows of the later ones (larger from 17 x 17 on), in the same space
        const size_t ring = (size_t)kAfRows * kChunkRow, rows = 6 * (size_t)fk_term_pitch(p);
        return (ring > rows ? ring : rows) + 16 * FTK_CHAIN_ROUND + kAfSumFloats;
    }
    return (size_t)kFkTerms * fk_term_pitch(p) + 16 * FTK_CHAIN_ROUND;
}

__host__ __device__ inline size_t fk_lds_bytes(int model, const KltParams &p) {
    size_t bytes = 16 * (size_t)p.Ppad;
    bytes += 4 * fk_terms_floats(model, p);
    bytes += 4 * (size_t)(p.ex_rows * 4 * ((p.ex_cols + 3) >> 2));  // the extended patch, rows of whole quads
    bytes += 2 * (size_t)fk_pad4(p.rwin_rows * p.rwin_cols);
    bytes += 2 * (size_t)fk_pad4(p.cwin_rows * p.cwin_cols);
    return (bytes + 15) & ~(size_t)15;
}

template <int MODEL>
__device__ __forceinline__ FkLds fk_carve(float4 *base, const KltParams &p) {
    FkLds c;
    c.rec = base;
    c.terms = reinterpret_cast<float *>(c.rec + p.Ppad);
    c.sums = c.terms + fk_terms_floats(FTK_MODEL_AFFINE, p) - kAfSumFloats;  // (behind the rows' prefetch slack; affine only)
    c.ex = c.terms + fk_terms_floats(MODEL, p);
    c.ref_win = reinterpret_cast<uint16_t *>(c.ex + p.ex_rows * 4 * ((p.ex_cols + 3) >> 2));
    c.cur_win = c.ref_win + fk_pad4(p.rwin_rows * p.rwin_cols);
    return c;
}

// Wave-local ordering point: the LDS operations of one wave execute in program order; this only keeps the compiler from moving
// them across.
__device__ __forceinline__ void fk_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The 2 x 2 neighbourhood at window element (lr, lc) — the caller guarantees that it lies inside the window.
__device__ __forceinline__ float win_bilinear(const uint16_t *win, int wcols, int idx, float w_tl, float w_tr, float w_bl, float w_br) {
    const unsigned a = win[idx];
    const unsigned bb = win[idx + wcols];
    return w_tl * (float)(a & 0xFFu) + w_tr * (float)(a >> 8) + w_bl * (float)(bb & 0xFFu) + w_br * (float)(bb >> 8);
}

constexpr int kFkCurQuads = 4;  // 8-byte window loads a lane may hold in flight for the next level's current window (21 x 21: 28 x 8 quads)
constexpr int kFkExPasses = 3;  // (extended-patch row, quad of four columns) pairs a lane may hold: 15 x 4 = 60 at 13 x 13, 23 x 6 = 138 at 21 x 21

// Pitch of the extended patch in LDS: whole quads of four columns (the register-fed extraction stores a float4 per lane)
__host__ __device__ inline int fk_ex_quads(const KltParams &p) { return (p.ex_cols + 3) >> 2; }
__host__ __device__ inline int fk_ex_pitch(const KltParams &p) { return 4 * fk_ex_quads(p); }

// The image rows under the extended patch, straight from global memory into registers: lane (row, quad) holds bytes
// [4 quad, 4 quad + 8) of image rows `row` and `row + 1` of the footprint.  Only for footprints that lie inside the image with room
// for the 8-byte reads (fk_ref_interior): then every element of the extended patch is valid, and nothing is staged through LDS.
template <int N>
struct RefRows {
    uint32_t x0[N], y0[N], x1[N], y1[N];
};

__device__ __forceinline__ bool fk_ref_interior(const KltParams &p, const DevImage &im, int r_lo, int c_lo) {
    return r_lo >= 0 && c_lo >= 0 && (long long)r_lo + p.ex_rows + 1 <= im.rows && (long long)c_lo + fk_ex_pitch(p) + 4 <= im.cols &&
           p.ex_rows * fk_ex_quads(p) <= kFkExPasses * kWave;
}

template <int N>
__device__ __forceinline__ void fk_issue_ref_rows(RefRows<N> &q, int lane, const KltParams &p, const DevImage &im, int r_lo, int c_lo) {
    const int quads = fk_ex_quads(p), total = p.ex_rows * quads;
    const uint8_t *base = im.data + (long long)r_lo * im.cols + c_lo;  // wave-uniform
    const uint32_t m20 = (uint32_t)(((1u << 20) + (uint32_t)quads - 1) / (uint32_t)quads);  // idx < 4096: (idx * m20) >> 20 == idx / quads
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k * kWave < total) {  // compile-time for the specialised geometries
            int idx = lane + k * kWave;
            idx = idx < total ? idx : 0;
            const int r = (int)(__umul24((unsigned)idx, m20) >> 20);
            const int qd = idx - imul(r, quads);
            const uint8_t *src = base + (size_t)(unsigned)(imul(r, im.cols) + 4 * qd);
            __builtin_memcpy(&q.x0[k], src, 4);
            __builtin_memcpy(&q.y0[k], src + 4, 4);
            __builtin_memcpy(&q.x1[k], src + im.cols, 4);
            __builtin_memcpy(&q.y1[k], src + im.cols + 4, 4);
        }
    }
}

// ExtractExtendPatchInReferenceImage (optical_flow.cpp:49-102) for an interior footprint: four elements per lane from the bytes it
// holds, with the reference's weight products and sum order; every element is valid.
template <int N>
__device__ __forceinline__ void fk_ex_from_rows(const RefRows<N> &q, int lane, const KltParams &p, float *ex, float w_tl, float w_tr, float w_bl, float w_br) {
    const int quads = fk_ex_quads(p), total = p.ex_rows * quads;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k * kWave < total) {
            const int idx = lane + k * kWave;
            if (idx < total) {
                const float a0 = (float)(q.x0[k] & 0xFFu), a1 = (float)((q.x0[k] >> 8) & 0xFFu), a2 = (float)((q.x0[k] >> 16) & 0xFFu), a3 = (float)(q.x0[k] >> 24),
                            a4 = (float)(q.y0[k] & 0xFFu);
                const float b0 = (float)(q.x1[k] & 0xFFu), b1 = (float)((q.x1[k] >> 8) & 0xFFu), b2 = (float)((q.x1[k] >> 16) & 0xFFu), b3 = (float)(q.x1[k] >> 24),
                            b4 = (float)(q.y1[k] & 0xFFu);
                float4 v;
                v.x = w_tl * a0 + w_tr * a1 + w_bl * b0 + w_br * b1;
                v.y = w_tl * a1 + w_tr * a2 + w_bl * b1 + w_br * b2;
                v.z = w_tl * a2 + w_tr * a3 + w_bl * b2 + w_br * b3;
                v.w = w_tl * a3 + w_tr * a4 + w_bl * b3 + w_br * b4;
                reinterpret_cast<float4 *>(ex)[idx] = v;  // idx = row * quads + quad: the pitch is 4 * quads floats
            }
        }
    }
}

// Eigen's LDLT of the 2 x 2 Hessian (ldlt_solve<2>, klt_common.h) split in two: the factorisation is a function of H alone, H is
// fixed for a level (basic_klt_fast.cpp:64-99), so it is done once per level and every iteration only runs the substitution — the
// same operations on the same operands in the same order as ldlt_solve<2>, hence the same bits.
struct Ldlt2 {
    float d0, l10, d1;
    bool swapped;
};

__device__ __forceinline__ Ldlt2 ldlt2_factor(float h00, float h10, float h11) {
    Ldlt2 f;
    float m00 = h00, m10 = h10, m11 = h11;
    f.swapped = fabsf(m11) > fabsf(m00);  // the first maximum of the diagonal magnitudes
    if (f.swapped) {
        swap_values(m00, m11);  // a transposition of a 2 x 2 symmetric matrix moves the diagonal only
    }
    const bool pivot_valid = fabsf(m00) > 0.0f;
    if (!pivot_valid) {
        f.swapped = false;  // Eigen stops: identity transpositions, the matrix as it is
    } else {
        m10 /= m00;
        const float temp0 = m00 * m10;
        const float dot = m10 * temp0;
        m11 -= dot;
    }
    f.d0 = m00;
    f.l10 = m10;
    f.d1 = m11;
    return f;
}

// lane: the two divisions of D^+ run side by side on lanes 0 / 1 (the operands are wave-uniform, all 64 lanes execute)
__device__ __forceinline__ void ldlt2_apply(const Ldlt2 &f, float b0, float b1, float &x0, float &x1, int lane) {
    float y0 = f.swapped ? b1 : b0, y1 = f.swapped ? b0 : b1;
    y1 -= f.l10 * y0;
    const float num = lane == 0 ? y0 : y1, den = lane == 0 ? f.d0 : f.d1;
    const float quot = (fabsf(den) > 1.17549435e-38f) ? num / den : 0.0f;
    y0 = uniform_lane(quot, 0);
    y1 = uniform_lane(quot, 1);
    y0 -= f.l10 * y1;
    x0 = f.swapped ? y1 : y0;
    x1 = f.swapped ? y0 : y1;
}

#ifndef FTK_WAVES_PER_EU
#define FTK_WAVES_PER_EU 4
#endif

// MODEL: FTK_MODEL_BASIC or FTK_MODEL_AFFINE.
// HR / HC: the half patch sizes as compile-time constants (the geometry folds into immediates), or 0 / 0 for "as passed".
template <int MODEL, int HR, int HC>
__global__ void __attribute__((amdgpu_waves_per_eu(FTK_WAVES_PER_EU))) __launch_bounds__(256) klt_fast_kernel(const KltParams p_arg) {
#ifdef FTK_STAMPS
    const unsigned long long stamp_kernel_t0 = __builtin_amdgcn_s_memtime();
#endif
    klt_touch_kernarg<sizeof(KltParams)>();
    KltParams p = p_arg;  // everything but the level tables (a run-time level index into a local copy would put it in scratch)
    if constexpr (HR > 0 && HC > 0) {
        p.half_rows = HR;
        p.half_cols = HC;
        klt_fill_geometry(p);
    }
    extern __shared__ float4 lds_raw[];
    uint32_t block = blockIdx.x;
    if (p.sort_iters) {
        // block 0 of a launch that carries the previous call's iteration counts sorts them into a later call's launch order
        if (block == 0) {
            klt_order_block(p.sort_iters, p.sort_order_out, p.n, reinterpret_cast<int *>(lds_raw), p.sort_ref_uv, p_arg.ref[0].cols, p_arg.ref[0].rows,
                            max(p.features_per_group, 1), p.sched_flags, p.sched_call);
            return;
        }
        block -= 1;
    }
    Blk b;
    b.solo = true;
    b.tid = b.lane = (int)(threadIdx.x & (kWave - 1));
    b.nt = kWave;
    b.wave = 0;
    b.nwaves = 1;
    const int lane = b.lane;
    const uint32_t group_slot = threadIdx.x >> 6;
    uint32_t id = block * (uint32_t)p.features_per_group + group_slot;
    if (id >= (uint32_t)p.n) {
        return;
    }
    const bool younger = 2u * id >= (uint32_t)p.n;
    if (p.order) {
        id = (uint32_t)p.order[id];  // launch slot -> feature
    }
    // the three per-feature inputs are requested together (one global round trip, not one after the other)
    const float2 in_uv = reinterpret_cast<const float2 *>(p.cur_uv_in)[id];
    const float2 full_ref = reinterpret_cast<const float2 *>(p.ref_uv)[id];
    uint8_t status = p.status_in[id];
    const float in_u = in_uv.x, in_v = in_uv.y;
    // features beyond kMaxTrackPointsNumber and features that already failed are passed through (basic_klt.cpp:9,15)
    if (id >= p.n_track || status > FTK_TRACKED) {
        if (lane == 0) {
            p.cur_uv_out[2 * id] = in_u;
            p.cur_uv_out[2 * id + 1] = in_v;
            p.status_out[id] = status;
            if (p.iters) {
                p.iters[id] = 0;
            }
            if (p.sched_iters) {
                p.sched_iters[id] = 0;
            }
        }
        return;
    }
    const FkLds c = fk_carve<MODEL>(lds_raw + (size_t)group_slot * (p.group_lds_stride >> 4), p);
    const int pitch = fk_term_pitch(p);
    const int exp = fk_ex_pitch(p);

    // basic_klt.cpp:10,18-19 (pyramid) / :59-86 (single level)
    const float full_ref_u = full_ref.x, full_ref_v = full_ref.y;
    const float scale = p.single_level ? 1.0f : (float)(1 << (p.n_levels - 1));
    float ref_u = p.single_level ? full_ref_u : full_ref_u / scale;
    float ref_v = p.single_level ? full_ref_v : full_ref_v / scale;
    float cur_u = p.single_level ? in_u : in_u / scale;
    float cur_v = p.single_level ? in_v : in_v / scale;
    // affine_klt.cpp:21,70: the warp starts from the identity on the pyramid path, from the prediction on the single-level one;
    // it is carried unchanged from level to level
    float a00 = p.single_level ? p.prior[0] : 1.0f, a01 = p.single_level ? p.prior[1] : 0.0f;
    float a10 = p.single_level ? p.prior[2] : 0.0f, a11 = p.single_level ? p.prior[3] : 1.0f;

    const int rrows = p.rwin_rows, rcols = p.rwin_cols;
    const bool cur_fits = p.cwin_rows * (p.cwin_cols >> 2) <= kFkCurQuads * kWave;

#ifdef FTK_STAMPS
    b.stamp_t0 = stamp_kernel_t0;
#endif
    // ---- the coarsest level's inputs: the reference rows into registers, the current window into LDS ----
    Win cw;
    cw.data = c.cur_win;
    cw.rows = p.cwin_rows;
    cw.cols = p.cwin_cols;
    RefRows<kFkExPasses> qr;   // the image rows under THIS level's extended patch (requested one level ahead)
    bool ref_interior;         // ... are valid (else the level stages a clamped window through LDS)
    int r_lo, c_lo;            // origin of this level's reference footprint
    {
        const int top = p.n_levels - 1;
        const DevImage ref = p_arg.ref[top], cur = p_arg.cur[top];
        footprint_origin(p, ref_u, ref_v, r_lo, c_lo);
        ref_interior = fk_ref_interior(p, ref, r_lo, c_lo);
        if (ref_interior) {
            fk_issue_ref_rows(qr, lane, p, ref, r_lo, c_lo);
        }
        int need_r, need_c;
        footprint_origin(p, cur_u, cur_v, need_r, need_c);
        cw.r_lo = wadd(need_r, -p.cwin_margin);
        cw.c_lo = wadd(need_c, -p.cwin_margin);
        win_set_cover(p, cw);
        if (cur_fits && window_inside(cur, cw.r_lo, cw.c_lo, cw.rows, cw.cols)) {
            RawQuads<kFkCurQuads> qc;
            issue_quads(qc, b, cur, cw.r_lo, cw.c_lo, cw.rows, cw.cols, p.magic_cwq);
            // the zero padding of the term rows (pixels P .. pitch - 1 are never written again) while the loads fly
            // (the affine ring has none: every sweep writes all 64 columns of the rows it chains)
            if constexpr (MODEL == FTK_MODEL_BASIC) {
                for (int idx = lane; idx < kFkTerms * (pitch - p.P); idx += kWave) {
                    const int k = idx / (pitch - p.P);
                    c.terms[k * pitch + p.P + (idx - k * (pitch - p.P))] = 0.0f;
                }
            }
            store_quads(qc, b, c.cur_win, cw.rows, cw.cols, p.magic_cwq);
        } else {
            if constexpr (MODEL == FTK_MODEL_BASIC) {
                for (int idx = lane; idx < kFkTerms * (pitch - p.P); idx += kWave) {
                    const int k = idx / (pitch - p.P);
                    c.terms[k * pitch + p.P + (idx - k * (pitch - p.P))] = 0.0f;
                }
            }
            stage_any(opaque_blk(b), cur, c.cur_win, cw.r_lo, cw.c_lo, cw.rows, cw.cols, p.magic_cwc, p.magic_cwq);
        }
    }
    fk_fence();
    FTK_STAMP_END(b, 0);

    uint32_t iters = 0;
    float out_u = in_u, out_v = in_v;
    const int n_pass = (p.P + kWave - 1) / kWave;
    for (int level = p.n_levels - 1; level > -1; --level) {
        const DevImage ref = p_arg.ref[level];
        const DevImage cur = p_arg.cur[level];
        set_level_priority(level, younger);
        FTK_STAMP_BEGIN(b);
        // ---- ExtractExtendPatchInReferenceImage (optical_flow.cpp:49-102): ONE weight set, integer lattice floor(ref) - ex / 2 ----
        uint32_t ref_valid = 0;
        {
            const float int_row = floorf(ref_v), int_col = floorf(ref_u);
            const float dec_row = ref_v - int_row, dec_col = ref_u - int_col;
            const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
            const float w_tr = (1.0f - dec_row) * dec_col;
            const float w_bl = dec_row * (1.0f - dec_col);
            const float w_br = dec_row * dec_col;
            if (ref_interior) {
                fk_ex_from_rows(qr, lane, p, c.ex, w_tl, w_tr, w_bl, w_br);
                ref_valid = (uint32_t)p.E;
            } else {
                // a footprint that touches the border: a clamped window through LDS; element (erow, ecol) of the extended patch is the
                // bilinear value at window element (erow, ecol) — the window's origin IS the patch's (footprint_origin)
                stage_any(opaque_blk(b), ref, c.ref_win, r_lo, c_lo, rrows, rcols, p.magic_rwc, p.magic_rwq);
                fk_fence();
                const int min_row = wadd(f2i(int_row), -(p.ex_rows / 2));
                const int min_col = wadd(f2i(int_col), -(p.ex_cols / 2));
#pragma unroll 2
                for (int base = 0; base < p.E; base += kWave) {
                    const int e = base + lane;
                    bool valid = false;
                    if (e < p.E) {
                        const int erow = (int)__umulhi((unsigned)e, p.magic_exc);
                        const int ecol = e - imul(erow, p.ex_cols);
                        const int row = wadd(min_row, erow);
                        const int col = wadd(min_col, ecol);
                        valid = !(row < 0 || row > ref.rows - 2 || col < 0 || col > ref.cols - 2);
                        const float value = win_bilinear(c.ref_win, rcols, imul(erow, rcols) + ecol, w_tl, w_tr, w_bl, w_br);
                        c.ex[imul(erow, exp) + ecol] = valid ? value : kFkInvalid;
                    }
                    ref_valid += (uint32_t)__popcll(wave_ballot(valid));
                }
            }
        }
        const bool all_ref_valid = ref_interior;
        fk_fence();
        FTK_STAMP_END(b, 1);
        // ---- the next level's inputs are requested now and arrive while this level runs ----
        RawQuads<kFkCurQuads> qcn;
        int nr_lo = 0, nc_lo = 0, ncr_lo = 0, ncc_lo = 0;
        bool next_interior = false, next_cur_async = false;
        if (level > 0) {
            const DevImage nref = p_arg.ref[level - 1], ncur = p_arg.cur[level - 1];
            footprint_origin(p, ref_u * 2.0f, ref_v * 2.0f, nr_lo, nc_lo);
            next_interior = fk_ref_interior(p, nref, nr_lo, nc_lo);
            if (next_interior) {
                fk_issue_ref_rows(qr, lane, p, nref, nr_lo, nc_lo);  // this level's rows are consumed: the registers are free
            }
            int need_r, need_c;
            footprint_origin(p, cur_u * 2.0f, cur_v * 2.0f, need_r, need_c);  // a guess: twice the position this level STARTS from
            ncr_lo = wadd(need_r, -p.cwin_margin);
            ncc_lo = wadd(need_c, -p.cwin_margin);
            next_cur_async = cur_fits && window_inside(ncur, ncr_lo, ncc_lo, cw.rows, cw.cols);
            if (next_cur_async) {
                issue_quads(qcn, b, ncur, ncr_lo, ncc_lo, cw.rows, cw.cols, p.magic_cwq);
            }
        }
        FTK_STAMP_END(b, 4);
        bool level_runs = true;
        if (ref_valid == 0) {
            status = FTK_OUTSIDE;  // basic_klt_fast.cpp:12-16
            level_runs = false;
        }
        if (level_runs) {
            if constexpr (MODEL == FTK_MODEL_BASIC) {
                // ---- PrecomputeJacobianAndHessian (basic_klt_fast.cpp:64-99): dx = dy = 0 where a 4-neighbour is missing ----
#pragma unroll 4
                for (int pass = 0; pass < n_pass; ++pass) {
                    const int pxi = pass * kWave + lane;
                    if (pxi < p.P) {
                        int prow, pcol;
                        pixel_rc(p, pxi, prow, pcol);
                        const int ei = imul(prow + 1, exp) + pcol + 1;
                        // five independent reads, no branch: the sign bits ARE the validity flags
                        const float e_l = c.ex[ei - 1], e_r = c.ex[ei + 1], e_t = c.ex[ei - exp], e_b = c.ex[ei + exp], e_c = c.ex[ei];
                        float dx = e_r - e_l, dy = e_b - e_t;
                        int usable = -1;
                        if (!all_ref_valid) {  // wave-uniform
                            const bool grad = (__float_as_int(e_l) | __float_as_int(e_r) | __float_as_int(e_t) | __float_as_int(e_b)) >= 0;
                            dx = grad ? dx : 0.0f;
                            dy = grad ? dy : 0.0f;
                            usable = __float_as_int(e_c) >= 0 ? -1 : 0;
                        }
                        c.rec[pxi] = make_float4(dx, dy, e_c, __int_as_float(usable));
                        c.terms[2 * pitch + pxi] = dx * dx;
                        c.terms[3 * pitch + pxi] = dx * dy;
                        c.terms[4 * pitch + pxi] = dy * dy;
                    }
                }
                fk_fence();
                FTK_STAMP_END(b, 2);
                status = FTK_LARGE_RESIDUAL;  // basic_klt_fast.cpp:29
                float last_squared_step = INFINITY;
                uint32_t large_step_cnt = 0;
                Ldlt2 fac = {0.0f, 0.0f, 0.0f, false};
                for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
                    ++iters;
                    FTK_STAMP_BEGIN(b);
                    if (!win_covers(cw, cur_u, cur_v)) {
                        // the patch has left the window (or the integer test has to decide): restage around the present position
                        int need_r, need_c;
                        footprint_origin(p, cur_u, cur_v, need_r, need_c);
                        const long long nr = need_r, nc = need_c;
                        const bool covered = nr >= (long long)cw.r_lo && nr + (2 * p.half_rows + 4) <= (long long)cw.r_lo + cw.rows &&
                                             nc >= (long long)cw.c_lo && nc + (2 * p.half_cols + 4) <= (long long)cw.c_lo + cw.cols + 1;
                        if (!covered) {
                            cw.r_lo = wadd(need_r, -p.cwin_margin);
                            cw.c_lo = wadd(need_c, -p.cwin_margin);
                            win_set_cover(p, cw);
                            stage_any(opaque_blk(b), cur, c.cur_win, cw.r_lo, cw.c_lo, cw.rows, cw.cols, p.magic_cwc, p.magic_cwq);
                            fk_fence();
                        }
                    }
                    // ---- ComputeBias (basic_klt_fast.cpp:101-195): integer lattice floor(cur) - patch / 2, one weight set ----
                    const float int_row = floorf(cur_v), int_col = floorf(cur_u);
                    const float dec_row = cur_v - int_row, dec_col = cur_u - int_col;
                    const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
                    const float w_tr = (1.0f - dec_row) * dec_col;
                    const float w_bl = dec_row * (1.0f - dec_col);
                    const float w_br = dec_row * dec_col;
                    const int min_row = __builtin_amdgcn_readfirstlane(wadd(f2i(int_row), -(p.patch_rows / 2)));
                    const int min_col = __builtin_amdgcn_readfirstlane(wadd(f2i(int_col), -(p.patch_cols / 2)));
                    const int rel_r = (int)((unsigned)min_row - (unsigned)cw.r_lo), rel_c = (int)((unsigned)min_col - (unsigned)cw.c_lo);
                    // the whole lattice (+ 1 neighbours) inside the image: no per-pixel bounds test (wave-uniform)
                    const bool all_inside = min_row >= 0 && (long long)min_row + p.patch_rows - 1 <= (long long)cur.rows - 2 && min_col >= 0 &&
                                            (long long)min_col + p.patch_cols - 1 <= (long long)cur.cols - 2;
                    uint32_t n_valid = 0;
                    if (all_inside && all_ref_valid) {
                        // the usual case: every pixel is used — no selects, no counts (the window covers the lattice: win_covers above)
                        const int rel = imul(rel_r, cw.cols) + rel_c;
#pragma unroll 4
                        for (int pass = 0; pass < n_pass; ++pass) {
                            const int pxi = pass * kWave + lane;
                            if (pxi < p.P) {
                                int prow, pcol;
                                pixel_rc(p, pxi, prow, pcol);
                                const float4 rec = c.rec[pxi];
                                const float i_cur = win_bilinear(c.cur_win, cw.cols, rel + imul(prow, cw.cols) + pcol, w_tl, w_tr, w_bl, w_br);
                                const float dt = i_cur - rec.z;
                                c.terms[pxi] = -(rec.x * dt);
                                c.terms[pitch + pxi] = -(rec.y * dt);
                            }
                        }
                        n_valid = (uint32_t)p.P;
                    } else {
#pragma unroll 2
                        for (int pass = 0; pass < n_pass; ++pass) {
                            const int pxi = pass * kWave + lane;
                            bool ok = false;
                            if (pxi < p.P) {
                                int prow, pcol;
                                pixel_rc(p, pxi, prow, pcol);
                                const float4 rec = c.rec[pxi];
                                const int lr = rel_r + prow, lc = rel_c + pcol;
                                const bool in_win = (unsigned)lr < (unsigned)(cw.rows - 1) && (unsigned)lc < (unsigned)cw.cols;  // always, for a pixel inside the image
                                const int row = wadd(min_row, prow), col = wadd(min_col, pcol);
                                const bool in_img = !(row < 0 || row > cur.rows - 2 || col < 0 || col > cur.cols - 2);
                                ok = in_img && in_win && __float_as_int(rec.w) != 0;
                                const float i_cur = win_bilinear(c.cur_win, cw.cols, in_win ? imul(lr, cw.cols) + lc : 0, w_tl, w_tr, w_bl, w_br);
                                const float dt = i_cur - rec.z;
                                c.terms[pxi] = ok ? -(rec.x * dt) : 0.0f;
                                c.terms[pitch + pxi] = ok ? -(rec.y * dt) : 0.0f;
                            }
                            n_valid += (uint32_t)__popcll(wave_ballot(ok));
                        }
                    }
                    fk_fence();
                    FTK_STAMP_END(b, 3);
                    if (n_valid == 0) {
                        break;  // basic_klt_fast.cpp:40-42
                    }
                    // the exact-order sums: lanes 0 / 1 the bias, and in the level's first iteration lanes 2 - 4 the Hessian
                    const int chains = iter == 0 ? kFkTerms : 2;
                    float acc = 0.0f;
#if FTK_FK_QUAD_CHAIN
                    // every lane: quad q carries sum q (klt_common.h "quad chain"); the quads behind the last sum follow its row, ignored
                    constexpr int kSumLanes = 4;
                    acc = chain_quads_row<true>(0.0f, c.terms + min(lane >> 2, chains - 1) * pitch + 4 * (lane & 3), p.Ppad >> 4);
#else
                    constexpr int kSumLanes = 1;
                    if (lane < chains) {
                        acc = chain_lane(c.terms + lane * pitch, p.Ppad);
                    }
#endif
                    const int acc_bits = __float_as_int(acc);
                    FTK_STAMP_END(b, 5);
                    if (iter == 0) {
                        fac = ldlt2_factor(__int_as_float(__builtin_amdgcn_readlane(acc_bits, 2 * kSumLanes)), __int_as_float(__builtin_amdgcn_readlane(acc_bits, 3 * kSumLanes)),
                                           __int_as_float(__builtin_amdgcn_readlane(acc_bits, 4 * kSumLanes)));
                    }
                    float v0, v1;
                    ldlt2_apply(fac, __int_as_float(__builtin_amdgcn_readlane(acc_bits, 0)), __int_as_float(__builtin_amdgcn_readlane(acc_bits, 1 * kSumLanes)), v0, v1, lane);  // basic_klt_fast.cpp:44
                    if (isnan(v0) || isnan(v1)) {
                        status = FTK_NUMERIC_ERROR;
                        break;
                    }
                    cur_u += v0;
                    cur_v += v1;
                    FTK_STAMP_END(b, 6);
                    if (fast_step_logic(p, v0 * v0 + v1 * v1, last_squared_step, large_step_cnt, status)) {
                        break;
                    }
                }
            } else {
                // ---- PrecomputeJacobianAndHessian, the per-pixel part (affine_klt_fast.cpp:71-94): dx = dy = 0 where a 4-neighbour is missing ----
#pragma unroll 4
                for (int pass = 0; pass < n_pass; ++pass) {
                    const int pxi = pass * kWave + lane;
                    if (pxi < p.P) {
                        int prow, pcol;
                        pixel_rc(p, pxi, prow, pcol);
                        const int ei = imul(prow + 1, exp) + pcol + 1;
                        const float e_l = c.ex[ei - 1], e_r = c.ex[ei + 1], e_t = c.ex[ei - exp], e_b = c.ex[ei + exp], e_c = c.ex[ei];
                        float dx = e_r - e_l, dy = e_b - e_t;
                        int flags = 3;
                        if (!all_ref_valid) {  // wave-uniform
                            const bool grad = (__float_as_int(e_l) | __float_as_int(e_r) | __float_as_int(e_t) | __float_as_int(e_b)) >= 0;
                            dx = grad ? dx : 0.0f;
                            dy = grad ? dy : 0.0f;
                            flags = (__float_as_int(e_c) >= 0 ? 1 : 0) | (grad ? 2 : 0);
                        }
                        c.rec[pxi] = make_float4(dx, dy, e_c, __int_as_float(flags));
                    }
                }
                fk_fence();
                FTK_STAMP_END(b, 2);
                status = FTK_LARGE_RESIDUAL;  // affine_klt_fast.cpp:29
                float last_squared_step = INFINITY;
                uint32_t large_step_cnt = 0;
                Ldlt6 fac;
                fac.perm = 0;
                fac.d_mine = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    fac.l[i] = 0.0f;
                }
                float *const ring_col = c.terms + lane;             // this lane's column of the ring (first iteration's sweeps)
                const float *const ring_row = c.terms + imul(lane < kAfRows ? lane : 0, kChunkRow);  // ... and its row (chain lanes)
                // One patch pixel of ComputeBias (affine_klt_fast.cpp:140-188): the current image at the warped position, and the
                // factors of the six bias products — zeroed for an unused pixel, so that it contributes exact zeros (five selects
                // instead of one per product; the products are then +0 or -0, and x + (+-0) == x for every value a sum that started
                // at +0 can hold).  Returns whether the pixel is used.
                auto bias_factors = [&](int pxi, bool in_patch, int &prow, int &pcol, float4 &rec, float &dt, float &bx, float &by, float &bdx,
                                        float &bdy) -> bool {
                    pixel_rc(p, in_patch ? pxi : 0, prow, pcol);
                    rec = c.rec[in_patch ? pxi : 0];
                    const float dcol = (float)(pcol - p.half_cols);
                    const float drow = (float)(prow - p.half_rows);
                    const float warped_x = a00 * dcol + a01 * drow;
                    const float warped_y = a10 * dcol + a11 * drow;
                    const float row_c = warped_y + cur_v;
                    const float col_c = warped_x + cur_u;
                    const Axis ar = make_axis(row_c, cur.rows - 1), ac = make_axis(col_c, cur.cols - 1);
                    bool hit = true;
                    float i_cur = tap(cw, ar, ac, hit);
                    const bool valid = ar.valid && ac.valid;
                    if (valid && !hit) {
                        sample(cur, cw, row_c, col_c, i_cur);  // a warped tap outside the window: global memory, same arithmetic
                    }
#ifdef FTK_STAMPS_MISSES
                    b.stamp_acc[2] += 100ull * (unsigned long long)__popcll(wave_ballot(in_patch && valid && !hit));  // diagnostic: taps that left the window
                    b.stamp_acc[0] += 100ull * (unsigned long long)__popcll(wave_ballot(in_patch && !valid));        // ... and taps outside the image
#endif
                    const bool ok = in_patch && valid && (__float_as_int(rec.w) & 1) != 0;
                    dt = ok ? i_cur - rec.z : 0.0f;
                    bx = ok ? col_c : 0.0f;
                    by = ok ? row_c : 0.0f;
                    bdx = ok ? rec.x : 0.0f;
                    bdy = ok ? rec.y : 0.0f;
                    return ok;
                };
                for (uint32_t iter = 0; iter < p.max_iteration; ++iter) {
                    ++iters;
                    FTK_STAMP_BEGIN(b);
                    if (!win_covers(cw, cur_u, cur_v)) {
                        // the (unwarped) patch has left the window: restage around the present position; warped taps outside the window
                        // are sampled from global memory with the same arithmetic
                        int need_r, need_c;
                        footprint_origin(p, cur_u, cur_v, need_r, need_c);
                        const long long nr = need_r, nc = need_c;
                        const bool covered = nr >= (long long)cw.r_lo && nr + (2 * p.half_rows + 4) <= (long long)cw.r_lo + cw.rows &&
                                             nc >= (long long)cw.c_lo && nc + (2 * p.half_cols + 4) <= (long long)cw.c_lo + cw.cols + 1;
                        if (!covered) {
                            cw.r_lo = wadd(need_r, -p.cwin_margin);
                            cw.c_lo = wadd(need_c, -p.cwin_margin);
                            win_set_cover(p, cw);
                            stage_any(opaque_blk(b), cur, c.cur_win, cw.r_lo, cw.c_lo, cw.rows, cw.cols, p.magic_cwc, p.magic_cwq);
                            fk_fence();
                        }
                    }
                    const bool first = iter == 0;
                    float acc = 0.0f;
                    uint32_t n_valid = 0;
                    if (first) {
                        // ---- the level's first iteration: the six bias products AND the Hessian's 18 (affine_klt_fast.cpp:95-129), 64 pixels
                        // at a time: sweep into the ring, then the exact-order sums of those 64 terms continue on 24 chain lanes ----
                        for (int base = 0; base < p.P; base += kChunkPixels) {
                            const int pxi = base + lane;
                            const bool in_patch = pxi < p.P;
                            int prow, pcol;
                            float4 rec;
                            float dt, bx, by, bdx, bdy;
                            const bool ok = bias_factors(pxi, in_patch, prow, pcol, rec, dt, bx, by, bdx, bdy);
                            ring_col[0 * kChunkRow] = -(dt * bx * bdx);
                            ring_col[1 * kChunkRow] = -(dt * bx * bdy);
                            ring_col[2 * kChunkRow] = -(dt * by * bdx);
                            ring_col[3 * kChunkRow] = -(dt * by * bdy);
                            ring_col[4 * kChunkRow] = -(dt * bdx);
                            ring_col[5 * kChunkRow] = -(dt * bdy);
                            n_valid += (uint32_t)__popcll(wave_ballot(ok));
                            // x, y: the patch offset + the position at level entry — cur_uv has not moved yet (affine_klt_fast.cpp:95-96)
                            const bool grad = in_patch && (__float_as_int(rec.w) & 2) != 0;
                            const float x = grad ? (float)(pcol - p.half_cols) + cur_u : 0.0f;
                            const float y = grad ? (float)(prow - p.half_rows) + cur_v : 0.0f;
                            const float dx = grad ? rec.x : 0.0f, dy = grad ? rec.y : 0.0f;
                            const float xx = x * x, yy = y * y, xy = x * y;
                            const float dxdx = dx * dx, dydy = dy * dy, dxdy = dx * dy;
                            float *const h = ring_col + 6 * kChunkRow;
                            h[A_XX_DXDX * kChunkRow] = xx * dxdx;
                            h[A_XX_DXDY * kChunkRow] = xx * dxdy;
                            h[A_XY_DXDX * kChunkRow] = xy * dxdx;
                            h[A_XY_DXDY * kChunkRow] = xy * dxdy;
                            h[A_X_DXDX * kChunkRow] = x * dxdx;
                            h[A_X_DXDY * kChunkRow] = x * dxdy;
                            h[A_XX_DYDY * kChunkRow] = xx * dydy;
                            h[A_XY_DYDY * kChunkRow] = xy * dydy;
                            h[A_X_DYDY * kChunkRow] = x * dydy;
                            h[A_YY_DXDX * kChunkRow] = yy * dxdx;
                            h[A_YY_DXDY * kChunkRow] = yy * dxdy;
                            h[A_Y_DXDX * kChunkRow] = y * dxdx;
                            h[A_Y_DXDY * kChunkRow] = y * dxdy;
                            h[A_YY_DYDY * kChunkRow] = yy * dydy;
                            h[A_Y_DYDY * kChunkRow] = y * dydy;
                            h[A_DXDX * kChunkRow] = dxdx;
                            h[A_DXDY * kChunkRow] = dxdy;
                            h[A_DYDY * kChunkRow] = dydy;
                            fk_fence();
                            if (lane < kAfRows) {
                                const int left = p.P - base;  // a short last chunk is chained to a multiple of four (lanes past the patch wrote zeros)
                                acc = left >= kChunkPixels ? chain_chunk(acc, ring_row) : chain_lane(ring_row, (left + 3) & ~3, acc);
                            }
                            fk_fence();  // the next sweep's stores stay behind these reads
                        }
                    } else {
                        // ---- every later iteration: six whole rows [6][pitch] over the ring's space — all passes of the sweep first (their
                        // LDS round trips overlap), then ONE uninterrupted chain per sum ----
#pragma unroll 4
                        for (int pass = 0; pass < n_pass; ++pass) {
                            const int pxi = pass * kWave + lane;
                            int prow, pcol;
                            float4 rec;
                            float dt, bx, by, bdx, bdy;
                            const bool ok = bias_factors(pxi, pxi < p.P, prow, pcol, rec, dt, bx, by, bdx, bdy);
                            if (pxi < p.Ppad) {  // the padding columns P .. Ppad - 1 are rewritten too: the ring's sweeps run over them
                                c.terms[0 * pitch + pxi] = -(dt * bx * bdx);
                                c.terms[1 * pitch + pxi] = -(dt * bx * bdy);
                                c.terms[2 * pitch + pxi] = -(dt * by * bdx);
                                c.terms[3 * pitch + pxi] = -(dt * by * bdy);
                                c.terms[4 * pitch + pxi] = -(dt * bdx);
                                c.terms[5 * pitch + pxi] = -(dt * bdy);
                            }
                            n_valid += (uint32_t)__popcll(wave_ballot(ok));  // (every lane counts every pass: the total stays wave-uniform)
                        }
                        fk_fence();
                        FTK_STAMP_END(b, 3);
#if FTK_FK_QUAD_CHAIN
                        acc = chain_quads_row<true>(0.0f, c.terms + imul(min(lane >> 2, 5), pitch) + 4 * (lane & 3), p.Ppad >> 4);  // quad q carries bias sum q
#else
                        if (lane < 6) {
                            acc = chain_lane(c.terms + imul(lane, pitch), p.Ppad);
                        }
#endif
                        fk_fence();
                    }
                    FTK_STAMP_END(b, 5);
                    if (n_valid == 0) {
                        break;  // affine_klt_fast.cpp:38-40
                    }
                    if (first) {
                        // The Hessian is fixed for the level: it is FACTORISED once (rows on lanes 0 - 5) and every iteration only runs
                        // the two substitutions — the factorisation is a pure function of H.  Lanes 6 + s publish sum s as the dense
                        // 6 x 6 (every alias of affine_klt_fast.cpp:130-132 written by the lane that owns the sum); the diagonal the
                        // pivot search needs comes from the accumulators themselves.
                        float *const dense = c.sums;
                        if (lane >= 6 && lane < kAfRows) {
                            const uint32_t slots = affine_dense_slots(lane - 6);
                            dense[slots & 0xffu] = acc;
                            dense[(slots >> 8) & 0xffu] = acc;
                            dense[(slots >> 16) & 0xffu] = acc;
                            dense[slots >> 24] = acc;
                        }
                        float ad_all[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            ad_all[j] = fabsf(bcast_lane(acc, 6 + kAffineDiagLane[j]));
                        }
                        float my_ad = ad_all[0];
#pragma unroll
                        for (int j = 1; j < 6; ++j) {
                            my_ad = (lane == j) ? ad_all[j] : my_ad;
                        }
                        fk_fence();
                        fac = ldlt6_factor_diag(my_ad, ad_all, Ldlt6Dense{dense}, lane);
                    }
                    // (the substitutions with L broadcast into wave-uniform registers — ~130 straight-line instructions instead of ~290
                    // with a cross-lane broadcast per term — were built and measured: 0.08 us per iteration SLOWER; docs/LAB_NOTES.md)
#if FTK_FK_QUAD_CHAIN
                    if (first ? lane < 6 : ((lane & 3) == 0 && lane < 24)) {  // the ring's chain lanes / the quads' first lanes hold the six bias sums
                        c.sums[40 + (first ? lane : lane >> 2)] = acc;
                    }
#else
                    if (lane < 6) {
                        c.sums[40 + lane] = acc;
                    }
#endif
                    fk_fence();
                    ldlt6_solve(fac, c.sums + 40, c.sums + 48, lane);  // affine_klt_fast.cpp:42
                    fk_fence();
                    float z[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        z[j] = c.sums[48 + j];
                    }
                    if (isnan(z[0]) || isnan(z[1]) || isnan(z[2]) || isnan(z[3]) || isnan(z[4]) || isnan(z[5])) {
                        status = FTK_NUMERIC_ERROR;
                        break;
                    }
                    const float v0 = (z[0] * cur_u + z[2] * cur_v) + z[4];  // affine_klt_fast.cpp:48
                    const float v1 = (z[1] * cur_u + z[3] * cur_v) + z[5];
                    cur_u += v0;
                    cur_v += v1;
                    a00 += z[0];  // col(0) += z.head<2>(), col(1) += z.segment<2>(2) (:50-53)
                    a10 += z[1];
                    a01 += z[2];
                    a11 += z[3];
                    FTK_STAMP_END(b, 6);
                    if (fast_step_logic(p, v0 * v0 + v1 * v1, last_squared_step, large_step_cnt, status)) {
                        break;
                    }
                }
            }
        }

        if (level == 0) {
            out_u = cur_u;
            out_v = cur_v;
            break;
        }
        ref_u *= 2.0f;
        ref_v *= 2.0f;
        cur_u *= 2.0f;
        cur_v *= 2.0f;
        // ---- the next level's inputs: what was requested at this level's entry, where it fits what is needed now ----
        FTK_STAMP_BEGIN(b);
        r_lo = nr_lo;
        c_lo = nc_lo;
        ref_interior = next_interior;
        bool cur_ready = false;
        if (next_cur_async) {
            // does the window requested around the GUESS cover the patch at the position the level ended with?
            Win guess = cw;
            guess.r_lo = ncr_lo;
            guess.c_lo = ncc_lo;
            win_set_cover(p, guess);
            if (win_covers(guess, cur_u, cur_v)) {
                cw = guess;
                store_quads(qcn, b, c.cur_win, cw.rows, cw.cols, p.magic_cwq);
                cur_ready = true;
            }
        }
        if (!cur_ready) {
            int need_r, need_c;
            footprint_origin(p, cur_u, cur_v, need_r, need_c);
            cw.r_lo = wadd(need_r, -p.cwin_margin);
            cw.c_lo = wadd(need_c, -p.cwin_margin);
            win_set_cover(p, cw);
            stage_any(opaque_blk(b), p_arg.cur[level - 1], c.cur_win, cw.r_lo, cw.c_lo, cw.rows, cw.cols, p.magic_cwc, p.magic_cwq);
        }
        fk_fence();
        FTK_STAMP_END(b, 4);
    }

    if (uv_outside(out_u, out_v, p_arg.cur[0])) {
        status = FTK_OUTSIDE;  // basic_klt.cpp:49-53
    }
    if (lane == 0) {
        p.cur_uv_out[2 * id] = out_u;
        p.cur_uv_out[2 * id + 1] = out_v;
        p.status_out[id] = status;
        if (p.iters) {
            p.iters[id] = iters;
        }
        tail_report(p, iters, id);  // the longest feature of the call, for the next call's wave policy
        sched_grid_record(p, full_ref_u, full_ref_v, out_u, out_v, iters);
        if (p.sched_iters) {
            p.sched_iters[id] = iters;
        }
    }
#ifdef FTK_STAMPS
    if (lane == 0 && p.stamps) {
        b.stamp_acc[7] = __builtin_amdgcn_s_memtime() - stamp_kernel_t0;
        for (int k = 0; k < 8; ++k) {
            p.stamps[(size_t)id * 8 + k] = b.stamp_acc[k];
        }
    }
#endif
}

template <int MODEL, int HR, int HC>
hipError_t fk_launch(const KltParams &p, size_t lds, hipStream_t stream) {
    void (*kernel)(const KltParams) = klt_fast_kernel<MODEL, HR, HC>;
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            return e;
        }
    }
    const unsigned sort_block = p.sort_iters ? 1u : 0u;
    const unsigned groups = (unsigned)((p.n + p.features_per_group - 1) / p.features_per_group);
    hipLaunchKernelGGL(kernel, dim3(groups + sort_block), dim3(kWave * p.features_per_group), lds, stream, p);
    return hipGetLastError();
}

}  // namespace

size_t klt_fast_lds_bytes(int model, const KltParams &p) {
    if (model != FTK_MODEL_BASIC && model != FTK_MODEL_AFFINE) {
        return 0;
    }
    const size_t one = fk_lds_bytes(model, p);
    return one * (size_t)(p.features_per_group > 1 ? p.features_per_group : 1);
}

hipError_t klt_fast_launch(int model, const KltParams &p_in, hipStream_t stream) {
    if (model != FTK_MODEL_BASIC && model != FTK_MODEL_AFFINE) {
        return hipErrorInvalidValue;
    }
    KltParams p = p_in;
    if (p.features_per_group < 1) {
        p.features_per_group = 1;
    }
    p.group_lds_stride = (int32_t)fk_lds_bytes(model, p);  // a multiple of 16
    size_t lds = klt_fast_lds_bytes(model, p);
    if (p.sort_iters && lds < (size_t)kOrderLdsBytes) {
        lds = kOrderLdsBytes;
    }
    static const bool specialise = !(getenv("FTK_FK_SPECIALISE") && atoi(getenv("FTK_FK_SPECIALISE")) == 0);  // experiment switch
    if (specialise && p.half_rows == p.half_cols) {
        KltParams check = p;
        klt_fill_geometry(check);  // what the specialised kernels recompute: it must be what the caller passed
        if (check.cwin_rows == p.cwin_rows && check.cwin_cols == p.cwin_cols && check.rwin_cols == p.rwin_cols && check.Ppad == p.Ppad) {
            if (model == FTK_MODEL_BASIC) {
                switch (p.half_rows) {
                    case 5: return fk_launch<FTK_MODEL_BASIC, 5, 5>(p, lds, stream);
                    case 6: return fk_launch<FTK_MODEL_BASIC, 6, 6>(p, lds, stream);
                    case 10: return fk_launch<FTK_MODEL_BASIC, 10, 10>(p, lds, stream);
                    default: break;
                }
            } else if (p.half_rows == 6) {
                return fk_launch<FTK_MODEL_AFFINE, 6, 6>(p, lds, stream);
            }
        }
    }
    return model == FTK_MODEL_BASIC ? fk_launch<FTK_MODEL_BASIC, 0, 0>(p, lds, stream) : fk_launch<FTK_MODEL_AFFINE, 0, 0>(p, lds, stream);
}

__global__ void klt_fast_warm_kernel() {}
hipError_t klt_fast_warm(hipStream_t stream) {
    hipLaunchKernelGGL(klt_fast_warm_kernel, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace ftk
