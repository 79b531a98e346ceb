// ftk_device.h — structures shared between the host-side C ABI (ftk_api.cpp) and the gfx950
// kernels (klt_kernels.hip, matcher_kernels.hip, pyramid_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ftk.h"

namespace ftk {

// Floats per pixel group (4 pixels x 24 sums + one float4 of padding) of the non-fast affine variants' product layout
// (klt_kernels.hip affine_all_terms); the host sizes KltParams::terms_floats with it.
constexpr int kAffineTermsGroupFloats = 4 * 24 + 4;
constexpr int kAffineTermsRoundGroups = 4;  // groups are allocated in whole prefetch rounds of the chain (klt_kernels.hip FTK_CHAIN_ROUND)

struct DevImage {
    const uint8_t *data;
    int32_t rows;
    int32_t cols;
};

// Kernel argument block of one KLT launch (passed by value: ~450 B of SGPR-loadable constants).
struct KltParams {
    DevImage ref[FTK_MAX_LEVELS];
    DevImage cur[FTK_MAX_LEVELS];
    int32_t n_levels;      // levels actually walked (1 in single-level mode)
    int32_t single_level;  // TrackSingleLevel semantics
    const float *ref_uv;
    const float *cur_uv_in;
    float *cur_uv_out;
    const uint8_t *status_in;
    uint8_t *status_out;
    uint32_t *iters;       // may be null
    const int32_t *order;  // may be null: launch slot -> feature index (a permutation of [0, n) made by an earlier launch's sort block)
    uint32_t *sched_iters; // may be null: iteration counts kept by the context for the launch order of later calls
    // Position-keyed slot swaps (klt_common.h klt_resolve_feature), all null / 0 when off: a coarse grid over the level-0 image in
    // which every feature leaves its iteration count, tagged with the call number; one claim word per launch slot; this call's number
    uint32_t *sched_grid;
    uint32_t *sched_claim;
    uint32_t *sched_flags;  // [2]: (call number << 1 | "the counts that call sorted had no tail"), written by the sort block of a launch
    uint32_t sched_call;
    // the call's longest feature, reported for the next call's wave policy (klt_common.h tail_report; null: not reported)
    uint32_t *tail_dev;
    uint32_t *tail_host;
    uint32_t tail_call;
    int32_t long_tail;          // host-side note: this variant's recent calls had a feature of many iterations (ftk_api.cpp, tail-aware policy)
    const uint32_t *sort_iters;  // may be null: the previous call's counts; one extra workgroup (block 0) sorts them ...
    int32_t *sort_order_out;     // ... into this permutation, longest first (klt_common.h, klt_order_block)
    const float *sort_ref_uv;    // reference positions the sort block may use for the spatial (tile) order: ref_uv, or null when this
                                 // launch's outputs overwrite them (in-place position buffer: the two passes of the sort would disagree)
    int32_t n;             // features in the buffers
    uint32_t n_track;      // min(n, kMaxTrackPointsNumber)
    uint32_t max_iteration;
    uint32_t max_large_step;
    int32_t half_rows, half_cols;
    float converge;
    float prior[4];
    int32_t consider_luminance;
    // derived patch geometry
    int32_t patch_rows, patch_cols, P, Ppad;  // Ppad = P rounded up to a multiple of 16
    int32_t ex_rows, ex_cols, E;
    int32_t a0_floats;  // size of the first LDS array: max(E padded, axis tables)
    uint32_t magic_pc;   // ceil(2^32 / patch_cols): row = umulhi(p, magic_pc)
    uint32_t magic_pc20; // ceil(2^20 / patch_cols) when P * patch_cols < 2^20 (row = (p * magic_pc20) >> 20 on the 24-bit multiplier), else 0
    uint32_t magic_exc;  // ceil(2^32 / ex_cols)
    // LDS image windows (16-bit pixel pairs): reference footprint and current footprint + margin
    int32_t rwin_rows, rwin_cols;
    int32_t cwin_rows, cwin_cols, cwin_margin;
    uint32_t magic_rwc, magic_cwc;  // division by window cols
    uint32_t magic_rwq, magic_cwq;  // division by window cols / 4
    int32_t waves_per_feature;  // workgroup = 64 * waves_per_feature lanes
    int32_t px_floats;          // per-pixel floats behind a0 in the generic kernel's LDS carve: 3, 4 (one float4 record: non-fast affine) or 6 (chunked LSSD)
    int32_t quad_chain;         // chunked one-wave LSSD levels: 1 = the exact-order sums through the DPP network (klt_common.h "quad chain": fewer
                                // instructions on a lone feature's critical path), 0 = one lane per sum (fewer LDS bytes and plain adds: better when
                                // the launch oversubscribes the chip — config 4: 136 against 141 us, with luminance 255 against 285)
    int32_t terms_floats;       // floats of the generic kernel's `terms` LDS region: K * Ppad, or the 64-pixel ring of a chunked variant
    int32_t lssd_chunked;       // LSSD fast, one wave per feature, no luminance scaling: chunked sweep / chain (klt_kernels.hip)
    int32_t features_per_group; // > 1 (only with waves_per_feature == 1): that many one-wave features share a workgroup, without
                                // meeting at a barrier (lifts the 16-workgroups-per-CU cap on resident one-wave features)
    int32_t group_lds_stride;   // bytes between the LDS carves of the features of a group (filled by the launcher)
    // pipelined Basic-KLT inverse kernel (klt_basic_kernels.hip); pb_enabled = 0 selects the generic kernel
    int32_t pb_enabled;
    int32_t pb_rwin_rows, pb_rwin_cols;  // reference window incl. the rounding row / column: 2h+5 (cols padded to 4)
    uint32_t pb_magic_rwc, pb_magic_rwq;
    int32_t pb_cap_r, pb_cap_c;          // lattice node capacity per axis: len + 2 + provable maximum of extras
    int32_t fk_enabled;                  // the one-wave `fast` kernels (klt_fast_kernels.hip); 0 selects the generic kernel
    // Patches whose per-pixel arrays exceed a workgroup's 160 KB of LDS (the reference has no patch-size ceiling: optical_flow.h:24-25
    // takes any int32 half size): the generic kernel with its product rows, extended patch, per-pixel records and flags in a
    // per-workgroup slice of device memory instead (L2-resident at these sizes); sums, counts and — where they fit — the image
    // windows stay in LDS.  Same code, same arithmetic; only the address space of those arrays differs (klt_kernels.hip SPILL).
    int32_t spill;
    float *spill_base;                   // spill_stride_floats floats per workgroup (launch slot)
    uint32_t spill_stride_floats;
    int32_t tree;                        // 0: sums in the reference's order (the contract); 1: throughput mode — the same per-pixel products, summed
                                         // by per-lane partials + a cross-lane butterfly (ftk_set_reduction_mode; reported, never the default)
    unsigned long long *stamps; // diagnostic build (-DFTK_STAMPS) only: 8 cycle totals per feature; else null
};

// ceil(2^32 / d): row = umulhi(index, magic)
__host__ __device__ constexpr uint32_t klt_div_magic(int32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); }

// Everything in KltParams that follows from (half_rows, half_cols) ALONE.  One definition for the host (fill_klt_params) and
// for the kernels' compile-time specialisations (klt_basic_kernels.hip instantiates the pipelined kernel for the common patch
// sizes: the geometry then folds into immediates instead of living in ~40 SGPRs, most of them spilled to vector lanes).
__host__ __device__ constexpr void klt_fill_geometry(KltParams &p) {
    p.patch_rows = 2 * p.half_rows + 1;
    p.patch_cols = 2 * p.half_cols + 1;
    p.P = p.patch_rows * p.patch_cols;
    p.Ppad = (p.P + 15) & ~15;  // (a multiple of 16 since round 5: the quad chains add 16 terms per step, klt_common.h; the padding holds exact zeros)
    p.ex_rows = p.patch_rows + 2;
    p.ex_cols = p.patch_cols + 2;
    p.E = p.ex_rows * p.ex_cols;
    const int32_t epad = (p.E + 3) & ~3, tables = 12 * (p.patch_rows + p.patch_cols);
    p.a0_floats = epad > tables ? epad : tables;
    p.magic_pc = klt_div_magic(p.patch_cols);
    // the same division on the full-rate 24-bit multiplier when it is exact over the whole patch: x * (M * d - 2^20) < 2^20 for all x < P
    p.magic_pc20 = (p.patch_cols > 1 && (long long)p.P * p.patch_cols < (1ll << 20) && p.P < (1 << 12))
                       ? (uint32_t)(((1u << 20) + (uint32_t)p.patch_cols - 1) / (uint32_t)p.patch_cols) : 0u;
    p.magic_exc = klt_div_magic(p.ex_cols);
    p.rwin_rows = p.patch_rows + 3;
    p.rwin_cols = (p.patch_cols + 3 + 3) & ~3;  // pixel-pair columns, rounded up to a multiple of 4 (8-byte LDS stores)
#ifndef FTK_CWIN_MARGIN
#define FTK_CWIN_MARGIN 2
#endif
    p.cwin_margin = FTK_CWIN_MARGIN;
    p.cwin_rows = p.rwin_rows + 2 * p.cwin_margin;
    p.cwin_cols = (p.patch_cols + 3 + 2 * p.cwin_margin + 3) & ~3;
    p.magic_rwc = klt_div_magic(p.rwin_cols);
    p.magic_cwc = klt_div_magic(p.cwin_cols);
    p.magic_rwq = klt_div_magic(p.rwin_cols / 4);
    p.magic_cwq = klt_div_magic(p.cwin_cols / 4);
    p.pb_rwin_rows = p.patch_rows + 4;
    p.pb_rwin_cols = (p.patch_cols + 4 + 3) & ~3;
    p.pb_magic_rwc = klt_div_magic(p.pb_rwin_cols);
    p.pb_magic_rwq = klt_div_magic(p.pb_rwin_cols / 4);
    // lattice extras per axis: a unit step crosses at most log2(len + 2) + 3 binade boundaries, two nodes each
    int bits_r = 0, bits_c = 0;
    while ((1 << bits_r) < p.patch_rows + 2) {
        ++bits_r;
    }
    while ((1 << bits_c) < p.patch_cols + 2) {
        ++bits_c;
    }
    p.pb_cap_r = p.patch_rows + 2 + 2 * (bits_r + 3);
    p.pb_cap_c = p.patch_cols + 2 + 2 * (bits_c + 3);
}

// LDS bytes a (model, method) variant needs for the given geometry; 0 if the variant is unknown.
size_t klt_lds_bytes(int model, int method, const KltParams &p);
// Floats of device memory one workgroup of the large-patch (p.spill) form needs; 0 if the variant is unknown.
size_t klt_spill_floats(int model, const KltParams &p);
// Launches the tracker kernel for (model, method) on `stream`; one workgroup of waves_per_feature wavefronts per feature.
hipError_t klt_launch(int model, int method, const KltParams &p, hipStream_t stream);
// Launch order of THIS call from the position table the last call wrote (klt_kernels.hip): order[slot] = feature, predicted-longest
// first.  last_table: 2^16 words; pred: n bytes; hist_and_cursor: 512 words (zeroed here).
hipError_t klt_position_order_launch(const float *ref_uv, int32_t n, const uint32_t *last_table, uint32_t last_call, uint8_t *pred, uint32_t *hist_and_cursor,
                                     int32_t *order, hipStream_t stream);
// Lane-parallel 6x6 LDLT (klt_common.h) on n systems, one wave each: the test hook behind ftk_ldlt6_solve.
hipError_t ldlt6_launch(const float *d_a, const float *d_b, float *d_x, int n, hipStream_t stream);
// Pipelined kernel for (FTK_MODEL_BASIC, FTK_METHOD_INVERSE); klt_launch dispatches to it when p.pb_enabled.
size_t klt_basic_pipelined_lds_bytes(const KltParams &p);
hipError_t klt_basic_pipelined_launch(const KltParams &p, hipStream_t stream);

// Reference descriptors a thread of the register-tiled Hamming scan keeps in registers (a 256-thread workgroup covers
// 256 * kMatchRefs reference rows); the host sizes its grid and its NearbyMatch boxes with the same number.
#ifndef FTK_MATCH_REFS
#define FTK_MATCH_REFS 2
#endif
constexpr int kMatchRefs = FTK_MATCH_REFS;
constexpr int kMatchRowsPerBlock = 256 * kMatchRefs;

struct MatchParams {
    const uint32_t *ref_words;
    const uint32_t *cur_words;
    const float *pred_uv;  // null => ForceMatch
    const float *cur_uv;
    int32_t *index_pairs;
    unsigned long long *keys;  // workspace: n_ref packed (distance << 32 | index)
    int32_t n_ref, n_cur, n_words, n_bits;
    float max_distance;
    float max_col, max_row;
    int32_t cur_per_block;  // candidates scanned by one workgroup
    int32_t keys_clean;     // keys already hold "no match" (context-owned workspace: the epilogue leaves it that way)
    unsigned long long *stamps;  // diagnostic build (-DFTK_MATCH_STAMPS) only: {start, end} s_memrealtime + HW_ID per workgroup; else null
    int32_t matrix_cores;   // 1: the scan on the matrix cores (hamming_match_mfma_kernel; n_words 8 / 16)
    int32_t lds_tiles;      // experiment (FTK_MATCH_KERNEL=lds): candidates staged through LDS tiles instead of the scalar path
    float4 *boxes;          // NearbyMatch, optional: ceil(n_ref / 512) prediction boxes, then one candidate box per split
                            // ({u min, u max, v min, v max}; hamming_box_kernel fills them, the scan leaves early on them)
    int32_t small_off;      // experiment (FTK_MATCH_SMALL=0, read once per context): never the one-launch form
};
hipError_t match_launch(const MatchParams &p, hipStream_t stream);
// Whether match_launch runs a call of this shape as ONE launch without the keys workspace or the NearbyMatch boxes (small calls).
bool match_small_form(int n_ref, int n_cur, int n_words, int n_bits, bool small_off);

// Float-descriptor (cosine distance) matcher: float_matcher_kernels.hip.
constexpr int kCosineCandCap = 64;       // candidates kept per ref row before the row falls back to the exact scan
constexpr int kCosineIrregularCap = 64;  // cur rows with a zero / non-finite / extreme norm kept in the side list
struct CosineParams {
    const float *ref, *cur;   // [n][dim] fp32 descriptors, row-major
    const float *pred_uv;     // null => ForceMatch
    const float *cur_uv;
    int32_t *index_pairs;
    // workspace (see ftk_cosine_workspace_bytes)
    _Float16 *ref_h, *cur_h;  // unit-length fp16 copies, [n_pad][dim_pad], zero padded
    float *ref_norm, *cur_norm, *cur_bias;
    float4 *cur_info;         // {bias, u, v, 0} per (padded) cur row: what cosine_gemm_rr_kernel streams beside the tile
    float4 *tile_box;         // NearbyMatch: {u min, u max, v min, v max} of every 64-row cur tile (cosine_tile_box_kernel); null: not used
    uint8_t *ref_irregular;
    uint32_t *row_max, *cand_count;
    int32_t *cand;            // [n_ref_pad][kCosineCandCap]
    float *cand_score;        // approximate score of each entry; non-null selects the single-walk contraction (ref-stationary only)
    uint32_t *irregular_count;
    int32_t *irregular_list;  // [kCosineIrregularCap]
    void *clear_begin;        // row_max | cand_count | irregular_count, contiguous: zeroed by one memset per call
    size_t clear_bytes;
    int32_t n_ref, n_cur, dim, n_ref_pad, n_cur_pad, dim_pad;
    int32_t tiles_per_split;  // cur tiles (128 rows; 256 in ref-stationary mode) walked by one workgroup
    int32_t ref_stationary;   // dim_pad <= 256.  2: cosine_gemm_rr_kernel (ref fragments in registers; n_ref_pad % 512 == 0,
                              // n_cur_pad % 64 == 0, `splits` workgroups share the cur tiles evenly);
                              // 1: cosine_gemm_rs_kernel (ref rows in LDS; n_cur_pad % 256 == 0, tiles_per_split)
    int32_t splits;
    float max_distance, max_col, max_row;
    int32_t small_off, small_any;  // experiments (FTK_COSINE_SMALL=0 / FTK_COSINE_SMALL_ANY=1, read once per context)
};
size_t cosine_rs_lds_bytes(int dim_pad);
size_t cosine_rr_lds_bytes(int dim_pad);
hipError_t cosine_match_launch(const CosineParams &p, hipStream_t stream);
// Whether cosine_match_launch runs a call of this shape as one exact launch without the workspace (small calls).
bool cosine_small_form(int n_ref, int n_cur, int dim, bool nearby, bool small_off, bool small_any);

// DirectMethod (direct_kernels.hip): one workgroup per pose problem; all problems of a launch share
// the pyramid depth and the options.
struct DirectProblem {
    DevImage ref[FTK_MAX_LEVELS];
    DevImage cur[FTK_MAX_LEVELS];
    float K[4];             // fx, fy, cx, cy at full resolution
    const float *p_ref;     // n x 3: points in the reference camera frame
    const float *ref_uv;    // n x 2
    float *cur_uv;          // n x 2, in/out
    float *pose;            // 7: q_rc (w, x, y, z), p_rc — in/out
    uint8_t *status;        // n, in/out
    uint32_t *iterations;   // optional: Gauss-Newton iterations over all levels
    int32_t n;
    int32_t status_valid;   // 0: reset every status to kTracked first (direct_method_tracker.cpp:73-75)
    float4 *feat;           // null: the per-feature projection table lives in LDS; else n_track entries of device memory (large problems)
};
#if defined(FTK_DM_WAVES) && FTK_DM_WAVES > 8
constexpr uint32_t kDirectLdsFeatures = 384;  // (experiment builds with more producer waves: a larger ring)
#else
constexpr uint32_t kDirectLdsFeatures = 768;  // tracked features whose per-feature table (64 B each) still fits in LDS beside the ring
#endif
struct DirectParams {
    const DirectProblem *problems;  // device memory, one per workgroup
    int32_t tree;                   // throughput mode (ftk_set_reduction_mode): butterfly sums instead of the scalar loop's order
    int32_t n_levels;
    uint32_t max_track_points, max_iteration;
    int32_t half_rows, half_cols, patch_rows, patch_cols;
    float converge;
    int32_t method;
    // ONE problem spread over the chip (direct_kernels.hip direct_track_spread_kernel): `spread` producer workgroups beside the
    // consumer, hand-offs through `spread_ws` (direct_spread_ws_bytes; its first direct_spread_clear_bytes zeroed before the launch)
    int32_t spread;
    uint32_t *spread_ws;
    uint32_t spread_ws_words;  // words between the workspaces of consecutive problems
    int32_t spread_poison;     // tests only (FTK_DIRECT_SPREAD_POISON=1): the consumer behaves as if its first wait had run out — the
                               // launch ends at once with header word 1 set and a NaN pose, as a launch that was not co-resident would
};
size_t direct_lds_bytes(uint32_t max_features);
size_t direct_spread_ws_bytes(uint32_t n_track, int32_t patch_rows, int32_t patch_cols);
size_t direct_spread_clear_bytes(uint32_t n_track, int32_t patch_rows, int32_t patch_cols);
int direct_spread_resident_groups(uint32_t max_features, int device);  // workgroups of the spread kernel the device holds at once (0: unknown)
hipError_t direct_track_launch(const DirectParams &p, int n_problems, uint32_t max_features, hipStream_t stream);

struct BriefParams {
    DevImage img;
    const float *uv;
    uint32_t *words;        // n * n_words, bit i of a descriptor in bit (i % 32) of word i / 32
    const int8_t *pattern;  // 4 offsets per bit, device memory
    int32_t n, n_bits, n_words, half;
};
hipError_t brief_launch(const BriefParams &p, hipStream_t stream);

struct HarrisParams {
    DevImage img;
    short *gx, *gy;                  // rows*cols each
    float *response;                 // rows*cols or null
    unsigned long long *key, *tmp, *wmax;  // rows*cols each (tmp / wmax unused when list is null)
    unsigned long long *list;        // survivors (keys), capacity entries; null = response only
    unsigned *count;
    unsigned capacity;
    int32_t min_distance;
    float min_response;
};
hipError_t harris_launch(const HarrisParams &p, hipStream_t stream);

// Scatter of the all-gathered packed tracker shards into (cur_uv, status) in global feature order (ftk_comm.cpp).
hipError_t unpack_klt_shards_launch(const uint8_t *d_gathered, int32_t n, int32_t world, int32_t cap, int64_t shard_bytes, float *d_uv_out,
                                    uint8_t *d_status_out, hipStream_t stream);
hipError_t pyramid_downsample_launch(const uint8_t *src, int32_t src_rows, int32_t src_cols, uint8_t *dst, hipStream_t stream);
// Levels 1 .. n_levels - 1 from level 0 (dst[l] = level l, l >= 1): one fused launch (pyramid_fused_kernel), deeper levels one by one.
// level0_keep (needs pyramid_fused_enabled() and n_levels >= 2): `level0` is a device-visible source outside the pyramid (pinned host
// memory) and the same launch stores it as the pyramid's level 0.
hipError_t pyramid_build_levels_launch(const uint8_t *level0, int32_t rows, int32_t cols, uint8_t *const *dst, int32_t n_levels, hipStream_t stream,
                                       uint8_t *level0_keep = nullptr);
bool pyramid_fused_enabled();
hipError_t extract_patch_launch(DevImage ref, float u, float v, int32_t ex_rows, int32_t ex_cols, float *d_patch, uint8_t *d_valid,
                                uint32_t *d_count, hipStream_t stream);

// One-wave kernels of the `fast` method (klt_fast_kernels.hip); klt_launch dispatches to them when p.fk_enabled.
size_t klt_fast_lds_bytes(int model, const KltParams &p);
hipError_t klt_fast_launch(int model, const KltParams &p, hipStream_t stream);
hipError_t klt_fast_warm(hipStream_t stream);

// One empty launch per translation unit: loads its code object (ftk_warmup).
hipError_t klt_warm(hipStream_t stream);
hipError_t klt_basic_warm(hipStream_t stream);
hipError_t matcher_warm(hipStream_t stream);
hipError_t cosine_warm(hipStream_t stream);
hipError_t direct_warm(hipStream_t stream);
hipError_t pyramid_warm(hipStream_t stream);
hipError_t feature_warm(hipStream_t stream);

}  // namespace ftk
