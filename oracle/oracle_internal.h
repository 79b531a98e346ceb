/* oracle_internal.h — helpers shared by the oracle's translation units (test infrastructure only). */
#ifndef FTK_ORACLE_INTERNAL_H_
#define FTK_ORACLE_INTERNAL_H_

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ftk_oracle.h"

/* float -> int32 with the x86-64 cvttss2si result the reference gets from static_cast<int32_t>
 * (out of range / NaN -> INT32_MIN); in-range values truncate toward zero. */
static inline int32_t orc_f2i(float x) {
    if (x >= -2147483648.0f && x < 2147483648.0f) {
        return (int32_t)x;
    }
    return INT32_MIN;
}

/* two's-complement wrapping int32 add (the reference's int arithmetic on x86 wraps in practice) */
static inline int32_t orc_wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

static inline uint8_t orc_px(const orc_image *img, int32_t row, int32_t col) { return img->data[(int64_t)row * img->cols + col]; }

/* floor() for values inside the int32 range without a libm call (the build has no SSE4.1) */
static inline float orc_floor_from_trunc(float x, int32_t truncated) {
    const float t = (float)truncated;
    return (t > x) ? t - 1.0f : t;
}

/* Bilinear sample.  Call sites: basic_klt.cpp:132-134, affine_klt.cpp:150-152,
 * lssd_klt.cpp:202-207, lssd_klt_fast.cpp:189.  Weight products and the summation order are
 * those of the reference's explicit formula, optical_flow.cpp:53-60,78-81. */
static inline float orc_bilinear(const orc_image *img, float row, float col) {
    int32_t r0 = orc_f2i(row);
    int32_t c0 = orc_f2i(col);
    const float sub_row = row - orc_floor_from_trunc(row, r0);
    const float sub_col = col - orc_floor_from_trunc(col, c0);
    /* memory safety only: every in-contract call has 0 <= row <= rows-1, 0 <= col <= cols-1 */
    r0 = r0 < 0 ? 0 : (r0 > img->rows - 1 ? img->rows - 1 : r0);
    c0 = c0 < 0 ? 0 : (c0 > img->cols - 1 ? img->cols - 1 : c0);
    const int32_t r1 = (r0 + 1 < img->rows) ? r0 + 1 : r0;
    const int32_t c1 = (c0 + 1 < img->cols) ? c0 + 1 : c0;
    const float inv_sub_row = 1.0f - sub_row;
    const float inv_sub_col = 1.0f - sub_col;
    const float w_tl = inv_sub_row * inv_sub_col;
    const float w_tr = inv_sub_row * sub_col;
    const float w_bl = sub_row * inv_sub_col;
    const float w_br = sub_row * sub_col;
    return w_tl * (float)orc_px(img, r0, c0) + w_tr * (float)orc_px(img, r0, c1) + w_bl * (float)orc_px(img, r1, c0) +
           w_br * (float)orc_px(img, r1, c1);
}

/* Bounds-checked bilinear sample; "inside" is the closed rectangle the reference itself uses
 * for features (basic_klt.cpp:107).  NaN coordinates are invalid. */
static inline int orc_sample(const orc_image *img, float row, float col, float *value) {
    if (!(row >= 0.0f && col >= 0.0f && row <= (float)(img->rows - 1) && col <= (float)(img->cols - 1))) {
        return 0;
    }
    *value = orc_bilinear(img, row, col);
    return 1;
}

/* per-model single-feature drivers (one pyramid level / one image) */
void orc_basic_track_one(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                         uint8_t *status, uint32_t *iters);
void orc_basic_track_one_fast(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                              uint8_t *status, uint32_t *iters);
void orc_affine_track_one(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                          float *affine, uint8_t *status, uint32_t *iters);
void orc_affine_track_one_fast(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                               float *affine, uint8_t *status, uint32_t *iters);
void orc_lssd_track_one(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *r_cr,
                        float *t_cr, uint8_t *status, uint32_t *iters);
void orc_lssd_track_one_fast(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *r_cr,
                             float *t_cr, int consider_luminance, uint8_t *status, uint32_t *iters);

/* shared by the three "fast" variants: central differences on the extended patch
 * (basic_klt_fast.cpp:64-99, affine_klt_fast.cpp:71-138, lssd_klt_fast.cpp:116-143). */
static inline int orc_ex_neighbours_valid(const uint8_t *valid, int32_t ex_index, int32_t ex_cols) {
    return valid[ex_index - 1] && valid[ex_index + 1] && valid[ex_index - ex_cols] && valid[ex_index + ex_cols];
}

#endif /* FTK_ORACLE_INTERNAL_H_ */
