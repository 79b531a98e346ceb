/*
 * oracle_basic_klt.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * Translation-only KLT, restated from
 *   src/optical_flow_tracker/basic_klt/optical_flow_basic_klt.cpp       (inverse, direct)
 *   src/optical_flow_tracker/basic_klt/optical_flow_basic_klt_fast.cpp  (fast)
 */
#include "oracle_internal.h"

/* ConstructIncrementalFunction, basic_klt.cpp:118-181.  h = [h00, h01, h11], b = [b0, b1]. */
static int32_t basic_build_normal_equations(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v,
                                            float cur_u, float cur_v, float *h, float *b) {
    const int direct = (opt->method != ORC_INVERSE);
    /* gradient image: ref for inverse (:132-134), cur for direct (:159-161) */
    const orc_image *grad_img = direct ? cur : ref;
    int32_t n_valid = 0;
    for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
        for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
            /* bilinear fractions are taken per fetch from the summed coordinate (:127-130) */
            const float row_i = (float)drow + ref_v;
            const float col_i = (float)dcol + ref_u;
            const float row_j = (float)drow + cur_v;
            const float col_j = (float)dcol + cur_u;
            const float grow = direct ? row_j : row_i;
            const float gcol = direct ? col_j : col_i;
            float left, right, top, bottom, i_ref, i_cur;
            if (orc_sample(grad_img, grow, gcol - 1.0f, &left) && orc_sample(grad_img, grow, gcol + 1.0f, &right) &&
                orc_sample(grad_img, grow - 1.0f, gcol, &top) && orc_sample(grad_img, grow + 1.0f, gcol, &bottom) &&
                orc_sample(ref, row_i, col_i, &i_ref) && orc_sample(cur, row_j, col_j, &i_cur)) {
                /* no 1/2 factor on the central difference (:135-137) */
                const float fx = right - left;
                const float fy = bottom - top;
                const float ft = i_cur - i_ref;
                h[0] += fx * fx;
                h[2] += fy * fy;
                h[1] += fx * fy;
                b[0] -= fx * ft;
                b[1] -= fy * ft;
                ++n_valid;
            }
        }
    }
    return n_valid;
}

/* TrackOneFeature, basic_klt.cpp:88-116.  Status is left untouched when the loop runs out
 * of iterations or a patch has no valid pixel. */
void orc_basic_track_one(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                         uint8_t *status, uint32_t *iters) {
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        float h[3] = {0.0f, 0.0f, 0.0f};
        float b[2] = {0.0f, 0.0f};
        ++*iters;
        if (basic_build_normal_equations(opt, ref, cur, ref_u, ref_v, cur_uv[0], cur_uv[1], h, b) == 0) {
            break;
        }
        const float hm[4] = {h[0], h[1], h[1], h[2]};
        float v[2];
        orc_ldlt_solve(2, hm, b, v);
        if (isnan(v[0]) || isnan(v[1])) {
            *status = ORC_NUMERIC_ERROR;
            break;
        }
        cur_uv[0] += v[0];
        cur_uv[1] += v[1];
        if (cur_uv[0] < 0.0f || cur_uv[0] > (float)(cur->cols - 1) || cur_uv[1] < 0.0f || cur_uv[1] > (float)(cur->rows - 1)) {
            *status = ORC_OUTSIDE;
            break;
        }
        if (v[0] * v[0] + v[1] * v[1] < opt->max_converge_step) {
            *status = ORC_TRACKED;
            break;
        }
    }
}

/* ComputeBias, basic_klt_fast.cpp:101-195: shared-weight bilinear of cur on the integer
 * lattice floor(cur) - patch/2; pixels invalid in cur (:132) or in the extended ref patch
 * (:138,171) are skipped. */
static uint32_t basic_fast_bias(const orc_image *cur, float cur_u, float cur_v, const float *ex_patch, const uint8_t *ex_valid, int32_t ex_rows,
                                int32_t ex_cols, const float *dx, const float *dy, float *b) {
    const int32_t patch_rows = ex_rows - 2;
    const int32_t patch_cols = ex_cols - 2;
    b[0] = 0.0f;
    b[1] = 0.0f;

    const float int_row = floorf(cur_v);
    const float int_col = floorf(cur_u);
    const float dec_row = cur_v - int_row;
    const float dec_col = cur_u - int_col;
    const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
    const float w_tr = (1.0f - dec_row) * dec_col;
    const float w_bl = dec_row * (1.0f - dec_col);
    const float w_br = dec_row * dec_col;

    const int32_t min_row = orc_wadd(orc_f2i(int_row), -(patch_rows / 2));
    const int32_t min_col = orc_wadd(orc_f2i(int_col), -(patch_cols / 2));
    const int32_t max_row = orc_wadd(min_row, patch_rows);
    const int32_t max_col = orc_wadd(min_col, patch_cols);

    uint32_t n_valid = 0;
    for (int32_t row = min_row; row < max_row; ++row) {
        const int32_t row_in_patch = row - min_row;
        for (int32_t col = min_col; col < max_col; ++col) {
            if (row < 0 || row > cur->rows - 2 || col < 0 || col > cur->cols - 2) {
                continue;
            }
            const int32_t col_in_patch = col - min_col;
            const int32_t ex_index = (row_in_patch + 1) * ex_cols + col_in_patch + 1;
            if (!ex_valid[ex_index]) {
                continue;
            }
            const float i_cur = w_tl * (float)orc_px(cur, row, col) + w_tr * (float)orc_px(cur, row, col + 1) +
                                w_bl * (float)orc_px(cur, row + 1, col) + w_br * (float)orc_px(cur, row + 1, col + 1);
            const float dt = i_cur - ex_patch[ex_index];
            const int32_t index = row_in_patch * patch_cols + col_in_patch;
            b[0] -= dx[index] * dt;
            b[1] -= dy[index] * dt;
            ++n_valid;
        }
    }
    return n_valid;
}

/* TrackOneFeatureFast, basic_klt_fast.cpp:7-62 (+ PrecomputeJacobianAndHessian :64-99). */
void orc_basic_track_one_fast(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                              uint8_t *status, uint32_t *iters) {
    const int32_t patch_rows = 2 * opt->half_rows + 1, patch_cols = 2 * opt->half_cols + 1;
    const int32_t ex_rows = patch_rows + 2, ex_cols = patch_cols + 2;
    float *ex_patch = (float *)malloc(sizeof(float) * ex_rows * ex_cols);
    uint8_t *ex_valid = (uint8_t *)malloc((size_t)ex_rows * ex_cols);
    float *dx = (float *)malloc(sizeof(float) * patch_rows * patch_cols);
    float *dy = (float *)malloc(sizeof(float) * patch_rows * patch_cols);
    float h[3] = {0.0f, 0.0f, 0.0f};
    float hm[4];
    float last_squared_step = INFINITY;
    uint32_t large_step_cnt = 0;

    if (orc_extract_extend_patch(ref, ref_u, ref_v, ex_rows, ex_cols, ex_patch, ex_valid) == 0) {
        *status = ORC_OUTSIDE; /* :16-19 */
        goto done;
    }

    /* dx, dy and the fixed Hessian (:64-99): pixels with an invalid 4-neighbour get dx = dy = 0 */
    for (int32_t row = 0; row < patch_rows; ++row) {
        for (int32_t col = 0; col < patch_cols; ++col) {
            const int32_t ex_index = (row + 1) * ex_cols + col + 1;
            const int32_t index = row * patch_cols + col;
            if (orc_ex_neighbours_valid(ex_valid, ex_index, ex_cols)) {
                const float gx = ex_patch[ex_index + 1] - ex_patch[ex_index - 1];
                const float gy = ex_patch[ex_index + ex_cols] - ex_patch[ex_index - ex_cols];
                dx[index] = gx;
                dy[index] = gy;
                h[0] += gx * gx;
                h[1] += gx * gy;
                h[2] += gy * gy;
            } else {
                dx[index] = 0.0f;
                dy[index] = 0.0f;
            }
        }
    }
    hm[0] = h[0];
    hm[1] = h[1];
    hm[2] = h[1];
    hm[3] = h[2];

    *status = ORC_LARGE_RESIDUAL; /* :29 */
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        float b[2];
        ++*iters;
        if (basic_fast_bias(cur, cur_uv[0], cur_uv[1], ex_patch, ex_valid, ex_rows, ex_cols, dx, dy, b) == 0) {
            break;
        }
        float v[2];
        orc_ldlt_solve(2, hm, b, v);
        if (isnan(v[0]) || isnan(v[1])) {
            *status = ORC_NUMERIC_ERROR;
            break;
        }
        cur_uv[0] += v[0];
        cur_uv[1] += v[1];
        const float squared_step = v[0] * v[0] + v[1] * v[1];
        if (squared_step < last_squared_step) {
            last_squared_step = squared_step;
            large_step_cnt = 0;
        } else {
            ++large_step_cnt;
            if (large_step_cnt >= opt->max_tolerance_large_step) {
                break;
            }
        }
        if (squared_step < opt->max_converge_step) {
            *status = ORC_TRACKED;
            break;
        }
    }

done:
    free(ex_patch);
    free(ex_valid);
    free(dx);
    free(dy);
}
