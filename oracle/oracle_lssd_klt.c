/*
 * oracle_lssd_klt.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * SE(2) KLT with locally scaled SSD, restated from
 *   src/optical_flow_tracker/lssd_klt/optical_flow_lssd_klt.cpp       (inverse, direct)
 *   src/optical_flow_tracker/lssd_klt/optical_flow_lssd_klt_fast.cpp  (fast)
 * r[] is row-major [r00, r01, r10, r11]; t[] = [tx, ty].
 */
#include "oracle_internal.h"

/* Vec3::squaredNorm().  Eigen reduces a fixed-size, non-vectorisable 3-vector with its
 * unrolled binary split: a0 + (a1 + a2). */
static inline float vec3_squared_norm(const float *v) { return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]); }

/* R_cr * Vec2(x, y) + t_cr */
static inline void se2_apply(const float *r, const float *t, float x, float y, float *out_x, float *out_y) {
    *out_x = (r[0] * x + r[1] * y) + t[0];
    *out_y = (r[2] * x + r[3] * y) + t[1];
}

/* delta_R << 1, -theta, theta, 1;  R *= delta_R;  R /= R.col(0).norm();  t += v.tail<2>()
 * (lssd_klt.cpp:114-117, lssd_klt_fast.cpp:95-98). */
static void se2_update(float *r, float *t, const float *v) {
    const float theta = v[0];
    const float d00 = 1.0f, d01 = -theta, d10 = theta, d11 = 1.0f;
    const float n00 = r[0] * d00 + r[1] * d10;
    const float n01 = r[0] * d01 + r[1] * d11;
    const float n10 = r[2] * d00 + r[3] * d10;
    const float n11 = r[2] * d01 + r[3] * d11;
    const float norm = sqrtf(n00 * n00 + n10 * n10);
    r[0] = n00 / norm;
    r[1] = n01 / norm;
    r[2] = n10 / norm;
    r[3] = n11 / norm;
    t[0] += v[1];
    t[1] += v[2];
}

/* hessian += j^T j ; bias -= j^T residual  (lssd_klt.cpp:214-215, lssd_klt_fast.cpp:220-221) */
static inline void accumulate3(float *h, float *b, const float *j, float residual) {
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 3; ++k) {
            h[i * 3 + k] += j[i] * j[k];
        }
        b[i] -= j[i] * residual;
    }
}

/* ConstructIncrementalFunction, lssd_klt.cpp:127-250: pass 1 = validity mask and patch means
 * (:140-184), pass 2 = mean-normalised Jacobian and residual (:186-247).  Always
 * mean-normalises (consider_patch_luminance_ is not consulted here). */
static int32_t lssd_build_normal_equations(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v,
                                           const float *r, const float *t, float *h, float *b) {
    const int direct = (opt->method != ORC_INVERSE);
    const orc_image *grad_img = direct ? cur : ref;
    const int32_t patch_cols = 2 * opt->half_cols + 1;
    const int32_t patch_size = (2 * opt->half_rows + 1) * patch_cols;
    uint8_t *pixel_valid = (uint8_t *)malloc((size_t)patch_size);
    int32_t n_valid = 0;
    float ref_average = 0.0f, cur_average = 0.0f;

    for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
        for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
            const float row_i = (float)drow + ref_v;
            const float col_i = (float)dcol + ref_u;
            float row_j, col_j;
            se2_apply(r, t, col_i, row_i, &col_j, &row_j);
            const float grow = direct ? row_j : row_i;
            const float gcol = direct ? col_j : col_i;
            float left, right, top, bottom, i_ref, i_cur;
            const int32_t index = (drow + opt->half_rows) * patch_cols + dcol + opt->half_cols;
            if (orc_sample(grad_img, grow, gcol - 1.0f, &left) && orc_sample(grad_img, grow, gcol + 1.0f, &right) &&
                orc_sample(grad_img, grow - 1.0f, gcol, &top) && orc_sample(grad_img, grow + 1.0f, gcol, &bottom) &&
                orc_sample(ref, row_i, col_i, &i_ref) && orc_sample(cur, row_j, col_j, &i_cur)) {
                ref_average += i_ref;
                cur_average += i_cur;
                ++n_valid;
                pixel_valid[index] = 1;
            } else {
                pixel_valid[index] = 0;
            }
        }
    }
    ref_average /= (float)n_valid;
    cur_average /= (float)n_valid;
    const float grad_average = direct ? cur_average : ref_average; /* :209 vs :237 */

    for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
        for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
            const int32_t index = (drow + opt->half_rows) * patch_cols + dcol + opt->half_cols;
            if (!pixel_valid[index]) {
                continue;
            }
            const float row_i = (float)drow + ref_v;
            const float col_i = (float)dcol + ref_u;
            float row_j, col_j;
            se2_apply(r, t, col_i, row_i, &col_j, &row_j);
            const float grow = direct ? row_j : row_i;
            const float gcol = direct ? col_j : col_i;
            const float left = orc_bilinear(grad_img, grow, gcol - 1.0f);
            const float right = orc_bilinear(grad_img, grow, gcol + 1.0f);
            const float top = orc_bilinear(grad_img, grow - 1.0f, gcol);
            const float bottom = orc_bilinear(grad_img, grow + 1.0f, gcol);
            const float i_ref = orc_bilinear(ref, row_i, col_i);
            const float i_cur = orc_bilinear(cur, row_j, col_j);

            /* jacobian_pixel (1x2) / average; jacobian_se2 = [R*(-row_i, col_i) | I2]; j = jp * jse2 */
            const float jp0 = (right - left) / grad_average;
            const float jp1 = (bottom - top) / grad_average;
            const float s0 = r[0] * (-row_i) + r[1] * col_i;
            const float s1 = r[2] * (-row_i) + r[3] * col_i;
            float j[3];
            j[0] = jp0 * s0 + jp1 * s1;
            j[1] = jp0 * 1.0f + jp1 * 0.0f;
            j[2] = jp0 * 0.0f + jp1 * 1.0f;
            const float residual = i_cur / cur_average - i_ref / ref_average;
            accumulate3(h, b, j, residual);
        }
    }
    free(pixel_valid);
    return n_valid;
}

/* TrackOneFeature, lssd_klt.cpp:96-125.  No outside test inside the loop. */
void orc_lssd_track_one(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *r, float *t,
                        uint8_t *status, uint32_t *iters) {
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        float h[9] = {0}, b[3] = {0}, v[3];
        ++*iters;
        if (lssd_build_normal_equations(opt, ref, cur, ref_u, ref_v, r, t, h, b) == 0) {
            break;
        }
        orc_ldlt_solve(3, h, b, v);
        if (isnan(v[0]) || isnan(v[1]) || isnan(v[2])) {
            *status = ORC_NUMERIC_ERROR;
            break;
        }
        se2_update(r, t, v);
        if (vec3_squared_norm(v) < opt->max_converge_step) {
            *status = ORC_TRACKED;
            break;
        }
    }
}

/* ExtractPatchInCurrentImage, lssd_klt_fast.cpp:145-195.  The "totally inside" test uses
 * truncation (static_cast<int32_t>) and a +-(2h+1) window (:150-156). */
static uint32_t lssd_extract_cur_patch(const orc_klt_options *opt, const orc_image *cur, float ref_u, float ref_v, const float *r, const float *t,
                                       int32_t patch_rows, int32_t patch_cols, float *cur_patch, uint8_t *cur_valid) {
    float centre_u, centre_v;
    se2_apply(r, t, ref_u, ref_v, &centre_u, &centre_v);
    const int32_t min_row = orc_wadd(orc_f2i(centre_v), -patch_rows);
    const int32_t min_col = orc_wadd(orc_f2i(centre_u), -patch_cols);
    const int32_t max_row = orc_wadd(min_row, patch_rows * 2);
    const int32_t max_col = orc_wadd(min_col, patch_cols * 2);
    const int partly_outside = (min_row < 0 || max_row > cur->rows - 2 || min_col < 0 || max_col > cur->cols - 2);

    uint32_t n_valid = 0;
    int32_t index = 0;
    for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
        for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol, ++index) {
            const float row_i = (float)drow + ref_v;
            const float col_i = (float)dcol + ref_u;
            float row_j, col_j;
            se2_apply(r, t, col_i, row_i, &col_j, &row_j);
            if (partly_outside) {
                float value = 0.0f;
                if (orc_sample(cur, row_j, col_j, &value)) {
                    cur_patch[index] = value;
                    cur_valid[index] = 1;
                    ++n_valid;
                } else {
                    cur_patch[index] = 0.0f;
                    cur_valid[index] = 0;
                }
            } else {
                cur_patch[index] = orc_bilinear(cur, row_j, col_j);
                cur_valid[index] = 1;
                ++n_valid;
            }
        }
    }
    return n_valid;
}

/* TrackOneFeatureFast, lssd_klt_fast.cpp:7-114 (+ PrecomputeJacobian :116-143,
 * ComputeHessianAndBias :197-229). */
void orc_lssd_track_one_fast(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *r, float *t,
                             int consider_luminance, uint8_t *status, uint32_t *iters) {
    const int32_t patch_rows = 2 * opt->half_rows + 1, patch_cols = 2 * opt->half_cols + 1;
    const int32_t patch_size = patch_rows * patch_cols;
    const int32_t ex_rows = patch_rows + 2, ex_cols = patch_cols + 2;
    const int32_t ex_size = ex_rows * ex_cols;
    float *ex_patch = (float *)malloc(sizeof(float) * ex_size);
    uint8_t *ex_valid = (uint8_t *)malloc((size_t)ex_size);
    float *dxs = (float *)malloc(sizeof(float) * patch_size);
    float *dys = (float *)malloc(sizeof(float) * patch_size);
    float *cur_patch = (float *)malloc(sizeof(float) * patch_size);
    uint8_t *cur_valid = (uint8_t *)malloc((size_t)patch_size);
    float last_squared_step = INFINITY;
    uint32_t large_step_cnt = 0;

    const uint32_t ref_valid_num = orc_extract_extend_patch(ref, ref_u, ref_v, ex_rows, ex_cols, ex_patch, ex_valid);
    if (ref_valid_num == 0) {
        *status = ORC_OUTSIDE;
        goto done;
    }

    for (int32_t row = 0; row < patch_rows; ++row) {
        for (int32_t col = 0; col < patch_cols; ++col) {
            const int32_t ex_index = (row + 1) * ex_cols + col + 1;
            const int32_t index = row * patch_cols + col;
            if (orc_ex_neighbours_valid(ex_valid, ex_index, ex_cols)) {
                dxs[index] = ex_patch[ex_index + 1] - ex_patch[ex_index - 1];
                dys[index] = ex_patch[ex_index + ex_cols] - ex_patch[ex_index - ex_cols];
            } else {
                dxs[index] = 0.0f;
                dys[index] = 0.0f;
            }
        }
    }

    /* Luminance scaling of the reference patch (:27-46): numerator = interior of the extended
     * patch, denominator = valid count of the WHOLE extended patch (sic). */
    if (consider_luminance) {
        float ref_average = 0.0f;
        for (int32_t row = 1; row < ex_rows - 1; ++row) {
            for (int32_t col = 1; col < ex_cols - 1; ++col) {
                ref_average += ex_patch[row * ex_cols + col];
            }
        }
        ref_average /= (float)ref_valid_num;
        for (int32_t i = 0; i < patch_size; ++i) {
            dxs[i] /= ref_average;
        }
        for (int32_t i = 0; i < patch_size; ++i) {
            dys[i] /= ref_average;
        }
        for (int32_t i = 0; i < ex_size; ++i) {
            ex_patch[i] /= ref_average;
        }
    }

    *status = ORC_LARGE_RESIDUAL;
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        ++*iters;
        const uint32_t cur_valid_num = lssd_extract_cur_patch(opt, cur, ref_u, ref_v, r, t, patch_rows, patch_cols, cur_patch, cur_valid);
        if (cur_valid_num == 0) {
            break;
        }

        /* Luminance scaling of the current patch (:65-78): numerator = rows/cols 1..P-2 only,
         * denominator = full valid count (sic). */
        if (consider_luminance) {
            float cur_average = 0.0f;
            for (int32_t row = 1; row < patch_rows - 1; ++row) {
                for (int32_t col = 1; col < patch_cols - 1; ++col) {
                    cur_average += cur_patch[row * patch_cols + col];
                }
            }
            cur_average /= (float)cur_valid_num;
            for (int32_t i = 0; i < patch_size; ++i) {
                cur_patch[i] /= cur_average;
            }
        }

        float h[9] = {0}, b[3] = {0}, v[3];
        int32_t n_valid = 0;
        for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
            for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
                const float row_i = (float)drow + ref_v;
                const float col_i = (float)dcol + ref_u;
                const int32_t ex_index = (drow + opt->half_rows + 1) * ex_cols + dcol + opt->half_cols + 1;
                const int32_t index = (drow + opt->half_rows) * patch_cols + dcol + opt->half_cols;
                if (ex_valid[ex_index] && cur_valid[index]) {
                    const float s0 = r[0] * (-row_i) + r[1] * col_i;
                    const float s1 = r[2] * (-row_i) + r[3] * col_i;
                    float j[3];
                    j[0] = dxs[index] * s0 + dys[index] * s1;
                    j[1] = dxs[index];
                    j[2] = dys[index];
                    const float residual = cur_patch[index] - ex_patch[ex_index];
                    accumulate3(h, b, j, residual);
                    ++n_valid;
                }
            }
        }
        if (n_valid == 0) {
            break;
        }

        orc_ldlt_solve(3, h, b, v);
        if (isnan(v[0]) || isnan(v[1]) || isnan(v[2])) {
            *status = ORC_NUMERIC_ERROR;
            break;
        }
        se2_update(r, t, v);

        const float squared_step = vec3_squared_norm(v);
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
    free(dxs);
    free(dys);
    free(cur_patch);
    free(cur_valid);
}
