/*
 * oracle_affine_klt.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * 6-DoF affine KLT, restated from
 *   src/optical_flow_tracker/affine_klt/optical_flow_affine_klt.cpp       (inverse, direct)
 *   src/optical_flow_tracker/affine_klt/optical_flow_affine_klt_fast.cpp  (fast)
 * affine[] is row-major [a00, a01, a10, a11].
 */
#include "oracle_internal.h"

/* Upper-triangle accumulators in the reference's own order (affine_klt.cpp:229-249). */
enum {
    H00, H01, H02, H03, H04, H05, H11, H12, H13, H14, H15, H22, H23, H24, H25, H33, H34, H35, H44, H45, H55, H_COUNT
};

static void affine_expand(const float *u, float *m /* 6x6 row-major */) {
    static const int idx[6][6] = {
        {H00, H01, H02, H03, H04, H05}, {H01, H11, H12, H13, H14, H15}, {H02, H12, H22, H23, H24, H25},
        {H03, H13, H23, H33, H34, H35}, {H04, H14, H24, H34, H44, H45}, {H05, H15, H25, H35, H45, H55},
    };
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) {
            m[i * 6 + j] = u[idx[i][j]];
        }
    }
}

/* ConstructIncrementalFunction, affine_klt.cpp:131-273.  The Jacobian uses the ABSOLUTE
 * warped coordinates x = col_j, y = row_j (:219-220).  H(3,4) accumulates yy*dxdy (:245, sic). */
static int32_t affine_build_normal_equations(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v,
                                             float cur_u, float cur_v, const float *affine, float *m, float *b) {
    const int direct = (opt->method == ORC_DIRECT);
    const orc_image *grad_img = direct ? cur : ref;
    float u[H_COUNT];
    memset(u, 0, sizeof(u));
    for (int i = 0; i < 6; ++i) {
        b[i] = 0.0f;
    }
    int32_t n_valid = 0;
    for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
        for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
            const float row_i = (float)drow + ref_v;
            const float col_i = (float)dcol + ref_u;
            /* affine * Vec2(dcol, drow) (:204-209) */
            const float warped_x = affine[0] * (float)dcol + affine[1] * (float)drow;
            const float warped_y = affine[2] * (float)dcol + affine[3] * (float)drow;
            const float row_j = warped_y + cur_v;
            const float col_j = warped_x + cur_u;
            const float grow = direct ? row_j : row_i;
            const float gcol = direct ? col_j : col_i;
            float left, right, top, bottom, i_ref, i_cur;
            if (orc_sample(grad_img, grow, gcol - 1.0f, &left) && orc_sample(grad_img, grow, gcol + 1.0f, &right) &&
                orc_sample(grad_img, grow - 1.0f, gcol, &top) && orc_sample(grad_img, grow + 1.0f, gcol, &bottom) &&
                orc_sample(ref, row_i, col_i, &i_ref) && orc_sample(cur, row_j, col_j, &i_cur)) {
                const float dx = right - left;
                const float dy = bottom - top;
                const float dt = i_cur - i_ref;
                const float x = col_j, y = row_j;
                const float xx = x * x, yy = y * y, xy = x * y;
                const float dxdx = dx * dx, dydy = dy * dy, dxdy = dx * dy;
                u[H00] += xx * dxdx;
                u[H01] += xx * dxdy;
                u[H02] += xy * dxdx;
                u[H03] += xy * dxdy;
                u[H04] += x * dxdx;
                u[H05] += x * dxdy;
                u[H11] += xx * dydy;
                u[H12] += xy * dxdy;
                u[H13] += xy * dydy;
                u[H14] += x * dxdy;
                u[H15] += x * dydy;
                u[H22] += yy * dxdx;
                u[H23] += yy * dxdy;
                u[H24] += y * dxdx;
                u[H25] += y * dxdy;
                u[H33] += yy * dydy;
                u[H34] += yy * dxdy; /* sic: reference quirk, kept */
                u[H35] += y * dydy;
                u[H44] += dxdx;
                u[H45] += dxdy;
                u[H55] += dydy;
                b[0] -= dt * x * dx;
                b[1] -= dt * x * dy;
                b[2] -= dt * y * dx;
                b[3] -= dt * y * dy;
                b[4] -= dt * dx;
                b[5] -= dt * dy;
                ++n_valid;
            }
        }
    }
    affine_expand(u, m);
    return n_valid;
}

/* cur += z[0:2]*cur.x + z[2:4]*cur.y + z[4:6]; affine columns += z[0:2], z[2:4]
 * (affine_klt.cpp:104,112-117 and affine_klt_fast.cpp:48-53). */
static void affine_step(const float *z, const float *cur_uv, float *v) {
    v[0] = (z[0] * cur_uv[0] + z[2] * cur_uv[1]) + z[4];
    v[1] = (z[1] * cur_uv[0] + z[3] * cur_uv[1]) + z[5];
}

static void affine_update_matrix(float *affine, const float *z) {
    affine[0] += z[0]; /* col(0) += z.head<2>() */
    affine[2] += z[1];
    affine[1] += z[2]; /* col(1) += z.segment<2>(2) */
    affine[3] += z[3];
}

/* TrackOneFeature, affine_klt.cpp:93-129. */
void orc_affine_track_one(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                          float *affine, uint8_t *status, uint32_t *iters) {
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        float m[36], b[6], z[6], v[2];
        ++*iters;
        if (affine_build_normal_equations(opt, ref, cur, ref_u, ref_v, cur_uv[0], cur_uv[1], affine, m, b) == 0) {
            break;
        }
        orc_ldlt_solve(6, m, b, z);
        affine_step(z, cur_uv, v);
        if (isnan(v[0]) || isnan(v[1])) {
            *status = ORC_NUMERIC_ERROR;
            break;
        }
        cur_uv[0] += v[0];
        cur_uv[1] += v[1];
        affine_update_matrix(affine, z);
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

/* ComputeBias, affine_klt_fast.cpp:140-188: one checked bilinear fetch of cur per patch pixel
 * at the affine-warped position; weights by the current absolute coordinates (:174-179). */
static int32_t affine_fast_bias(const orc_klt_options *opt, const orc_image *cur, float cur_u, float cur_v, const float *ex_patch,
                                const uint8_t *ex_valid, int32_t ex_cols, const float *dxs, const float *dys, const float *affine, float *b) {
    const int32_t patch_cols = ex_cols - 2;
    int32_t n_valid = 0;
    for (int i = 0; i < 6; ++i) {
        b[i] = 0.0f;
    }
    for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
        for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
            const float warped_x = affine[0] * (float)dcol + affine[1] * (float)drow;
            const float warped_y = affine[2] * (float)dcol + affine[3] * (float)drow;
            const float row_c = warped_y + cur_v;
            const float col_c = warped_x + cur_u;
            float i_cur = 0.0f;
            if (!orc_sample(cur, row_c, col_c, &i_cur)) {
                continue;
            }
            const int32_t row_in_ex = drow + opt->half_rows + 1;
            const int32_t col_in_ex = dcol + opt->half_cols + 1;
            const int32_t ex_index = row_in_ex * ex_cols + col_in_ex;
            if (!ex_valid[ex_index]) {
                continue;
            }
            const float dt = i_cur - ex_patch[ex_index];
            const int32_t index = (row_in_ex - 1) * patch_cols + (col_in_ex - 1);
            const float dx = dxs[index], dy = dys[index];
            b[0] -= dt * col_c * dx;
            b[1] -= dt * col_c * dy;
            b[2] -= dt * row_c * dx;
            b[3] -= dt * row_c * dy;
            b[4] -= dt * dx;
            b[5] -= dt * dy;
            ++n_valid;
        }
    }
    return n_valid;
}

/* TrackOneFeatureFast, affine_klt_fast.cpp:7-69 (+ PrecomputeJacobianAndHessian :71-138). */
void orc_affine_track_one_fast(const orc_klt_options *opt, const orc_image *ref, const orc_image *cur, float ref_u, float ref_v, float *cur_uv,
                               float *affine, uint8_t *status, uint32_t *iters) {
    const int32_t patch_rows = 2 * opt->half_rows + 1, patch_cols = 2 * opt->half_cols + 1;
    const int32_t ex_rows = patch_rows + 2, ex_cols = patch_cols + 2;
    float *ex_patch = (float *)malloc(sizeof(float) * ex_rows * ex_cols);
    uint8_t *ex_valid = (uint8_t *)malloc((size_t)ex_rows * ex_cols);
    float *dxs = (float *)malloc(sizeof(float) * patch_rows * patch_cols);
    float *dys = (float *)malloc(sizeof(float) * patch_rows * patch_cols);
    float u[H_COUNT];
    float m[36];
    float last_squared_step = INFINITY;
    uint32_t large_step_cnt = 0;
    memset(u, 0, sizeof(u));

    if (orc_extract_extend_patch(ref, ref_u, ref_v, ex_rows, ex_cols, ex_patch, ex_valid) == 0) {
        *status = ORC_OUTSIDE;
        goto done;
    }

    /* H once per level, anchored at cur_uv on level entry (:95-96); only 18 sums are
     * accumulated, the other three are copies (:130-132), which keeps the H(3,4) quirk. */
    for (int32_t row = 0; row < patch_rows; ++row) {
        for (int32_t col = 0; col < patch_cols; ++col) {
            const int32_t ex_index = (row + 1) * ex_cols + col + 1;
            const int32_t index = row * patch_cols + col;
            if (orc_ex_neighbours_valid(ex_valid, ex_index, ex_cols)) {
                const float dx = ex_patch[ex_index + 1] - ex_patch[ex_index - 1];
                const float dy = ex_patch[ex_index + ex_cols] - ex_patch[ex_index - ex_cols];
                dxs[index] = dx;
                dys[index] = dy;
                const float x = (float)(col - opt->half_cols) + cur_uv[0];
                const float y = (float)(row - opt->half_rows) + cur_uv[1];
                const float xx = x * x, yy = y * y, xy = x * y;
                const float dxdx = dx * dx, dydy = dy * dy, dxdy = dx * dy;
                u[H00] += xx * dxdx;
                u[H01] += xx * dxdy;
                u[H02] += xy * dxdx;
                u[H03] += xy * dxdy;
                u[H04] += x * dxdx;
                u[H05] += x * dxdy;
                u[H11] += xx * dydy;
                u[H13] += xy * dydy;
                u[H15] += x * dydy;
                u[H22] += yy * dxdx;
                u[H23] += yy * dxdy;
                u[H24] += y * dxdx;
                u[H25] += y * dxdy;
                u[H33] += yy * dydy;
                u[H35] += y * dydy;
                u[H44] += dxdx;
                u[H45] += dxdy;
                u[H55] += dydy;
            } else {
                dxs[index] = 0.0f;
                dys[index] = 0.0f;
            }
        }
    }
    u[H12] = u[H03];
    u[H14] = u[H05];
    u[H34] = u[H23];
    affine_expand(u, m);

    *status = ORC_LARGE_RESIDUAL;
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        float b[6], z[6], v[2];
        ++*iters;
        if (affine_fast_bias(opt, cur, cur_uv[0], cur_uv[1], ex_patch, ex_valid, ex_cols, dxs, dys, affine, b) == 0) {
            break;
        }
        orc_ldlt_solve(6, m, b, z);
        if (isnan(z[0]) || isnan(z[1]) || isnan(z[2]) || isnan(z[3]) || isnan(z[4]) || isnan(z[5])) {
            *status = ORC_NUMERIC_ERROR;
            break;
        }
        affine_step(z, cur_uv, v);
        cur_uv[0] += v[0];
        cur_uv[1] += v[1];
        affine_update_matrix(affine, z);
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
    free(dxs);
    free(dys);
}
