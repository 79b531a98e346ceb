/*
 * oracle_direct_method.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * DirectMethod, restated from
 * src/direct_method_tracker/direct_method_tracker.cpp: the pyramid driver TrackFeatures (:35-86)
 * and TrackAllFeaturesDirect (:115-192) — photometric Gauss-Newton on ONE 6-DoF pose (q_rc, p_rc)
 * over all features jointly.  kInverse / kFast are empty stubs in the reference (:108-113, :194-199):
 * they leave pose and pixels untouched, and so does this restatement.
 *
 * Substrate that is un-vendored and therefore defined here (normative for this repo):
 *   - sensor_model::CameraPinhole::LiftFromNormalizedPlaneToImagePlane (Sensor_Model repo):
 *       u = fx * x + cx, v = fy * y + cy (pinhole, no distortion).
 *   - kZeroFloat (Slam_Utility slam_basic_math.h): 1e-6f.
 *   - Eigen::Quaternionf for the reference's SSE2 build (Eigen 3.3.7 Geometry/Quaternion.h,
 *     Geometry/arch/Geometry_SSE.h), coefficients stored (x, y, z, w):
 *       a * b      x = (ax bw - az by) + (ay bz + aw bx)      [quat_product<Architecture::SSE>]
 *                  y = (ay bw - ax bz) + (az bx + aw by)
 *                  z = (az bw - ay bx) + (ax by + aw bz)
 *                  w = (aw bw - ax bx) - (az bz + ay by)
 *       q * v      uv = q.vec x v; uv += uv; (v + w uv) + q.vec x uv          [_transformVector]
 *       squaredNorm (x^2 + z^2) + (y^2 + w^2)                                  [SSE2 predux]
 *       inverse    conjugate / squaredNorm (zero quaternion when the norm is 0)
 *       normalize  coefficients / sqrt(squaredNorm) when squaredNorm > 0
 *   - Vec6::squaredNorm(): ((d0^2 + d2^2) + (d1^2 + d3^2)) + d4^2 + d5^2 (one packet, then the tail).
 *   - GrayImage::GetPixelValue and LDLT<6> as in ftk_oracle.h.
 * All fp32, no FMA.  H and b accumulate in the reference's loop order: features ascending, patch
 * pixels row-major, one rounding per addition (:181-182).
 */
#include "oracle_internal.h"

#define ORC_ZERO_FLOAT 1e-6f

typedef struct {
    float x, y, z, w;
} quat;

static quat q_mul(quat a, quat b) {
    quat r;
    r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
    r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
    r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
    r.w = (a.w * b.w - a.x * b.x) - (a.z * b.z + a.y * b.y);
    return r;
}

static float q_squared_norm(quat q) { return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w); }

static quat q_inverse(quat q) {
    const float n2 = q_squared_norm(q);
    quat r = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n2 > 0.0f) {
        r.x = -q.x / n2;
        r.y = -q.y / n2;
        r.z = -q.z / n2;
        r.w = q.w / n2;
    }
    return r;
}

static quat q_normalized(quat q) {
    const float z = q_squared_norm(q);
    if (z > 0.0f) {
        const float n = sqrtf(z);
        q.x /= n;
        q.y /= n;
        q.z /= n;
        q.w /= n;
    }
    return q;
}

static void q_rotate(quat q, const float v[3], float out[3]) {
    float uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0];
    uv[1] += uv[1];
    uv[2] += uv[2];
    const float c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
    out[0] = (v[0] + q.w * uv[0]) + c[0];
    out[1] = (v[1] + q.w * uv[1]) + c[1];
    out[2] = (v[2] + q.w * uv[2]) + c[2];
}

/* exported for the tests (quaternions as w, x, y, z) */
void orc_quat_mul(const float *a, const float *b, float *out) {
    const quat r = q_mul((quat){a[1], a[2], a[3], a[0]}, (quat){b[1], b[2], b[3], b[0]});
    out[0] = r.w;
    out[1] = r.x;
    out[2] = r.y;
    out[3] = r.z;
}
void orc_quat_rotate(const float *q, const float *v, float *out) { q_rotate((quat){q[1], q[2], q[3], q[0]}, v, out); }
void orc_quat_inverse(const float *q, float *out) {
    const quat r = q_inverse((quat){q[1], q[2], q[3], q[0]});
    out[0] = r.w;
    out[1] = r.x;
    out[2] = r.y;
    out[3] = r.z;
}

/* TrackAllFeaturesDirect, direct_method_tracker.cpp:115-192 */
static void track_all_features_direct(const orc_direct_options *opt, const orc_image *ref_image, const orc_image *cur_image, const float K[4],
                                      const float *p_c_in_ref, const float *ref_uv, float *cur_uv, int32_t n, quat *q_rc, float p_rc[3],
                                      uint32_t *iterations) {
    const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const uint32_t max_feature_id = ((uint32_t)n < opt->max_track_points) ? (uint32_t)n : opt->max_track_points;
    for (uint32_t iter = 0; iter < opt->max_iteration; ++iter) {
        float H[6][6];
        float b[6];
        memset(H, 0, sizeof(H));
        memset(b, 0, sizeof(b));
        if (iterations) {
            ++*iterations;
        }
        const quat q_inv = q_inverse(*q_rc);
        for (uint32_t i = 0; i < max_feature_id; ++i) {
            const float p_r_x = p_c_in_ref[3 * i], p_r_y = p_c_in_ref[3 * i + 1], p_r_z = p_c_in_ref[3 * i + 2];
            if (p_r_z < ORC_ZERO_FLOAT) {
                continue; /* :128 */
            }
            const float p_r_z_inv = 1.0f / p_r_z;
            const float p_r_z2_inv = p_r_z_inv * p_r_z_inv;
            /* :138-139 */
            const float diff[3] = {p_r_x - p_rc[0], p_r_y - p_rc[1], p_r_z - p_rc[2]};
            float p_cur[3];
            q_rotate(q_inv, diff, p_cur);
            if (p_cur[2] < ORC_ZERO_FLOAT) {
                continue;
            }
            /* :141-142 */
            const float nx = p_cur[0] / p_cur[2], ny = p_cur[1] / p_cur[2];
            cur_uv[2 * i] = fx * nx + cx;
            cur_uv[2 * i + 1] = fy * ny + cy;
            /* :145-148 — operator precedence as written */
            float J[2][6];
            J[0][0] = fx * p_r_z_inv;
            J[0][1] = 0.0f;
            J[0][2] = -fx * p_r_x * p_r_z2_inv;
            J[0][3] = -fx * p_r_x * p_r_y * p_r_z2_inv;
            J[0][4] = fx + fx * p_r_x * p_r_x * p_r_z2_inv;
            J[0][5] = -fx * p_r_y * p_r_z_inv;
            J[1][0] = 0.0f;
            J[1][1] = fy * p_r_z_inv;
            J[1][2] = -fy * p_r_y * p_r_z2_inv;
            J[1][3] = -fy - fy * p_r_y * p_r_y * p_r_z2_inv;
            J[1][4] = fy * p_r_x * p_r_y * p_r_z2_inv;
            J[1][5] = fy * p_r_x * p_r_z_inv;

            for (int32_t drow = -opt->half_rows; drow <= opt->half_rows; ++drow) {
                for (int32_t dcol = -opt->half_cols; dcol <= opt->half_cols; ++dcol) {
                    const float row_i = (float)drow + ref_uv[2 * i + 1];
                    const float col_i = (float)dcol + ref_uv[2 * i];
                    const float row_j = (float)drow + cur_uv[2 * i + 1];
                    const float col_j = (float)dcol + cur_uv[2 * i];
                    float t[6];
                    if (orc_get_pixel_value(cur_image, row_j, col_j - 1.0f, &t[0]) && orc_get_pixel_value(cur_image, row_j, col_j + 1.0f, &t[1]) &&
                        orc_get_pixel_value(cur_image, row_j - 1.0f, col_j, &t[2]) && orc_get_pixel_value(cur_image, row_j + 1.0f, col_j, &t[3]) &&
                        orc_get_pixel_value(ref_image, row_i, col_i, &t[4]) && orc_get_pixel_value(cur_image, row_j, col_j, &t[5])) {
                        const float gx = (t[1] - t[0]) * 0.5f, gy = (t[3] - t[2]) * 0.5f;
                        const float residual = t[5] - t[4];
                        float jac[6];
                        for (int k = 0; k < 6; ++k) {
                            jac[k] = gx * J[0][k] + gy * J[1][k];
                        }
                        for (int r = 0; r < 6; ++r) {
                            for (int c = 0; c < 6; ++c) {
                                H[r][c] += jac[r] * jac[c];
                            }
                            b[r] += residual * jac[r];
                        }
                    }
                }
            }
        }
        float dx[6];
        orc_ldlt_solve(6, &H[0][0], b, dx);
        int has_nan = 0;
        for (int k = 0; k < 6; ++k) {
            has_nan |= isnan(dx[k]);
        }
        if (has_nan) {
            break; /* :173 */
        }
        p_rc[0] += dx[0];
        p_rc[1] += dx[1];
        p_rc[2] += dx[2];
        const quat dq = q_normalized((quat){dx[3] * 0.5f, dx[4] * 0.5f, dx[5] * 0.5f, 1.0f});
        *q_rc = q_normalized(q_mul(dq, *q_rc));
        const float sq = (((dx[0] * dx[0] + dx[2] * dx[2]) + (dx[1] * dx[1] + dx[3] * dx[3])) + dx[4] * dx[4]) + dx[5] * dx[5];
        if (sq < opt->max_converge_step) {
            break; /* :181 */
        }
    }
}

/* DirectMethod::TrackFeatures (camera-frame overload), direct_method_tracker.cpp:35-86.
 * q_rc is (w, x, y, z), in/out; p_rc in/out; cur_uv in/out (the caller applies the "sizes differ ->
 * cur = ref" rule, :42-44); status in/out, status_valid = 0 reproduces the size-mismatch reset to
 * kTracked (:73-75).  iterations (optional) counts Gauss-Newton iterations over all levels. */
int orc_direct_track(const orc_direct_options *opt, const orc_image *ref_levels, const orc_image *cur_levels, int32_t n_levels, const float *K,
                     const float *p_c_in_ref, const float *ref_uv, float *cur_uv, int32_t n, float *q_rc_wxyz, float *p_rc, uint8_t *status,
                     int status_valid, uint32_t *iterations) {
    if (n <= 0 || n_levels <= 0) {
        return 0; /* :38-39 */
    }
    if (iterations) {
        *iterations = 0;
    }
    float *scaled_ref = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    const float scale = (float)(1 << (n_levels - 1));
    for (int32_t i = 0; i < 2 * n; ++i) {
        scaled_ref[i] = ref_uv[i] / scale;
    }
    float scaled_K[4] = {K[0] / scale, K[1] / scale, K[2] / scale, K[3] / scale};
    quat q = {q_rc_wxyz[1], q_rc_wxyz[2], q_rc_wxyz[3], q_rc_wxyz[0]};
    for (int32_t level = n_levels - 1; level > -1; --level) {
        if (opt->method == ORC_DIRECT) {
            track_all_features_direct(opt, &ref_levels[level], &cur_levels[level], scaled_K, p_c_in_ref, scaled_ref, cur_uv, n, &q, p_rc, iterations);
        }
        if (level == 0) {
            break;
        }
        for (int32_t i = 0; i < 2 * n; ++i) {
            scaled_ref[i] *= 2.0f;
        }
        for (int k = 0; k < 4; ++k) {
            scaled_K[k] *= 2.0f;
        }
    }
    free(scaled_ref);
    q_rc_wxyz[0] = q.w;
    q_rc_wxyz[1] = q.x;
    q_rc_wxyz[2] = q.y;
    q_rc_wxyz[3] = q.z;
    /* :72-83 */
    if (!status_valid) {
        memset(status, ORC_TRACKED, (size_t)n);
    }
    const orc_image *bottom = &ref_levels[0];
    for (int32_t i = 0; i < n; ++i) {
        if (cur_uv[2 * i] < 0.0f || cur_uv[2 * i] > (float)(bottom->cols - 1) || cur_uv[2 * i + 1] < 0.0f || cur_uv[2 * i + 1] > (float)(bottom->rows - 1)) {
            status[i] = ORC_OUTSIDE;
        }
    }
    return 1;
}
