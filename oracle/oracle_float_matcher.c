/*
 * oracle_float_matcher.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * DescriptorMatcher<FloatDescriptor>::ForceMatch / NearbyMatch (src/descriptor_matcher/
 * descriptor_matcher.h:55-79, :90-124) with the distance the reference's float-descriptor callers
 * define (test/test_descriptor_matcher_superpoint.cpp:32-34, test_descriptor_matcher_disk.cpp:32-34):
 *
 *     0.5f - ref.dot(cur) / ref.norm() / cur.norm() * 0.5f
 *
 * on Eigen float vectors (SuperPoint: 256, DISK: 128 components).  Eigen itself is not in the image
 * (un-vendored; README.md:39 asks for >= 3.3.7), so dot() / norm() follow Eigen 3.3.7's published
 * reduction for the reference's build (-O3, no -march => SSE2 packets of 4 floats, no FMA):
 * redux_impl<LinearVectorizedTraversal, NoUnrolling> (Core/Redux.h) — two packet accumulators
 * over the 8-aligned part, their sum, one more packet if size % 8 >= 4, then the SSE2 predux
 * (a0 + a2) + (a1 + a3), then the scalar tail left to right.  Sizes below 4 reduce scalar, left
 * to right.  (Cost of both descriptor sizes exceeds EIGEN_UNROLLING_LIMIT, so the unrolled tree
 * variant does not apply; a dynamic-size vector takes the same path.)
 */
#include "oracle_internal.h"

/* sum_k x[k] * y[k] in Eigen's order; products are rounded to fp32 before they are added */
float orc_eigen_dot(const float *x, const float *y, int32_t size) {
    if (size <= 0) {
        return 0.0f;
    }
    const int32_t aligned_size = (size / 4) * 4;
    const int32_t aligned_end2 = (size / 8) * 8;
    float res;
    if (aligned_size) {
        float p0[4], p1[4];
        for (int q = 0; q < 4; ++q) {
            p0[q] = x[q] * y[q];
        }
        if (aligned_size > 4) {
            for (int q = 0; q < 4; ++q) {
                p1[q] = x[4 + q] * y[4 + q];
            }
            for (int32_t index = 8; index < aligned_end2; index += 8) {
                for (int q = 0; q < 4; ++q) {
                    p0[q] = p0[q] + x[index + q] * y[index + q];
                    p1[q] = p1[q] + x[index + 4 + q] * y[index + 4 + q];
                }
            }
            for (int q = 0; q < 4; ++q) {
                p0[q] = p0[q] + p1[q];
            }
            if (aligned_size > aligned_end2) {
                for (int q = 0; q < 4; ++q) {
                    p0[q] = p0[q] + x[aligned_end2 + q] * y[aligned_end2 + q];
                }
            }
        }
        res = (p0[0] + p0[2]) + (p0[1] + p0[3]); /* SSE2 predux<Packet4f> */
        for (int32_t index = aligned_size; index < size; ++index) {
            res = res + x[index] * y[index];
        }
    } else {
        res = x[0] * y[0];
        for (int32_t index = 1; index < size; ++index) {
            res = res + x[index] * y[index];
        }
    }
    return res;
}

/* norm() = sqrt(squaredNorm()), squaredNorm() = cwiseAbs2().sum() (Core/Dot.h) */
float orc_eigen_norm(const float *x, int32_t size) { return sqrtf(orc_eigen_dot(x, x, size)); }

/* SuperpointMatcher / DiskMatcher::ComputeDistance */
float orc_cosine_distance(const float *ref, const float *cur, int32_t size) {
    return 0.5f - orc_eigen_dot(ref, cur, size) / orc_eigen_norm(ref, size) / orc_eigen_norm(cur, size) * 0.5f;
}

static int match_float(const float *ref, int32_t n_ref, const float *cur, int32_t n_cur, int32_t dim, float max_distance, const float *pred_uv,
                       const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *index_pairs) {
    if (n_cur <= 0) {
        return 0; /* descriptor_matcher.h:58,94 */
    }
    /* norm() of a descriptor does not depend on its partner: evaluate once (same value every call) */
    float *norm_ref = (float *)malloc(sizeof(float) * (size_t)(n_ref > 0 ? n_ref : 1));
    float *norm_cur = (float *)malloc(sizeof(float) * (size_t)n_cur);
    for (int32_t i = 0; i < n_ref; ++i) {
        norm_ref[i] = orc_eigen_norm(ref + (int64_t)i * dim, dim);
    }
    for (int32_t j = 0; j < n_cur; ++j) {
        norm_cur[j] = orc_eigen_norm(cur + (int64_t)j * dim, dim);
    }
    for (int32_t i = 0; i < n_ref; ++i) {
        float min_distance = max_distance;
        for (int32_t j = 0; j < n_cur; ++j) {
            if (pred_uv) {
                if (fabsf(pred_uv[2 * i] - cur_uv[2 * j]) > (float)max_col_distance ||
                    fabsf(pred_uv[2 * i + 1] - cur_uv[2 * j + 1]) > (float)max_row_distance) {
                    continue; /* :108-111 */
                }
            }
            const float distance = 0.5f - orc_eigen_dot(ref + (int64_t)i * dim, cur + (int64_t)j * dim, dim) / norm_ref[i] / norm_cur[j] * 0.5f;
            if (distance < min_distance && distance < max_distance) {
                min_distance = distance;
                index_pairs[i] = j;
            }
            if (pred_uv && distance == 0.0f) {
                break; /* :119 */
            }
        }
    }
    free(norm_ref);
    free(norm_cur);
    return 1;
}

int orc_force_match_float(const float *ref, int32_t n_ref, const float *cur, int32_t n_cur, int32_t dim, float max_distance, int32_t *index_pairs) {
    return match_float(ref, n_ref, cur, n_cur, dim, max_distance, NULL, NULL, 0, 0, index_pairs);
}

int orc_nearby_match_float(const float *ref, int32_t n_ref, const float *cur, int32_t n_cur, int32_t dim, float max_distance, const float *pred_uv,
                           const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *index_pairs) {
    return match_float(ref, n_ref, cur, n_cur, dim, max_distance, pred_uv, cur_uv, max_col_distance, max_row_distance, index_pairs);
}
