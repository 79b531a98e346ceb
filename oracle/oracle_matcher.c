/*
 * oracle_matcher.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED (the reference's
 * matcher tests assert nothing; the integer arithmetic here has no substrate dependency).
 *
 * DescriptorMatcher<BriefType>, restated from src/descriptor_matcher/descriptor_matcher.h with
 * the per-bit Hamming distance of test/test_descriptor_matcher_brief.cpp:33-45.
 * index_pairs is in/out: the caller resets it to -1 only when its size differed from n_ref
 * (descriptor_matcher.h:60-62,98-100), so stale entries survive when no candidate improves.
 */
#include "oracle_internal.h"

/* BriefMatcher::ComputeDistance, test_descriptor_matcher_brief.cpp:33-45 */
static float brief_distance(const uint8_t *a, const uint8_t *b, int32_t n_bits) {
    if (n_bits == 0) {
        return (float)INT32_MAX; /* kMaxInt32 */
    }
    int32_t distance = 0;
    for (int32_t i = 0; i < n_bits; ++i) {
        if (a[i] != b[i]) {
            ++distance;
        }
    }
    return (float)distance;
}

/* ForceMatch, descriptor_matcher.h:55-79: strict '<' against a running minimum that starts at
 * the threshold, so the lowest j wins ties and distance == threshold never matches. */
int orc_force_match_bits(const uint8_t *ref_bits, int32_t n_ref, const uint8_t *cur_bits, int32_t n_cur, int32_t n_bits, float max_distance,
                         int32_t *index_pairs) {
    if (n_cur <= 0) {
        return 0; /* :58 */
    }
    for (int32_t i = 0; i < n_ref; ++i) {
        float min_distance = max_distance;
        for (int32_t j = 0; j < n_cur; ++j) {
            const float distance = brief_distance(ref_bits + (int64_t)i * n_bits, cur_bits + (int64_t)j * n_bits, n_bits);
            if (distance < min_distance && distance < max_distance) {
                min_distance = distance;
                index_pairs[i] = j;
            }
        }
    }
    return 1;
}

/* NearbyMatch, descriptor_matcher.h:90-124: window reject on |du| > max_col, |dv| > max_row
 * (float vs int compare, :108-111), early break on distance == 0 (:119, result-neutral). */
int orc_nearby_match_bits(const uint8_t *ref_bits, int32_t n_ref, const uint8_t *cur_bits, int32_t n_cur, int32_t n_bits, float max_distance,
                          const float *pred_uv, const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance,
                          int32_t *index_pairs) {
    if (n_cur <= 0) {
        return 0;
    }
    for (int32_t i = 0; i < n_ref; ++i) {
        float min_distance = max_distance;
        for (int32_t j = 0; j < n_cur; ++j) {
            if (fabsf(pred_uv[2 * i] - cur_uv[2 * j]) > (float)max_col_distance ||
                fabsf(pred_uv[2 * i + 1] - cur_uv[2 * j + 1]) > (float)max_row_distance) {
                continue;
            }
            const float distance = brief_distance(ref_bits + (int64_t)i * n_bits, cur_bits + (int64_t)j * n_bits, n_bits);
            if (distance < min_distance && distance < max_distance) {
                min_distance = distance;
                index_pairs[i] = j;
            }
            if (distance == 0.0f) {
                break;
            }
        }
    }
    return 1;
}

/* FillMatchedPixelByPairIndices, descriptor_matcher.h:135-157.  status is in/out (the caller
 * resets it to kNotTracked when its size differed); entries > kTracked are skipped. */
int orc_fill_matched_pixels(const int32_t *index_pairs, int32_t n_ref, const float *cur_uv, int32_t n_cur, float *matched_uv, uint8_t *status) {
    for (int32_t i = 0; i < n_ref; ++i) {
        if (status[i] > ORC_TRACKED) {
            continue;
        }
        const int32_t j = index_pairs[i];
        if (j >= 0 && j < n_cur) {
            matched_uv[2 * i] = cur_uv[2 * j];
            matched_uv[2 * i + 1] = cur_uv[2 * j + 1];
            status[i] = ORC_TRACKED;
        } else {
            status[i] = ORC_LARGE_RESIDUAL;
        }
    }
    return 1;
}
