/*
 * oracle_brief.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * BRIEF descriptor (Calonder et al., ECCV 2010) as the matcher's producer
 * (call sites test/test_descriptor_matcher_brief.cpp:70-76: kLength = 256, kHalfPatchSize = 8).
 * The reference takes it from the un-vendored Feature_Detector repo, so the sampling pattern
 * and smoothing below are this repo's normative definition (the same one the host stand-in
 * feature_tracker_amd/host/compat/descriptor_brief.cpp documents):
 *   - pattern: 4 offsets (dr1, dc1, dr2, dc2) per bit from the LCG x <- 1664525 x + 1013904223
 *     (seed 0x2545F491), offset = ((x >> 8) mod (2 half + 1)) - half
 *   - bit i = S(r + dr1, c + dc1) < S(r + dr2, c + dc2), S = 3x3 box SUM of the 8-bit image
 *   - (r, c) = (trunc(v + 0.5), trunc(u + 0.5)); features closer than half + 1 px to the border
 *     get an all-zero descriptor
 * Output: one byte per bit (the reference's per-bit container).
 */
#include "oracle_internal.h"

void orc_brief_pattern(int32_t n_bits, int32_t half, int8_t *pattern /* 4 * n_bits */) {
    uint32_t state = 0x2545F491u;
    const uint32_t span = (uint32_t)(2 * half + 1);
    for (int32_t i = 0; i < 4 * n_bits; ++i) {
        state = state * 1664525u + 1013904223u;
        pattern[i] = (int8_t)((int32_t)((state >> 8) % span) - half);
    }
}

static int32_t box_sum(const orc_image *img, int32_t r, int32_t c) {
    int32_t s = 0;
    for (int32_t dr = -1; dr <= 1; ++dr) {
        for (int32_t dc = -1; dc <= 1; ++dc) {
            s += orc_px(img, r + dr, c + dc);
        }
    }
    return s;
}

int orc_brief_compute(const orc_image *img, const float *uv, int32_t n, int32_t n_bits, int32_t half, uint8_t *bits /* n * n_bits */) {
    if (n_bits <= 0 || half <= 0 || half > 63) {
        return 0;
    }
    int8_t *pattern = (int8_t *)malloc((size_t)4 * n_bits);
    orc_brief_pattern(n_bits, half, pattern);
    const int32_t margin = half + 1;
    for (int32_t f = 0; f < n; ++f) {
        uint8_t *out = bits + (int64_t)f * n_bits;
        memset(out, 0, (size_t)n_bits);
        const int32_t r = orc_f2i(uv[2 * f + 1] + 0.5f);
        const int32_t c = orc_f2i(uv[2 * f] + 0.5f);
        if (!(r >= margin && c >= margin && r < img->rows - margin && c < img->cols - margin)) {
            continue;
        }
        for (int32_t i = 0; i < n_bits; ++i) {
            const int8_t *o = pattern + 4 * i;
            out[i] = box_sum(img, r + o[0], c + o[1]) < box_sum(img, r + o[2], c + o[3]);
        }
    }
    free(pattern);
    return 1;
}
