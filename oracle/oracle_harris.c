/*
 * oracle_harris.c — TEST INFRASTRUCTURE (see ftk_oracle.h).  PARITY UNPINNED.
 *
 * Harris corner detector (Harris & Stephens 1988) as the trackers' feature source
 * (call sites test/test_optical_flow.cpp:34-39: kMinFeatureDistance = 25, kMinValidResponse = 40,
 * at most 300 features).  The reference takes it from the un-vendored Feature_Detector repo, so the
 * details below are this repo's normative definition:
 *   - gradients: 3x3 Sobel on the 8-bit image, integer
 *   - structure tensor: a = sum gx^2, d = sum gy^2, b = sum gx*gy over the 5x5 window, int32 (exact)
 *   - response = ((fa*fd - fb*fb) - (0.04f*(fa+fd))*(fa+fd)) * 1e-6f in fp32, this operation order
 *   - valid centres: border = 11 px (half window + 1 + 8)
 *   - candidates: response > min_response
 *   - non-maximum suppression: a candidate survives iff it is the maximum of its
 *     (2*min_distance - 1)^2 neighbourhood under the total order (response, then smaller
 *     row-major pixel index) — a window maximum, not a greedy scan, so it is order-free
 *   - output: survivors sorted by (response descending, pixel index ascending), first max_count,
 *     as (u, v) = (col, row)
 */
#include "oracle_internal.h"

#define HARRIS_HALF 2
#define HARRIS_BORDER 11

static inline int32_t sobel_x(const orc_image *im, int32_t r, int32_t c) {
    return ((int32_t)orc_px(im, r - 1, c + 1) + 2 * (int32_t)orc_px(im, r, c + 1) + (int32_t)orc_px(im, r + 1, c + 1)) -
           ((int32_t)orc_px(im, r - 1, c - 1) + 2 * (int32_t)orc_px(im, r, c - 1) + (int32_t)orc_px(im, r + 1, c - 1));
}
static inline int32_t sobel_y(const orc_image *im, int32_t r, int32_t c) {
    return ((int32_t)orc_px(im, r + 1, c - 1) + 2 * (int32_t)orc_px(im, r + 1, c) + (int32_t)orc_px(im, r + 1, c + 1)) -
           ((int32_t)orc_px(im, r - 1, c - 1) + 2 * (int32_t)orc_px(im, r - 1, c) + (int32_t)orc_px(im, r - 1, c + 1));
}

/* response map: rows*cols floats, 0 outside the valid region */
void orc_harris_response(const orc_image *im, float *response) {
    const int32_t rows = im->rows, cols = im->cols;
    memset(response, 0, sizeof(float) * (size_t)rows * cols);
    for (int32_t r = HARRIS_BORDER; r < rows - HARRIS_BORDER; ++r) {
        for (int32_t c = HARRIS_BORDER; c < cols - HARRIS_BORDER; ++c) {
            int32_t a = 0, b = 0, d = 0;
            for (int32_t dr = -HARRIS_HALF; dr <= HARRIS_HALF; ++dr) {
                for (int32_t dc = -HARRIS_HALF; dc <= HARRIS_HALF; ++dc) {
                    const int32_t gx = sobel_x(im, r + dr, c + dc), gy = sobel_y(im, r + dr, c + dc);
                    a += gx * gx;
                    b += gx * gy;
                    d += gy * gy;
                }
            }
            const float fa = (float)a, fb = (float)b, fd = (float)d;
            const float det = fa * fd - fb * fb;
            const float tr = fa + fd;
            response[(int64_t)r * cols + c] = (det - (0.04f * tr) * tr) * 1e-6f;
        }
    }
}

typedef struct {
    float response;
    int32_t index;
} harris_cand;

static int cand_cmp(const void *pa, const void *pb) {
    const harris_cand *a = (const harris_cand *)pa, *b = (const harris_cand *)pb;
    if (a->response != b->response) {
        return a->response > b->response ? -1 : 1;
    }
    return a->index < b->index ? -1 : (a->index > b->index ? 1 : 0);
}

/* returns the number of features written (<= max_count) */
int32_t orc_harris_detect(const orc_image *im, int32_t max_count, int32_t min_distance, float min_response, float *uv_out) {
    const int32_t rows = im->rows, cols = im->cols;
    if (rows < 2 * HARRIS_BORDER + 1 || cols < 2 * HARRIS_BORDER + 1 || max_count <= 0) {
        return 0;
    }
    float *resp = (float *)malloc(sizeof(float) * (size_t)rows * cols);
    orc_harris_response(im, resp);
    const int32_t reach = (min_distance > 1 ? min_distance : 1) - 1;
    harris_cand *kept = (harris_cand *)malloc(sizeof(harris_cand) * (size_t)rows * cols);
    int32_t n_kept = 0;
    for (int32_t r = HARRIS_BORDER; r < rows - HARRIS_BORDER; ++r) {
        for (int32_t c = HARRIS_BORDER; c < cols - HARRIS_BORDER; ++c) {
            const float v = resp[(int64_t)r * cols + c];
            if (!(v > min_response)) {
                continue;
            }
            const int32_t idx = r * cols + c;
            int is_max = 1;
            for (int32_t rr = r - reach; rr <= r + reach && is_max; ++rr) {
                if (rr < HARRIS_BORDER || rr >= rows - HARRIS_BORDER) {
                    continue;
                }
                for (int32_t cc = c - reach; cc <= c + reach; ++cc) {
                    if (cc < HARRIS_BORDER || cc >= cols - HARRIS_BORDER || (rr == r && cc == c)) {
                        continue; /* only valid centres are candidates */
                    }
                    const float o = resp[(int64_t)rr * cols + cc];
                    if (!(o > min_response)) {
                        continue; /* only candidates compete */
                    }
                    if (o > v || (o == v && rr * cols + cc < idx)) {
                        is_max = 0;
                        break;
                    }
                }
            }
            if (is_max) {
                kept[n_kept].response = v;
                kept[n_kept].index = idx;
                ++n_kept;
            }
        }
    }
    qsort(kept, (size_t)n_kept, sizeof(harris_cand), cand_cmp);
    const int32_t n = n_kept < max_count ? n_kept : max_count;
    for (int32_t i = 0; i < n; ++i) {
        uv_out[2 * i] = (float)(kept[i].index % cols);
        uv_out[2 * i + 1] = (float)(kept[i].index / cols);
    }
    free(kept);
    free(resp);
    return n;
}
