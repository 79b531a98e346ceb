/*
 * oracle_substrate.c — TEST INFRASTRUCTURE (see ftk_oracle.h).
 *
 * Restates the pieces of the un-vendored substrate the reference's hot path calls
 * (Slam_Utility GrayImage / ImagePyramid, Eigen LDLT) plus the shared extended-patch
 * extractor.  PARITY UNPINNED: the substrate sources are not under /root/reference; the
 * definitions here are the normative ones for this repo (rationale in ftk_oracle.h).
 */
#include "oracle_internal.h"

/* exported wrappers of the inlined samplers (oracle_internal.h) */
float orc_get_pixel_value_nocheck(const orc_image *img, float row, float col) { return orc_bilinear(img, row, col); }

int orc_get_pixel_value(const orc_image *img, float row, float col, float *value) { return orc_sample(img, row, col, value); }

/* ImagePyramid::CreateImagePyramid (call sites test/test_optical_flow.cpp:70-71):
 * level 0 aliases the raw image; level i+1 is the truncating 2x2 box mean of level i. */
int64_t orc_create_pyramid(const uint8_t *raw, int32_t rows, int32_t cols, int32_t n_levels, uint8_t *buf) {
    const uint8_t *src = raw;
    int32_t src_rows = rows, src_cols = cols;
    int64_t written = 0;
    for (int32_t level = 1; level < n_levels; ++level) {
        const int32_t dst_rows = src_rows / 2, dst_cols = src_cols / 2;
        uint8_t *dst = buf + written;
        for (int32_t r = 0; r < dst_rows; ++r) {
            const uint8_t *top = src + (int64_t)(2 * r) * src_cols;
            const uint8_t *bottom = top + src_cols;
            for (int32_t c = 0; c < dst_cols; ++c) {
                const uint32_t sum = (uint32_t)top[2 * c] + top[2 * c + 1] + bottom[2 * c] + bottom[2 * c + 1];
                dst[(int64_t)r * dst_cols + c] = (uint8_t)(sum >> 2);
            }
        }
        written += (int64_t)dst_rows * dst_cols;
        src = dst;
        src_rows = dst_rows;
        src_cols = dst_cols;
    }
    return written;
}

/* OpticalFlow::ExtractExtendPatchInReferenceImage, optical_flow.cpp:49-102.
 * One weight set from frac(uv) (:53-60); integer lattice starting at floor(uv) - ex/2 (:63-66);
 * a lattice pixel is valid iff 0 <= row <= rows-2 && 0 <= col <= cols-2 (:73); invalid -> 0. */
uint32_t orc_extract_extend_patch(const orc_image *ref, float u, float v, int32_t ex_rows, int32_t ex_cols, float *ex_patch, uint8_t *valid) {
    const float int_row = floorf(v);
    const float int_col = floorf(u);
    const float dec_row = v - int_row;
    const float dec_col = u - int_col;
    const float w_tl = (1.0f - dec_row) * (1.0f - dec_col);
    const float w_tr = (1.0f - dec_row) * dec_col;
    const float w_bl = dec_row * (1.0f - dec_col);
    const float w_br = dec_row * dec_col;

    const int32_t min_row = orc_wadd(orc_f2i(int_row), -(ex_rows / 2));
    const int32_t min_col = orc_wadd(orc_f2i(int_col), -(ex_cols / 2));
    const int32_t max_row = orc_wadd(min_row, ex_rows);
    const int32_t max_col = orc_wadd(min_col, ex_cols);

    uint32_t valid_cnt = 0;
    int32_t k = 0;
    for (int32_t row = min_row; row < max_row; ++row) {
        for (int32_t col = min_col; col < max_col; ++col, ++k) {
            if (row < 0 || row > ref->rows - 2 || col < 0 || col > ref->cols - 2) {
                valid[k] = 0;
                ex_patch[k] = 0.0f;
            } else {
                valid[k] = 1;
                ex_patch[k] = w_tl * (float)orc_px(ref, row, col) + w_tr * (float)orc_px(ref, row, col + 1) +
                              w_bl * (float)orc_px(ref, row + 1, col) + w_br * (float)orc_px(ref, row + 1, col + 1);
                ++valid_cnt;
            }
        }
    }
    /* A lattice that wrapped around INT32 (non-finite uv) produces fewer than ex_rows*ex_cols
     * entries in the reference; every consumer then sees "no valid pixel". Zero-fill the rest. */
    for (; k < ex_rows * ex_cols; ++k) {
        valid[k] = 0;
        ex_patch[k] = 0.0f;
    }
    return valid_cnt;
}

/* x = A.ldlt().solve(b) for the fixed sizes the reference uses (2: basic_klt.cpp:97,
 * 3: lssd_klt.cpp:107, 6: affine_klt.cpp:103).  Eigen 3.3.7+ LDLT (lower, in place):
 * at step k pivot on the largest remaining |diagonal| (first maximum wins), symmetric
 * transposition restricted to the lower triangle, rank-1 downdate of column k, scale by the
 * pivot unless it is exactly zero; solve = P^T L^-T D^+ L^-1 P b with D^+ zeroing components
 * whose |d| <= FLT_MIN.  Every inner sum is scalar and left to right: sum first, then one
 * subtraction (the shape of Eigen's coefficient-based products and of its unrolled
 * fixed-size triangular solves).  For n = 2 and 3 no sum has more than two terms, so the
 * result does not depend on Eigen's reduction order; for n = 6 that order is an
 * implementation detail of the absent Eigen build and is unpinned. */
void orc_ldlt_solve(int n, const float *a, const float *b, float *x) {
    float m[6][6];
    int tr[6];
    float temp[6];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            m[i][j] = a[i * n + j];
        }
    }

    if (n == 1) {
        tr[0] = 0;
    } else {
        for (int k = 0; k < n; ++k) {
            int p = k;
            float biggest = fabsf(m[k][k]);
            for (int i = k + 1; i < n; ++i) {
                const float cand = fabsf(m[i][i]);
                if (cand > biggest) {
                    biggest = cand;
                    p = i;
                }
            }
            tr[k] = p;
            if (p != k) {
                for (int j = 0; j < k; ++j) {
                    const float t = m[k][j];
                    m[k][j] = m[p][j];
                    m[p][j] = t;
                }
                for (int i = p + 1; i < n; ++i) {
                    const float t = m[i][k];
                    m[i][k] = m[i][p];
                    m[i][p] = t;
                }
                {
                    const float t = m[k][k];
                    m[k][k] = m[p][p];
                    m[p][p] = t;
                }
                for (int i = k + 1; i < p; ++i) {
                    const float t = m[i][k];
                    m[i][k] = m[p][i];
                    m[p][i] = t;
                }
            }

            if (k > 0) {
                for (int j = 0; j < k; ++j) {
                    temp[j] = m[j][j] * m[k][j];
                }
                float dot = m[k][0] * temp[0];
                for (int j = 1; j < k; ++j) {
                    dot += m[k][j] * temp[j];
                }
                m[k][k] -= dot;
                for (int i = k + 1; i < n; ++i) {
                    float s = m[i][0] * temp[0];
                    for (int j = 1; j < k; ++j) {
                        s += m[i][j] * temp[j];
                    }
                    m[i][k] -= s;
                }
            }

            const float akk = m[k][k];
            const int pivot_valid = fabsf(akk) > 0.0f;
            if (k == 0 && !pivot_valid) {
                for (int j = 0; j < n; ++j) {
                    tr[j] = j;
                }
                break;
            }
            if (pivot_valid) {
                for (int i = k + 1; i < n; ++i) {
                    m[i][k] /= akk;
                }
            }
        }
    }
    float y[6];
    for (int i = 0; i < n; ++i) {
        y[i] = b[i];
    }
    /* y = P b */
    for (int k = 0; k < n; ++k) {
        if (tr[k] != k) {
            const float t = y[k];
            y[k] = y[tr[k]];
            y[tr[k]] = t;
        }
    }
    /* y = L^-1 y, unit lower; Eigen unrolls fixed sizes <= 8 as "y_i -= (row_i . y).sum()" */
    for (int i = 1; i < n; ++i) {
        float s = m[i][0] * y[0];
        for (int j = 1; j < i; ++j) {
            s += m[i][j] * y[j];
        }
        y[i] -= s;
    }
    /* y = D^+ y */
    for (int i = 0; i < n; ++i) {
        if (fabsf(m[i][i]) > 1.17549435e-38f) {
            y[i] /= m[i][i];
        } else {
            y[i] = 0.0f;
        }
    }
    /* y = L^-T y, unit upper seen through the transpose, dot-oriented from the bottom */
    for (int i = n - 2; i >= 0; --i) {
        float s = m[i + 1][i] * y[i + 1];
        for (int j = i + 2; j < n; ++j) {
            s += m[j][i] * y[j];
        }
        y[i] -= s;
    }
    /* x = P^T y */
    for (int k = n - 1; k >= 0; --k) {
        if (tr[k] != k) {
            const float t = y[k];
            y[k] = y[tr[k]];
            y[tr[k]] = t;
        }
    }
    for (int i = 0; i < n; ++i) {
        x[i] = y[i];
    }
}
