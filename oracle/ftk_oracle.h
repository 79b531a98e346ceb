/*
 * ftk_oracle.h — CPU restatement of the reference's sparse-tracker hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so.  The product path
 * (feature_tracker_amd/, include/ftk.h) never links, imports or calls anything here.
 *
 * PARITY UNPINNED: the reference (Horizon1026/Feature_Tracker) ships no golden vectors or
 * assertions (the programs under test/ only print timings), and it cannot be built in this image: it
 * needs the un-vendored sibling repos Slam_Utility / Feature_Detector / Visualizor2D and Eigen3
 * (CMakeLists.txt:11-41, README.md:33-41).  This oracle is therefore a line-by-line
 * restatement of the reference's own sources (cited per function as file:line relative to
 * /root/reference) on top of the substrate definitions below, which are normative for this
 * repo because their sources are absent:
 *
 *   - GrayImage::GetPixelValue(row, col, *v)  (Slam_Utility datatype_image.h, un-vendored):
 *       valid iff 0 <= row <= rows-1 && 0 <= col <= cols-1 (the author's "inside" idiom,
 *       basic_klt.cpp:107); value = bilinear with weights from row-floor(row), col-floor(col),
 *       summed ((tl + tr) + bl) + br exactly as the reference's own explicit formula
 *       (optical_flow.cpp:53-60,78-81); the +1 neighbour index is clamped to the image
 *       (its weight is exactly 0 whenever the clamp is active).  The unchecked overload clamps
 *       its base index to the image too, purely for memory safety.
 *   - ImagePyramid::CreateImagePyramid: level i+1 = 2x2 box mean of level i, truncating
 *       ((a+b+c+d) >> 2), rows/2 x cols/2.
 *   - Eigen::LDLT<MatN>::solve: Eigen 3.3.7+ published algorithm (diagonal pivoting on the
 *       largest |d_ii|, zero pivots -> pseudo-inverse component 0, unit-triangular
 *       substitutions of the form y_i -= sum_j l_ij y_j), scalar left-to-right sums.  For
 *       n = 2, 3 no inner sum has more than two terms, so Eigen's reduction order cannot
 *       matter; for n = 6 (affine) it can, and is unpinned.
 *   - Vec3::squaredNorm() = a0 + (a1 + a2): Eigen's unrolled fixed-size reduction splits in
 *       halves (redux_novec_unroller).
 *
 * All arithmetic is IEEE fp32, no FMA contraction, sequential row-major accumulation
 * (the reference builds with -O3 and no -march: CMakeLists.txt:6).
 */
#ifndef FTK_ORACLE_H_
#define FTK_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/feature_tracker.h:8-14 */
enum {
    ORC_NOT_TRACKED = 0,
    ORC_TRACKED = 1,
    ORC_LARGE_RESIDUAL = 2,
    ORC_OUTSIDE = 3,
    ORC_NUMERIC_ERROR = 4,
};

/* src/optical_flow_tracker/optical_flow.h:12-18 (kSse/kNeon fall through to kFast) */
enum { ORC_INVERSE = 0, ORC_DIRECT = 1, ORC_FAST = 2 };
enum { ORC_BASIC = 0, ORC_AFFINE = 1, ORC_LSSD = 2 };

typedef struct {
    const uint8_t *data; /* row-major, pitch == cols */
    int32_t rows;
    int32_t cols;
} orc_image;

/* src/optical_flow_tracker/optical_flow.h:20-28 */
typedef struct {
    uint32_t max_track_points;         /* kMaxTrackPointsNumber   (500)  */
    uint32_t max_iteration;            /* kMaxIteration           (15)   */
    uint32_t max_tolerance_large_step; /* kMaxToleranceLargeStep  (3)    */
    int32_t half_rows;                 /* kPatchRowHalfSize       (6)    */
    int32_t half_cols;                 /* kPatchColHalfSize       (6)    */
    float max_converge_step;           /* kMaxConvergeStep        (4e-2) */
    int32_t method;                    /* kMethod                 (kFast)*/
} orc_klt_options;

/* substrate */
int orc_get_pixel_value(const orc_image *img, float row, float col, float *value);
float orc_get_pixel_value_nocheck(const orc_image *img, float row, float col);
/* Writes levels 1..n_levels-1 into buf (contiguous, level after level); returns bytes written. */
int64_t orc_create_pyramid(const uint8_t *raw, int32_t rows, int32_t cols, int32_t n_levels, uint8_t *buf);
void orc_ldlt_solve(int n, const float *a_rowmajor, const float *b, float *x);

/* optical_flow.cpp:49-102; valid[] is one byte per extended-patch pixel */
uint32_t orc_extract_extend_patch(const orc_image *ref, float u, float v, int32_t ex_rows, int32_t ex_cols, float *ex_patch, uint8_t *valid);

/*
 * Pyramid tracker (TrackMultipleLevel of the three models) and single-image tracker
 * (TrackSingleLevel).  prior = row-major 2x2 (predict_affine_ / predict_R_cr_).
 * cur_uv and status are in/out.  iters (optional, may be NULL) receives per feature the
 * total number of Gauss-Newton iterations that sampled the images, summed over levels.
 */
int orc_klt_track_pyramid(int model, const orc_klt_options *opt, const orc_image *ref_levels, const orc_image *cur_levels, int32_t n_levels,
                          const float *ref_uv, float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance,
                          uint32_t *iters);
int orc_klt_track_single(int model, const orc_klt_options *opt, const orc_image *ref_image, const orc_image *cur_image, const float *ref_uv,
                         float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance, uint32_t *iters);

/*
 * descriptor_matcher.h:55-79 / :90-124 with the BRIEF distance of
 * test/test_descriptor_matcher_brief.cpp:33-45.  Descriptors are one byte per bit
 * (the reference's per-bit container), n_bits each.
 */
int orc_force_match_bits(const uint8_t *ref_bits, int32_t n_ref, const uint8_t *cur_bits, int32_t n_cur, int32_t n_bits, float max_distance,
                         int32_t *index_pairs);
int orc_nearby_match_bits(const uint8_t *ref_bits, int32_t n_ref, const uint8_t *cur_bits, int32_t n_cur, int32_t n_bits, float max_distance,
                          const float *pred_uv, const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance,
                          int32_t *index_pairs);
/* descriptor_matcher.h:135-157 */
int orc_fill_matched_pixels(const int32_t *index_pairs, int32_t n_ref, const float *cur_uv, int32_t n_cur, float *matched_uv, uint8_t *status);

/*
 * descriptor_matcher.h:55-79 / :90-124 with the cosine distance of the float-descriptor callers
 * (test/test_descriptor_matcher_superpoint.cpp:32-34, test_descriptor_matcher_disk.cpp:32-34);
 * descriptors are row-major float[n][dim].  dot / norm follow Eigen 3.3.7's SSE2 reduction order
 * (normative definition in oracle_float_matcher.c; unpinned — Eigen is un-vendored).
 */
float orc_eigen_dot(const float *x, const float *y, int32_t size);
float orc_eigen_norm(const float *x, int32_t size);
float orc_cosine_distance(const float *ref, const float *cur, int32_t size);
int orc_force_match_float(const float *ref, int32_t n_ref, const float *cur, int32_t n_cur, int32_t dim, float max_distance, int32_t *index_pairs);
int orc_nearby_match_float(const float *ref, int32_t n_ref, const float *cur, int32_t n_cur, int32_t dim, float max_distance, const float *pred_uv,
                           const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *index_pairs);

/* src/direct_method_tracker/direct_method_tracker.h:20-28 */
typedef struct {
    uint32_t max_track_points;   /* kMaxTrackPointsNumber (500)  */
    uint32_t max_iteration;      /* kMaxIteration         (15)   */
    int32_t half_rows;           /* kPatchRowHalfSize     (6)    */
    int32_t half_cols;           /* kPatchColHalfSize     (6)    */
    float max_converge_step;     /* kMaxConvergeStep      (1e-6) */
    float max_converge_residual; /* kMaxConvergeResidual  (2.0, unused by the reference) */
    int32_t method;              /* kMethod               (kDirect) */
} orc_direct_options;

/* DirectMethod (direct_method_tracker.cpp:35-86, :115-192); normative substrate in oracle_direct_method.c.
 * Quaternions are (w, x, y, z). */
int orc_direct_track(const orc_direct_options *opt, const orc_image *ref_levels, const orc_image *cur_levels, int32_t n_levels, const float *K,
                     const float *p_c_in_ref, const float *ref_uv, float *cur_uv, int32_t n, float *q_rc_wxyz, float *p_rc, uint8_t *status,
                     int status_valid, uint32_t *iterations);
void orc_quat_mul(const float *a, const float *b, float *out);
void orc_quat_rotate(const float *q, const float *v, float *out);
void orc_quat_inverse(const float *q, float *out);

/* BRIEF descriptor producer of the matcher (normative definition in oracle_brief.c). */
void orc_brief_pattern(int32_t n_bits, int32_t half, int8_t *pattern);
int orc_brief_compute(const orc_image *img, const float *uv, int32_t n, int32_t n_bits, int32_t half, uint8_t *bits);

/* Harris corner detector feeding the trackers (normative definition in oracle_harris.c). */
void orc_harris_response(const orc_image *img, float *response);
int32_t orc_harris_detect(const orc_image *img, int32_t max_count, int32_t min_distance, float min_response, float *uv_out);

#ifdef __cplusplus
}
#endif
#endif /* FTK_ORACLE_H_ */
