// Replaces affine_klt.cpp:6-91 (+ :93-273 and affine_klt_fast.cpp) with one kernel launch per call.
#include "optical_flow_affine_klt.h"

#include "ftk.h"

namespace feature_tracker {

namespace {
// Mat2 is column-major (Eigen's default); the C ABI takes row-major [m00, m01, m10, m11].
inline void RowMajor(const Mat2 &m, float out[4]) {
    out[0] = m(0, 0);
    out[1] = m(0, 1);
    out[2] = m(1, 0);
    out[3] = m(1, 1);
}
}  // namespace

bool OpticalFlowAffineKlt::TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                              std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    float prior[4];
    RowMajor(predict_affine_, prior);
    return TrackOnDevice(FTK_MODEL_AFFINE, &ref_pyramid, &cur_pyramid, nullptr, nullptr, ref_pixel_uv, cur_pixel_uv, status, prior, false);
}

bool OpticalFlowAffineKlt::TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                            std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    float prior[4];
    RowMajor(predict_affine_, prior);
    return TrackOnDevice(FTK_MODEL_AFFINE, nullptr, nullptr, &ref_image, &cur_image, ref_pixel_uv, cur_pixel_uv, status, prior, false);
}

}  // namespace feature_tracker
