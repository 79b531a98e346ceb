// Replaces lssd_klt.cpp:7-94 (+ :96-250 and lssd_klt_fast.cpp) with one kernel launch per call.
#include "optical_flow_lssd_klt.h"

#include "ftk.h"

namespace feature_tracker {

bool OpticalFlowLssdKlt::TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                            std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    const float prior[4] = {predict_R_cr_(0, 0), predict_R_cr_(0, 1), predict_R_cr_(1, 0), predict_R_cr_(1, 1)};
    return TrackOnDevice(FTK_MODEL_LSSD, &ref_pyramid, &cur_pyramid, nullptr, nullptr, ref_pixel_uv, cur_pixel_uv, status, prior,
                         consider_patch_luminance_);
}

bool OpticalFlowLssdKlt::TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                          std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    const float prior[4] = {predict_R_cr_(0, 0), predict_R_cr_(0, 1), predict_R_cr_(1, 0), predict_R_cr_(1, 1)};
    return TrackOnDevice(FTK_MODEL_LSSD, nullptr, nullptr, &ref_image, &cur_image, ref_pixel_uv, cur_pixel_uv, status, prior,
                         consider_patch_luminance_);
}

}  // namespace feature_tracker
