// Replaces basic_klt.cpp:7-86 (+ the per-feature code it calls, :88-181 and basic_klt_fast.cpp):
// the whole per-feature / per-level / per-iteration loop nest is one kernel launch.
#include "optical_flow_basic_klt.h"

#include "ftk.h"

namespace feature_tracker {

bool OpticalFlowBasicKlt::TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                             std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    return TrackOnDevice(FTK_MODEL_BASIC, &ref_pyramid, &cur_pyramid, nullptr, nullptr, ref_pixel_uv, cur_pixel_uv, status, nullptr, false);
}

bool OpticalFlowBasicKlt::TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                           std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    return TrackOnDevice(FTK_MODEL_BASIC, nullptr, nullptr, &ref_image, &cur_image, ref_pixel_uv, cur_pixel_uv, status, nullptr, false);
}

}  // namespace feature_tracker
