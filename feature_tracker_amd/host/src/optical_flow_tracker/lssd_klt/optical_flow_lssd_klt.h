// optical_flow_lssd_klt.h — SE(2) KLT with locally scaled SSD, the reference's class name and surface
// (src/optical_flow_tracker/lssd_klt/optical_flow_lssd_klt.h:9-55).
#ifndef _OPTICAL_FLOW_LSSD_KLT_H_
#define _OPTICAL_FLOW_LSSD_KLT_H_

#include <vector>

#include "optical_flow.h"

namespace feature_tracker {

class OpticalFlowLssdKlt: public OpticalFlow {

public:
    OpticalFlowLssdKlt(): OpticalFlow() {}
    virtual ~OpticalFlowLssdKlt() = default;

    virtual std::string OpticalFlowMethodName() const override { return "Lssd-Klt"; }

    // Reference for member variables.
    Mat2 &predict_R_cr() { return predict_R_cr_; }
    bool &consider_patch_luminance() { return consider_patch_luminance_; }

    // Const reference for member variables.
    const Mat2 &predict_R_cr() const { return predict_R_cr_; }
    const bool &consider_patch_luminance() const { return consider_patch_luminance_; }

private:
    virtual bool TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                    std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) override;
    virtual bool TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                  std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) override;

private:
    Mat2 predict_R_cr_ = Mat2::Identity();
    bool consider_patch_luminance_ = false;  // consulted by the fast method only, as in the reference
};

}  // namespace feature_tracker

#endif  // _OPTICAL_FLOW_LSSD_KLT_H_
