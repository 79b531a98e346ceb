// optical_flow_affine_klt.h — 6-DoF affine KLT with the reference's class name and surface
// (src/optical_flow_tracker/affine_klt/optical_flow_affine_klt.h:9-51).
#ifndef _OPTICAL_FLOW_AFFINE_KLT_H_
#define _OPTICAL_FLOW_AFFINE_KLT_H_

#include <vector>

#include "optical_flow.h"

namespace feature_tracker {

class OpticalFlowAffineKlt: public OpticalFlow {

public:
    OpticalFlowAffineKlt(): OpticalFlow() {}
    virtual ~OpticalFlowAffineKlt() = default;

    virtual std::string OpticalFlowMethodName() const override { return "Affine-Klt"; }

    // Reference for member variables.
    Mat2 &predict_affine() { return predict_affine_; }

    // Const reference for member variables.
    const Mat2 &predict_affine() const { return predict_affine_; }

private:
    virtual bool TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                    std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) override;
    virtual bool TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                  std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) override;

private:
    // Honoured by the single-image overload only, as in the reference (affine_klt.cpp:21 vs :70).
    Mat2 predict_affine_ = Mat2::Identity();
};

}  // namespace feature_tracker

#endif  // _OPTICAL_FLOW_AFFINE_KLT_H_
