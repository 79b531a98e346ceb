// optical_flow_basic_klt.h — translation-only KLT with the reference's class name and surface
// (src/optical_flow_tracker/basic_klt/optical_flow_basic_klt.h:9-40).  inverse / direct / fast are
// selected by options().kMethod; all three run on the device.
#ifndef _OPTICAL_FLOW_BASIC_KLT_H_
#define _OPTICAL_FLOW_BASIC_KLT_H_

#include <vector>

#include "optical_flow.h"

namespace feature_tracker {

class OpticalFlowBasicKlt: public OpticalFlow {

public:
    OpticalFlowBasicKlt(): OpticalFlow() {}
    virtual ~OpticalFlowBasicKlt() = default;

    virtual std::string OpticalFlowMethodName() const override { return "Basic-Klt"; }

private:
    virtual bool TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                    std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) override;
    virtual bool TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                  std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) override;
};

}  // namespace feature_tracker

#endif  // _OPTICAL_FLOW_BASIC_KLT_H_
