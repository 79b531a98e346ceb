// feature_point_harris_detector.h — stand-in for Feature_Detector's Harris corner detector
// (un-vendored).  Interface as the reference's callers use it (test_optical_flow.cpp:34-39):
// options().kMinFeatureDistance / kMinValidResponse, DetectGoodFeatures(image, max_count, out).
// Definition (pinned by oracle/oracle_harris.c, computed on the device by ftk_harris_detect):
// 3x3 Sobel gradients, structure tensor summed over a 5x5 window in exact integers,
// response = (det - 0.04 trace^2) * 1e-6 (so that thresholds around 40 are meaningful on 8-bit
// images), window-maximum suppression over (2 kMinFeatureDistance - 1)^2, strongest first.
#ifndef _FEATURE_POINT_HARRIS_DETECTOR_H_
#define _FEATURE_POINT_HARRIS_DETECTOR_H_

#include <cstdint>
#include <vector>

#include "basic_type.h"
#include "datatype_image.h"

namespace feature_detector {

class FeaturePointHarrisDetector {
public:
    struct Options {
        int32_t kMinFeatureDistance = 20;
        float kMinValidResponse = 40.0f;
    };

    FeaturePointHarrisDetector() = default;
    virtual ~FeaturePointHarrisDetector() = default;

    bool DetectGoodFeatures(const GrayImage &image, const uint32_t needed_feature_num, std::vector<Vec2> &features);

    Options &options() { return options_; }
    const Options &options() const { return options_; }

private:
    Options options_;
};

}  // namespace feature_detector

#endif  // _FEATURE_POINT_HARRIS_DETECTOR_H_
