// feature_point_harris_detector.cpp — DetectGoodFeatures on the device (ftk_harris_detect).
#include "feature_point_harris_detector.h"

#include <string>

#include "device_runtime.h"
#include "ftk.h"
#include "slam_log_reporter.h"

namespace feature_detector {

bool FeaturePointHarrisDetector::DetectGoodFeatures(const GrayImage &image, const uint32_t needed_feature_num, std::vector<Vec2> &features) {
    features.clear();
    if (image.data() == nullptr || needed_feature_num == 0) {
        return false;
    }
    std::string error;
    ftk_context *ctx = feature_tracker::device::SharedContext(&error);
    if (ctx == nullptr) {
        ReportError("[FeaturePointHarrisDetector] " << error);
        return false;
    }
    ftk_image level = {image.data(), image.rows(), image.cols()};
    ftk_pyramid *dev = nullptr;
    if (ftk_pyramid_upload(ctx, &level, 1, &dev) != FTK_OK) {
        ReportError("[FeaturePointHarrisDetector] " << ftk_last_error(ctx));
        return false;
    }
    std::vector<float> uv(2 * static_cast<size_t>(needed_feature_num));
    int32_t n = 0;
    const int rc = ftk_harris_detect(ctx, dev, 0, static_cast<int32_t>(needed_feature_num), options_.kMinFeatureDistance,
                                     options_.kMinValidResponse, uv.data(), &n);
    ftk_pyramid_destroy(dev);
    if (rc != FTK_OK) {
        ReportError("[FeaturePointHarrisDetector] " << ftk_last_error(ctx));
        return false;
    }
    features.reserve(n);
    for (int32_t i = 0; i < n; ++i) {
        features.emplace_back(uv[2 * i], uv[2 * i + 1]);
    }
    return true;
}

}  // namespace feature_detector
