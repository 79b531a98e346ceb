// descriptor_brief.cpp — BriefDescriptor::Compute on the device (ftk_brief_compute); the per-bit
// containers the reference's callers expect are unpacked from the device's packed words.
#include "descriptor_brief.h"

#include <string>

#include "bit_words.h"
#include "device_runtime.h"
#include "ftk.h"
#include "slam_log_reporter.h"

namespace feature_detector {

bool BriefDescriptor::Compute(const GrayImage &image, const std::vector<Vec2> &pixel_uv, std::vector<BriefType> &descriptor) {
    if (image.data() == nullptr || options_.kLength <= 0 || options_.kHalfPatchSize <= 0) {
        return false;
    }
    descriptor.clear();
    if (pixel_uv.empty()) {
        return true;
    }
    std::string error;
    ftk_context *ctx = feature_tracker::device::SharedContext(&error);
    if (ctx == nullptr) {
        ReportError("[BriefDescriptor] " << error);
        return false;
    }
    ftk_image level = {image.data(), image.rows(), image.cols()};
    ftk_pyramid *dev = nullptr;
    if (ftk_pyramid_upload(ctx, &level, 1, &dev) != FTK_OK) {
        ReportError("[BriefDescriptor] " << ftk_last_error(ctx));
        return false;
    }
    const int32_t n = static_cast<int32_t>(pixel_uv.size());
    const int32_t n_words = (options_.kLength + 31) / 32;
    std::vector<uint32_t> words(static_cast<size_t>(n) * n_words);
    const int rc = ftk_brief_compute(ctx, dev, 0, pixel_uv[0].data(), n, options_.kLength, options_.kHalfPatchSize, words.data());
    ftk_pyramid_destroy(dev);
    if (rc != FTK_OK) {
        ReportError("[BriefDescriptor] " << ftk_last_error(ctx));
        return false;
    }
    descriptor.resize(pixel_uv.size());
    for (int32_t f = 0; f < n; ++f) {
        feature_tracker::bit_words::Unpack(&words[static_cast<size_t>(f) * n_words], static_cast<size_t>(options_.kLength), descriptor[f]);
    }
    return true;
}

}  // namespace feature_detector
