#include "descriptor_brief.h"

namespace feature_detector {

void BriefDescriptor::CreatePattern() {
    pattern_length_ = options_.kLength;
    pattern_half_ = options_.kHalfPatchSize;
    pattern_.resize(size_t(pattern_length_) * 4);
    uint32_t state = 0x2545F491u;
    const int32_t span = 2 * pattern_half_ + 1;
    for (auto &v : pattern_) {
        state = state * 1664525u + 1013904223u;
        v = static_cast<int8_t>(int32_t((state >> 8) % uint32_t(span)) - pattern_half_);
    }
}

bool BriefDescriptor::Compute(const GrayImage &image, const std::vector<Vec2> &pixel_uv, std::vector<BriefType> &descriptor) {
    if (image.data() == nullptr || options_.kLength <= 0 || options_.kHalfPatchSize <= 0) {
        return false;
    }
    if (pattern_length_ != options_.kLength || pattern_half_ != options_.kHalfPatchSize) {
        CreatePattern();
    }
    const int32_t rows = image.rows(), cols = image.cols();
    auto smooth = [&](int32_t r, int32_t c) {
        int32_t s = 0;
        for (int32_t dr = -1; dr <= 1; ++dr) {
            for (int32_t dc = -1; dc <= 1; ++dc) {
                s += image.GetPixelValueNoCheck(r + dr, c + dc);
            }
        }
        return s;
    };
    descriptor.clear();
    descriptor.reserve(pixel_uv.size());
    const int32_t margin = pattern_half_ + 1;
    for (const Vec2 &uv : pixel_uv) {
        BriefType bits(size_t(pattern_length_), false);
        const int32_t r = int32_t(uv.y() + 0.5f), c = int32_t(uv.x() + 0.5f);
        if (r >= margin && c >= margin && r < rows - margin && c < cols - margin) {
            for (int32_t i = 0; i < pattern_length_; ++i) {
                const int8_t *o = &pattern_[size_t(i) * 4];
                bits[i] = smooth(r + o[0], c + o[1]) < smooth(r + o[2], c + o[3]);
            }
        }
        descriptor.emplace_back(std::move(bits));
    }
    return true;
}

}  // namespace feature_detector
