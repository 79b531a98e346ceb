// descriptor_brief.h — stand-in for Feature_Detector's BRIEF descriptor (un-vendored).
// BriefType is a per-bit container (one bool per test), which is what the reference's matcher test
// iterates over (test_descriptor_matcher_brief.cpp:33-45).  Pattern: kLength point pairs drawn once
// from a fixed-seed LCG inside [-kHalfPatchSize, kHalfPatchSize]^2; bit = I(p1) < I(p2) on the 3x3
// box-smoothed image; features closer than the patch to the border get an all-zero descriptor.
#ifndef _FEATURE_DESCRIPTOR_BRIEF_H_
#define _FEATURE_DESCRIPTOR_BRIEF_H_

#include <cstdint>
#include <vector>

#include "basic_type.h"
#include "datatype_image.h"

namespace feature_detector {

using BriefType = std::vector<bool>;

class BriefDescriptor {
public:
    struct Options {
        int32_t kLength = 256;
        int32_t kHalfPatchSize = 8;
    };

    BriefDescriptor() = default;
    virtual ~BriefDescriptor() = default;

    bool Compute(const GrayImage &image, const std::vector<Vec2> &pixel_uv, std::vector<BriefType> &descriptor);

    Options &options() { return options_; }
    const Options &options() const { return options_; }

private:
    void CreatePattern();

    Options options_;
    std::vector<int8_t> pattern_;  // 4 offsets (dr1, dc1, dr2, dc2) per bit
    int32_t pattern_length_ = 0;
    int32_t pattern_half_ = 0;
};

}  // namespace feature_detector

#endif  // _FEATURE_DESCRIPTOR_BRIEF_H_
