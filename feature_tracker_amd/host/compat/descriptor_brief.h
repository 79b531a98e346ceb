// descriptor_brief.h — stand-in for Feature_Detector's BRIEF descriptor (un-vendored).
// BriefType is a per-bit container (one bool per test), which is what the reference's matcher test
// iterates over (test_descriptor_matcher_brief.cpp:33-45).  Pattern: kLength point pairs drawn once
// from a fixed-seed LCG inside [-kHalfPatchSize, kHalfPatchSize]^2; bit = S(p1) < S(p2) on the 3x3
// box sums; features closer than the patch to the border get an all-zero descriptor.  Compute runs
// on the device (ftk_brief_compute, include/ftk.h); the definition is pinned by oracle/oracle_brief.c.
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
    Options options_;
};

}  // namespace feature_detector

#endif  // _FEATURE_DESCRIPTOR_BRIEF_H_
