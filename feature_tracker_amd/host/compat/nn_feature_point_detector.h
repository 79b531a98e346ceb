// nn_feature_point_detector.h — stand-in for Feature_Detector's NNFeaturePointDetector (un-vendored;
// the real one runs SuperPoint / DISK through onnxruntime, and both the runtime and the .onnx blobs
// are absent: /root/reference/.MISSING_LARGE_BLOBS).  It exists so that the reference's own callers
// test/test_descriptor_matcher_superpoint.cpp and test_descriptor_matcher_disk.cpp compile, link and
// run unchanged against DescriptorMatcher<T>; the network itself is out of scope (SURVEY.md section 2
// rows 9-10).  Interface as those callers use it (:45-62): options(), Initialize(),
// DetectGoodFeaturesWithDescriptor(image, features, descriptors).
//
// What it computes instead of a network: Harris corners (the device detector behind
// FeaturePointHarrisDetector) and, per corner, a hand-made float descriptor of the right length —
// a 16 x 16 (SuperPoint, 256) or 8 x 16 (DISK, 128) grid of bilinear samples spaced 2 px around the
// corner, mean removed, scaled to unit length — i.e. the same KIND of object (a unit float vector
// compared by cosine distance), not the same values as the networks'.
#ifndef _NN_FEATURE_POINT_DETECTOR_H_
#define _NN_FEATURE_POINT_DETECTOR_H_

#include <cstdint>
#include <vector>

#include "basic_type.h"
#include "datatype_image.h"

namespace feature_detector {

using SuperpointDescriptorType = FixedMat<256, 1>;
using DiskDescriptorType = FixedMat<128, 1>;

class NNFeaturePointDetector {
public:
    enum class ModelType : uint8_t {
        kSuperpoint = 0,
        kSuperpointNms = 1,
        kDisk = 2,
        kDiskNms = 3,
    };

    struct Options {
        float kMinResponse = 0.1f;
        int32_t kMinFeatureDistance = 20;
        int32_t kMaxNumberOfDetectedFeatures = 300;
        ModelType kModelType = ModelType::kSuperpointNms;
        int32_t kMaxImageRows = 480;
        int32_t kMaxImageCols = 752;
    };

    NNFeaturePointDetector() = default;
    virtual ~NNFeaturePointDetector() = default;

    bool Initialize() { return true; }

    bool DetectGoodFeaturesWithDescriptor(const GrayImage &image, std::vector<Vec2> &features, std::vector<SuperpointDescriptorType> &descriptors);
    bool DetectGoodFeaturesWithDescriptor(const GrayImage &image, std::vector<Vec2> &features, std::vector<DiskDescriptorType> &descriptors);

    Options &options() { return options_; }
    const Options &options() const { return options_; }

private:
    bool Detect(const GrayImage &image, std::vector<Vec2> &features);

    Options options_;
};

}  // namespace feature_detector

#endif  // _NN_FEATURE_POINT_DETECTOR_H_
