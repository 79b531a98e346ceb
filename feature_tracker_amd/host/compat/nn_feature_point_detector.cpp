// nn_feature_point_detector.cpp — see the header: Harris corners on the device + a hand-made patch
// descriptor on the host (stand-in for networks that are not in this image).
#include "nn_feature_point_detector.h"

#include <cmath>

#include "feature_point_harris_detector.h"

namespace feature_detector {

namespace {

// grid_rows x grid_cols bilinear samples spaced 2 px, centred on the feature; mean removed; unit length
template <int N>
void PatchDescriptor(const GrayImage &image, const Vec2 &uv, int grid_rows, int grid_cols, FixedMat<N, 1> &out) {
    float mean = 0.0f;
    int k = 0;
    for (int r = 0; r < grid_rows; ++r) {
        for (int c = 0; c < grid_cols; ++c, ++k) {
            float row = uv.y() + 2.0f * (static_cast<float>(r) - 0.5f * static_cast<float>(grid_rows - 1));
            float col = uv.x() + 2.0f * (static_cast<float>(c) - 0.5f * static_cast<float>(grid_cols - 1));
            row = std::fmin(std::fmax(row, 0.0f), static_cast<float>(image.rows() - 1));
            col = std::fmin(std::fmax(col, 0.0f), static_cast<float>(image.cols() - 1));
            float value = 0.0f;
            image.GetPixelValue(row, col, &value);
            out(k) = value;
            mean += value;
        }
    }
    mean /= static_cast<float>(N);
    float sq = 0.0f;
    for (int i = 0; i < N; ++i) {
        out(i) -= mean;
        sq += out(i) * out(i);
    }
    const float norm = std::sqrt(sq);
    if (norm > 0.0f) {
        for (int i = 0; i < N; ++i) {
            out(i) /= norm;
        }
    }
}

}  // namespace

bool NNFeaturePointDetector::Detect(const GrayImage &image, std::vector<Vec2> &features) {
    FeaturePointHarrisDetector harris;
    harris.options().kMinFeatureDistance = options_.kMinFeatureDistance;
    harris.options().kMinValidResponse = 40.0f;  // kMinResponse is a network score in (0, 1); corners use the Harris scale
    return harris.DetectGoodFeatures(image, static_cast<uint32_t>(options_.kMaxNumberOfDetectedFeatures), features);
}

bool NNFeaturePointDetector::DetectGoodFeaturesWithDescriptor(const GrayImage &image, std::vector<Vec2> &features,
                                                              std::vector<SuperpointDescriptorType> &descriptors) {
    descriptors.clear();
    if (!Detect(image, features)) {
        return false;
    }
    descriptors.resize(features.size());
    for (size_t i = 0; i < features.size(); ++i) {
        PatchDescriptor<256>(image, features[i], 16, 16, descriptors[i]);
    }
    return true;
}

bool NNFeaturePointDetector::DetectGoodFeaturesWithDescriptor(const GrayImage &image, std::vector<Vec2> &features,
                                                              std::vector<DiskDescriptorType> &descriptors) {
    descriptors.clear();
    if (!Detect(image, features)) {
        return false;
    }
    descriptors.resize(features.size());
    for (size_t i = 0; i < features.size(); ++i) {
        PatchDescriptor<128>(image, features[i], 8, 16, descriptors[i]);
    }
    return true;
}

}  // namespace feature_detector
