// datatype_image_pyramid.h — stand-in for Slam_Utility's ImagePyramid (un-vendored).
// Level 0 aliases the raw image; levels >= 1 live in the side buffer handed over by
// SetPyramidBuff (test_optical_flow.cpp:49-53).  CreateImagePyramid is this repo's normative rule:
// level i+1 = truncating 2x2 box mean of level i.
//
// The pyramid also carries an opaque device-side twin (filled by the trackers on first use, see
// optical_flow.cpp in this directory tree) so that repeated TrackFeatures calls on the same
// pyramid upload it to HBM only once; every mutation bumps generation() and drops the twin.
#ifndef _SLAM_UTILITY_DATATYPE_IMAGE_PYRAMID_H_
#define _SLAM_UTILITY_DATATYPE_IMAGE_PYRAMID_H_

#include <cstdint>
#include <cstdlib>
#include <memory>

#include "datatype_image.h"

class ImagePyramid {
public:
    static constexpr uint32_t kMaxLevel = 12;

    ImagePyramid() = default;
    ImagePyramid(const ImagePyramid &) = delete;
    ImagePyramid &operator=(const ImagePyramid &) = delete;
    ~ImagePyramid() {
        if (own_buff_ && buff_ != nullptr) {
            std::free(buff_);
        }
    }

    void SetPyramidBuff(uint8_t *buff, bool is_owner = false) {
        if (own_buff_ && buff_ != nullptr) {
            std::free(buff_);
        }
        buff_ = buff;
        own_buff_ = is_owner;
        Touch();
    }
    void SetRawImage(uint8_t *image_data, int32_t rows, int32_t cols) {
        images_[0].SetImage(image_data, rows, cols, false);
        Touch();
    }

    bool CreateImagePyramid(uint32_t level) {
        if (images_[0].data() == nullptr || buff_ == nullptr || level == 0 || level > kMaxLevel) {
            return false;
        }
        level_ = level;
        uint8_t *dst = buff_;
        for (uint32_t i = 1; i < level_; ++i) {
            const GrayImage &src = images_[i - 1];
            const int32_t rows = src.rows() / 2, cols = src.cols() / 2;
            if (rows <= 0 || cols <= 0) {
                level_ = i;
                break;
            }
            images_[i].SetImage(dst, rows, cols, false);
            for (int32_t r = 0; r < rows; ++r) {
                const uint8_t *top = src.data() + static_cast<int64_t>(2 * r) * src.cols();
                const uint8_t *bottom = top + src.cols();
                uint8_t *out = dst + static_cast<int64_t>(r) * cols;
                for (int32_t c = 0; c < cols; ++c) {
                    out[c] = static_cast<uint8_t>((static_cast<uint32_t>(top[2 * c]) + top[2 * c + 1] + bottom[2 * c] + bottom[2 * c + 1]) >> 2);
                }
            }
            dst += static_cast<int64_t>(rows) * cols;
        }
        Touch();
        return true;
    }

    uint32_t level() const { return level_; }
    GrayImage &GetImage(uint32_t level_idx) { return images_[level_idx]; }
    const GrayImage &GetImageConst(uint32_t level_idx) const { return images_[level_idx]; }
    uint8_t *data() const { return buff_; }

    // device twin bookkeeping (used by the trackers; not part of the reference API)
    uint64_t generation() const { return generation_; }
    std::shared_ptr<void> &device_twin() const { return device_twin_; }
    uint64_t &device_twin_generation() const { return device_twin_generation_; }

private:
    void Touch() {
        ++generation_;
        device_twin_.reset();
    }

    GrayImage images_[kMaxLevel];
    uint8_t *buff_ = nullptr;
    bool own_buff_ = false;
    uint32_t level_ = 0;
    uint64_t generation_ = 0;
    mutable std::shared_ptr<void> device_twin_;
    mutable uint64_t device_twin_generation_ = 0;
};

#endif  // _SLAM_UTILITY_DATATYPE_IMAGE_PYRAMID_H_
