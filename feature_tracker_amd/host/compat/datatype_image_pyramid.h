// datatype_image_pyramid.h — stand-in for Slam_Utility's ImagePyramid (un-vendored).
// Level 0 aliases the raw image; levels >= 1 live in the side buffer handed over by
// SetPyramidBuff (test_optical_flow.cpp:49-53).  CreateImagePyramid is this repo's normative rule:
// level i+1 = truncating 2x2 box mean of level i.
//
// Where the pixels of levels >= 1 are made: CreateImagePyramid sits inside the reference's timed region
// (test/test_optical_flow.cpp:69-73), right in front of TrackFeatures, so when the device runtime is linked
// and a device is present it uploads level 0 ONCE and builds the other levels in HBM (ftk_pyramid_build:
// the pyramid's "device twin", what the trackers read).  The host copies of levels >= 1 are then filled
// lazily — downloaded from the twin — the first time a caller asks for one of those images
// (GetImage / GetImageConst with level > 0); a caller that only tracks never pays for them.  Without a
// device runtime the host loop below runs at once, as before.  Every mutation bumps generation() and drops
// the twin; so does a non-const GetImage() (the caller may write through it).
#ifndef _SLAM_UTILITY_DATATYPE_IMAGE_PYRAMID_H_
#define _SLAM_UTILITY_DATATYPE_IMAGE_PYRAMID_H_

#include <cstdint>
#include <cstdlib>
#include <memory>

#include "datatype_image.h"

class ImagePyramid;

namespace feature_tracker {
namespace device {
// Defined in src/device_runtime.cpp (part of every lib_* archive of this repo).  Weak: a program that uses the
// pyramid without any tracker links none of it and keeps the host loop.
// Builds the device twin of `pyramid` from its level 0; false when no device is usable.
bool BuildPyramidOnDevice(const ImagePyramid &pyramid) __attribute__((weak));
// Fills the host copies of levels >= 1 from the device twin; false when there is no valid twin.
bool DownloadPyramidLevels(const ImagePyramid &pyramid) __attribute__((weak));
}  // namespace device
}  // namespace feature_tracker

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
        // geometry of every level first (views into the side buffer) ...
        level_ = level;
        uint8_t *dst = buff_;
        for (uint32_t i = 1; i < level_; ++i) {
            const int32_t rows = images_[i - 1].rows() / 2, cols = images_[i - 1].cols() / 2;
            if (rows <= 0 || cols <= 0) {
                level_ = i;
                break;
            }
            images_[i].SetImage(dst, rows, cols, false);
            dst += static_cast<int64_t>(rows) * cols;
        }
        Touch();
        // ... then the pixels: in HBM when a device runtime is linked and usable, else on the host right away
        host_levels_valid_ = (level_ <= 1);
        if (!(feature_tracker::device::BuildPyramidOnDevice != nullptr && feature_tracker::device::BuildPyramidOnDevice(*this))) {
            FillHostLevels();
        }
        return true;
    }

    uint32_t level() const { return level_; }
    GrayImage &GetImage(uint32_t level_idx) {
        // the caller may write through the reference: the host copy must be complete, and the device twin is stale from here on
        EnsureHostLevels();
        Touch();
        return images_[level_idx];
    }
    const GrayImage &GetImageConst(uint32_t level_idx) const {
        if (level_idx > 0) {
            EnsureHostLevels();
        }
        return images_[level_idx];
    }
    uint8_t *data() const { return buff_; }

    // ---- device twin bookkeeping (used by the trackers; not part of the reference API) ----
    uint64_t generation() const { return generation_; }
    std::shared_ptr<void> &device_twin() const { return device_twin_; }
    uint64_t &device_twin_generation() const { return device_twin_generation_; }
    // Geometry of a level without touching (or materialising) its host pixels.
    void LevelGeometry(uint32_t level_idx, const uint8_t **data, int32_t *rows, int32_t *cols) const {
        *data = images_[level_idx].data();
        *rows = images_[level_idx].rows();
        *cols = images_[level_idx].cols();
    }
    bool host_levels_valid() const { return host_levels_valid_; }
    // Makes the host copies of levels >= 1 valid (download from the twin, else the host loop).
    void EnsureHostLevels() const {
        if (host_levels_valid_) {
            return;
        }
        if (!(feature_tracker::device::DownloadPyramidLevels != nullptr && feature_tracker::device::DownloadPyramidLevels(*this))) {
            FillHostLevels();
        }
        host_levels_valid_ = true;
    }
    // Cheap content stamp of level 0 — its address, its size and an FNV-1a hash of eight evenly spaced rows — taken when a
    // twin is made and compared before the twin is reused: level 0 aliases a caller-owned buffer, and a caller that writes
    // the next frame into the same buffer without calling SetRawImage / CreateImagePyramid again would otherwise track
    // against the previous frame's copy in HBM.  (A change confined to the unsampled rows is not seen: re-create the pyramid.)
    uint64_t ContentStamp() const {
        const GrayImage &im = images_[0];
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](uint64_t v) {
            h ^= v;
            h *= 1099511628211ull;
        };
        mix(reinterpret_cast<uintptr_t>(im.data()));
        mix(static_cast<uint64_t>(im.rows()) << 32 | static_cast<uint32_t>(im.cols()));
        if (im.data() != nullptr && im.rows() > 0 && im.cols() > 0) {
            const int32_t picks = im.rows() < 8 ? im.rows() : 8;
            for (int32_t k = 0; k < picks; ++k) {
                const int32_t r = static_cast<int32_t>((static_cast<int64_t>(im.rows() - 1) * k) / (picks > 1 ? picks - 1 : 1));
                const uint8_t *row = im.data() + static_cast<int64_t>(r) * im.cols();
                int32_t c = 0;
                for (; c + 8 <= im.cols(); c += 8) {
                    uint64_t w;
                    __builtin_memcpy(&w, row + c, 8);
                    mix(w);
                }
                for (; c < im.cols(); ++c) {
                    mix(row[c]);
                }
            }
        }
        return h;
    }
    uint64_t &device_twin_stamp() const { return device_twin_stamp_; }

private:
    void Touch() {
        ++generation_;
        device_twin_.reset();
    }
    // Levels >= 1 on the host: the normative truncating 2x2 box mean.
    void FillHostLevels() const {
        for (uint32_t i = 1; i < level_; ++i) {
            const GrayImage &src = images_[i - 1];
            const int32_t rows = images_[i].rows(), cols = images_[i].cols();
            uint8_t *dst = images_[i].data();
            for (int32_t r = 0; r < rows; ++r) {
                const uint8_t *top = src.data() + static_cast<int64_t>(2 * r) * src.cols();
                const uint8_t *bottom = top + src.cols();
                uint8_t *out = dst + static_cast<int64_t>(r) * cols;
                for (int32_t c = 0; c < cols; ++c) {
                    out[c] = static_cast<uint8_t>((static_cast<uint32_t>(top[2 * c]) + top[2 * c + 1] + bottom[2 * c] + bottom[2 * c + 1]) >> 2);
                }
            }
        }
        host_levels_valid_ = true;
    }

    GrayImage images_[kMaxLevel];
    uint8_t *buff_ = nullptr;
    bool own_buff_ = false;
    uint32_t level_ = 0;
    uint64_t generation_ = 0;
    mutable bool host_levels_valid_ = true;
    mutable std::shared_ptr<void> device_twin_;
    mutable uint64_t device_twin_generation_ = 0;
    mutable uint64_t device_twin_stamp_ = 0;
};

#endif  // _SLAM_UTILITY_DATATYPE_IMAGE_PYRAMID_H_
