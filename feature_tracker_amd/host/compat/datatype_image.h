// datatype_image.h — stand-in for Slam_Utility's GrayImage (un-vendored).  An 8-bit, row-major,
// pitch == cols image view that may own its buffer.  The sampling rules are this repo's normative
// definitions (DESIGN.md "Substrate"): bounds-checked bilinear on the closed rectangle
// [0, rows-1] x [0, cols-1], weights from row - floor(row), col - floor(col), summed
// ((tl + tr) + bl) + br, +1 neighbour clamped to the image (its weight is exactly 0 there).
#ifndef _SLAM_UTILITY_DATATYPE_IMAGE_H_
#define _SLAM_UTILITY_DATATYPE_IMAGE_H_

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "basic_type.h"

class GrayImage {
public:
    GrayImage() = default;
    GrayImage(uint8_t *data, int32_t rows, int32_t cols, bool is_owner = false) { SetImage(data, rows, cols, is_owner); }
    GrayImage(const GrayImage &) = delete;
    GrayImage &operator=(const GrayImage &) = delete;
    GrayImage(GrayImage &&o) noexcept { *this = std::move(o); }
    GrayImage &operator=(GrayImage &&o) noexcept {
        if (this != &o) {
            Release();
            data_ = o.data_;
            rows_ = o.rows_;
            cols_ = o.cols_;
            owner_ = o.owner_;
            o.data_ = nullptr;
            o.owner_ = false;
        }
        return *this;
    }
    ~GrayImage() { Release(); }

    void SetImage(uint8_t *data, int32_t rows, int32_t cols, bool is_owner = false) {
        Release();
        data_ = data;
        rows_ = rows;
        cols_ = cols;
        owner_ = is_owner;
    }
    void SetSize(int32_t rows, int32_t cols) {
        rows_ = rows;
        cols_ = cols;
    }

    uint8_t *data() const { return data_; }
    int32_t rows() const { return rows_; }
    int32_t cols() const { return cols_; }
    bool memory_owner() const { return owner_; }

    uint8_t GetPixelValueNoCheck(int32_t row, int32_t col) const { return data_[static_cast<int64_t>(row) * cols_ + col]; }
    void SetPixelValueNoCheck(int32_t row, int32_t col, uint8_t value) { data_[static_cast<int64_t>(row) * cols_ + col] = value; }

    float GetPixelValueNoCheck(float row, float col) const {
        int32_t r0 = static_cast<int32_t>(row);
        int32_t c0 = static_cast<int32_t>(col);
        const float sub_row = row - std::floor(row);
        const float sub_col = col - std::floor(col);
        r0 = r0 < 0 ? 0 : (r0 > rows_ - 1 ? rows_ - 1 : r0);
        c0 = c0 < 0 ? 0 : (c0 > cols_ - 1 ? cols_ - 1 : c0);
        const int32_t r1 = (r0 + 1 < rows_) ? r0 + 1 : r0;
        const int32_t c1 = (c0 + 1 < cols_) ? c0 + 1 : c0;
        const float inv_sub_row = 1.0f - sub_row;
        const float inv_sub_col = 1.0f - sub_col;
        return (inv_sub_row * inv_sub_col) * static_cast<float>(GetPixelValueNoCheck(r0, c0)) +
               (inv_sub_row * sub_col) * static_cast<float>(GetPixelValueNoCheck(r0, c1)) +
               (sub_row * inv_sub_col) * static_cast<float>(GetPixelValueNoCheck(r1, c0)) +
               (sub_row * sub_col) * static_cast<float>(GetPixelValueNoCheck(r1, c1));
    }

    bool GetPixelValue(float row, float col, float *value) const {
        if (!(row >= 0.0f && col >= 0.0f && row <= static_cast<float>(rows_ - 1) && col <= static_cast<float>(cols_ - 1))) {
            return false;
        }
        *value = GetPixelValueNoCheck(row, col);
        return true;
    }

private:
    void Release() {
        if (owner_ && data_ != nullptr) {
            std::free(data_);
        }
        data_ = nullptr;
        owner_ = false;
    }

    uint8_t *data_ = nullptr;
    int32_t rows_ = 0;
    int32_t cols_ = 0;
    bool owner_ = false;
};

#endif  // _SLAM_UTILITY_DATATYPE_IMAGE_H_
