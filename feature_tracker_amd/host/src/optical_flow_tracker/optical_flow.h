// optical_flow.h — abstract sparse optical-flow tracker with the reference's public surface
// (src/optical_flow_tracker/optical_flow.h:20-112): options struct, the two TrackFeatures overloads,
// the public extended-patch extractor and the scratch accessors.  The numerical work of every
// subclass runs on the MI355X through the C ABI (include/ftk.h); this layer only normalises the
// caller's vectors and marshals pointers.  If no HIP device is usable TrackFeatures reports the
// error and returns false — there is no CPU path.
#ifndef _OPTICAL_FLOW_TRACKER_H_
#define _OPTICAL_FLOW_TRACKER_H_

#include <string>
#include <vector>

#include "basic_type.h"
#include "datatype_image.h"
#include "datatype_image_pyramid.h"
#include "feature_tracker.h"
#include "slam_basic_math.h"

namespace feature_tracker {

enum class OpticalFlowMethod : uint8_t {
    kInverse = 0,
    kDirect = 1,
    kFast = 2,
    kSse = 3,   // handled like kFast, as in the reference's `default:` branches
    kNeon = 4,  // idem
};

struct OpticalFlowOptions {
    uint32_t kMaxTrackPointsNumber = 500;
    uint32_t kMaxIteration = 15;
    uint32_t kMaxToleranceLargeStep = 3;
    int32_t kPatchRowHalfSize = 6;
    int32_t kPatchColHalfSize = 6;
    float kMaxConvergeStep = 4e-2f;
    OpticalFlowMethod kMethod = OpticalFlowMethod::kFast;
};

class OpticalFlow {

public:
    OpticalFlow();  // = default in the reference; here it also takes the first-use device cost out of the first TrackFeatures (device_runtime.h, WarmUp)
    virtual ~OpticalFlow() = default;

    virtual std::string OpticalFlowMethodName() const { return "None"; }

    bool TrackFeatures(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv,
                       std::vector<uint8_t> &status);

    bool TrackFeatures(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv,
                       std::vector<uint8_t> &status);

    // Shared helper of the fast variants, public in the reference too.
    uint32_t ExtractExtendPatchInReferenceImage(const GrayImage &ref_image, const Vec2 &ref_pixel_uv, int32_t ex_ref_patch_rows, int32_t ex_ref_patch_cols,
                                                std::vector<float> &ex_ref_patch, std::vector<bool> &ex_ref_patch_pixel_valid);

    // Reference for member variables.
    OpticalFlowOptions &options() { return options_; }
    std::vector<float> &ex_ref_patch() { return ex_ref_patch_; }
    std::vector<bool> &ex_ref_patch_pixel_valid() { return ex_ref_patch_pixel_valid_; }
    std::vector<float> &all_dx_in_ref_patch() { return all_dx_in_ref_patch_; }
    std::vector<float> &all_dy_in_ref_patch() { return all_dy_in_ref_patch_; }
    std::vector<float> &cur_patch() { return cur_patch_; }
    std::vector<bool> &cur_patch_pixel_valid() { return cur_patch_pixel_valid_; }
    std::vector<float> &all_dx_in_cur_patch() { return all_dx_in_cur_patch_; }
    std::vector<float> &all_dy_in_cur_patch() { return all_dy_in_cur_patch_; }
    int32_t &patch_rows() { return patch_rows_; }
    int32_t &patch_cols() { return patch_cols_; }
    int32_t &patch_size() { return patch_size_; }
    int32_t &ex_ref_patch_rows() { return ex_patch_rows_; }
    int32_t &ex_ref_patch_cols() { return ex_patch_cols_; }
    int32_t &ex_patch_size() { return ex_patch_size_; }

    // Const reference for member variables.
    const OpticalFlowOptions &options() const { return options_; }
    const std::vector<float> &ex_ref_patch() const { return ex_ref_patch_; }
    const std::vector<bool> &ex_ref_patch_pixel_valid() const { return ex_ref_patch_pixel_valid_; }
    const std::vector<float> &all_dx_in_ref_patch() const { return all_dx_in_ref_patch_; }
    const std::vector<float> &all_dy_in_ref_patch() const { return all_dy_in_ref_patch_; }
    const std::vector<float> &cur_patch() const { return cur_patch_; }
    const std::vector<bool> &cur_patch_pixel_valid() const { return cur_patch_pixel_valid_; }
    const std::vector<float> &all_dx_in_cur_patch() const { return all_dx_in_cur_patch_; }
    const std::vector<float> &all_dy_in_cur_patch() const { return all_dy_in_cur_patch_; }
    const int32_t &patch_rows() const { return patch_rows_; }
    const int32_t &patch_cols() const { return patch_cols_; }
    const int32_t &patch_size() const { return patch_size_; }
    const int32_t &ex_ref_patch_rows() const { return ex_patch_rows_; }
    const int32_t &ex_ref_patch_cols() const { return ex_patch_cols_; }
    const int32_t &ex_patch_size() const { return ex_patch_size_; }

    // Extensions (not in the reference): Gauss-Newton iterations per feature of the last call
    // (summed over levels) and the text of the last device failure.
    const std::vector<uint32_t> &last_iterations() const { return last_iterations_; }
    const std::string &last_error() const { return last_error_; }

protected:
    // Marshals one TrackFeatures call to ftk_klt_track.  model: FTK_MODEL_*; prior: row-major 2x2 or
    // nullptr; exactly one of (pyramids, images) is non-null.
    bool TrackOnDevice(int model, const ImagePyramid *ref_pyramid, const ImagePyramid *cur_pyramid, const GrayImage *ref_image,
                       const GrayImage *cur_image, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv,
                       std::vector<uint8_t> &status, const float *prior, bool consider_luminance);

private:
    virtual bool TrackMultipleLevel(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                    std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) = 0;
    virtual bool TrackSingleLevel(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                  std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) = 0;
    virtual bool PrepareForTracking();

private:
    OpticalFlowOptions options_;

    // Host-side scratch kept for API compatibility (the device path keeps its scratch in LDS).
    std::vector<float> ex_ref_patch_;
    std::vector<bool> ex_ref_patch_pixel_valid_;
    std::vector<float> all_dx_in_ref_patch_;
    std::vector<float> all_dy_in_ref_patch_;
    std::vector<float> cur_patch_;
    std::vector<bool> cur_patch_pixel_valid_;
    std::vector<float> all_dx_in_cur_patch_;
    std::vector<float> all_dy_in_cur_patch_;

    int32_t patch_rows_ = 0;
    int32_t patch_cols_ = 0;
    int32_t patch_size_ = 0;
    int32_t ex_patch_rows_ = 0;
    int32_t ex_patch_cols_ = 0;
    int32_t ex_patch_size_ = 0;

    std::vector<uint32_t> last_iterations_;
    std::string last_error_;
};

}  // namespace feature_tracker

#endif  // _OPTICAL_FLOW_TRACKER_H_
