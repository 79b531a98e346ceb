// optical_flow.cpp — input normalisation of OpticalFlow::TrackFeatures (reference behaviour:
// src/optical_flow_tracker/optical_flow.cpp:6-47) and the marshalling of a call into the C ABI.
#include "optical_flow.h"

#include <memory>

#include "device_runtime.h"
#include "ftk.h"
#include "slam_log_reporter.h"
#include "slam_operations.h"

namespace feature_tracker {

namespace {

void NormaliseInOut(const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    // A cur vector of another size carries no prediction; a status vector of another size carries
    // no history (optical_flow.cpp:12-19).
    if (cur_pixel_uv.size() != ref_pixel_uv.size()) {
        cur_pixel_uv = ref_pixel_uv;
    }
    if (status.size() != ref_pixel_uv.size()) {
        status.assign(ref_pixel_uv.size(), static_cast<uint8_t>(TrackStatus::kNotTracked));
    }
}

struct PyramidDeleter {
    void operator()(void *p) const { ftk_pyramid_destroy(static_cast<ftk_pyramid *>(p)); }
};

ftk_pyramid *UploadSingle(ftk_context *ctx, const GrayImage &image, std::string *error) {
    ftk_image level;
    level.data = image.data();
    level.rows = image.rows();
    level.cols = image.cols();
    ftk_pyramid *dev = nullptr;
    if (ftk_pyramid_upload(ctx, &level, 1, &dev) != FTK_OK) {
        *error = ftk_last_error(ctx);
        return nullptr;
    }
    return dev;
}

}  // namespace

OpticalFlow::OpticalFlow() { device::WarmUp(FTK_WARM_KLT); }

bool OpticalFlow::TrackFeatures(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::vector<Vec2> &ref_pixel_uv,
                                std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    RETURN_FALSE_IF(ref_pixel_uv.empty());
    RETURN_FALSE_IF(cur_pyramid.level() != ref_pyramid.level());
    NormaliseInOut(ref_pixel_uv, cur_pixel_uv, status);
    PrepareForTracking();
    return TrackMultipleLevel(ref_pyramid, cur_pyramid, ref_pixel_uv, cur_pixel_uv, status);
}

bool OpticalFlow::TrackFeatures(const GrayImage &ref_image, const GrayImage &cur_image, const std::vector<Vec2> &ref_pixel_uv,
                                std::vector<Vec2> &cur_pixel_uv, std::vector<uint8_t> &status) {
    RETURN_FALSE_IF(ref_pixel_uv.empty());
    NormaliseInOut(ref_pixel_uv, cur_pixel_uv, status);
    PrepareForTracking();
    return TrackSingleLevel(ref_image, cur_image, ref_pixel_uv, cur_pixel_uv, status);
}

bool OpticalFlow::PrepareForTracking() {
    // Patch geometry members are part of the public surface (optical_flow.cpp:104-124).
    patch_rows_ = (options_.kPatchRowHalfSize << 1) + 1;
    patch_cols_ = (options_.kPatchColHalfSize << 1) + 1;
    patch_size_ = patch_rows_ * patch_cols_;
    ex_patch_rows_ = patch_rows_ + 2;
    ex_patch_cols_ = patch_cols_ + 2;
    ex_patch_size_ = ex_patch_rows_ * ex_patch_cols_;
    return true;
}

bool OpticalFlow::TrackOnDevice(int model, const ImagePyramid *ref_pyramid, const ImagePyramid *cur_pyramid, const GrayImage *ref_image,
                                const GrayImage *cur_image, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv,
                                std::vector<uint8_t> &status, const float *prior, bool consider_luminance) {
    last_error_.clear();
    ftk_context *ctx = device::SharedContext(&last_error_);
    if (ctx == nullptr) {
        ReportError("[OpticalFlow] " << OpticalFlowMethodName() << ": " << last_error_);
        return false;
    }

    const bool single_level = (ref_pyramid == nullptr);
    ftk_pyramid *ref_dev = nullptr;
    ftk_pyramid *cur_dev = nullptr;
    std::unique_ptr<void, PyramidDeleter> ref_guard, cur_guard;  // single-image uploads are released on return
    if (single_level) {
        ref_dev = UploadSingle(ctx, *ref_image, &last_error_);
        ref_guard.reset(ref_dev);
        cur_dev = ref_dev ? UploadSingle(ctx, *cur_image, &last_error_) : nullptr;
        cur_guard.reset(cur_dev);
    } else {
        ref_dev = device::PyramidTwin(ctx, *ref_pyramid, &last_error_);
        cur_dev = ref_dev ? device::PyramidTwin(ctx, *cur_pyramid, &last_error_) : nullptr;
    }
    if (ref_dev == nullptr || cur_dev == nullptr) {
        ReportError("[OpticalFlow] " << OpticalFlowMethodName() << ": " << last_error_);
        return false;
    }

    ftk_klt_options opt;
    opt.max_track_points = options_.kMaxTrackPointsNumber;
    opt.max_iteration = options_.kMaxIteration;
    opt.max_tolerance_large_step = options_.kMaxToleranceLargeStep;
    opt.half_rows = options_.kPatchRowHalfSize;
    opt.half_cols = options_.kPatchColHalfSize;
    opt.max_converge_step = options_.kMaxConvergeStep;
    opt.method = static_cast<int32_t>(options_.kMethod);

    const int32_t n = static_cast<int32_t>(ref_pixel_uv.size());
    last_iterations_.assign(n, 0u);
    // One rank of several (device_runtime.h, SharedComm): the feature list is sharded over the ranks' GPUs and one RCCL all-gather
    // hands every process the complete result, identical to the single-GPU one (features do not interact, basic_klt.cpp:13-54).
    std::string comm_error;
    ftk_comm *comm = device::SharedComm(ctx, &comm_error);
    if (comm == nullptr && !comm_error.empty()) {
        last_error_ = comm_error;
        ReportError("[OpticalFlow] " << OpticalFlowMethodName() << ": " << last_error_);
        return false;
    }
    const int rc = comm != nullptr
                       ? ftk_klt_track_sharded(ctx, comm, model, &opt, ref_dev, cur_dev, ref_pixel_uv[0].data(), cur_pixel_uv[0].data(), status.data(), n,
                                               prior, consider_luminance ? 1 : 0, single_level ? 1 : 0, last_iterations_.data())
                       : ftk_klt_track(ctx, model, &opt, ref_dev, cur_dev, ref_pixel_uv[0].data(), cur_pixel_uv[0].data(), status.data(), n, prior,
                                       consider_luminance ? 1 : 0, single_level ? 1 : 0, last_iterations_.data());
    if (rc != FTK_OK) {
        last_error_ = ftk_last_error(ctx);
        ReportError("[OpticalFlow] " << OpticalFlowMethodName() << ": " << last_error_);
        return false;
    }
    return true;
}

uint32_t OpticalFlow::ExtractExtendPatchInReferenceImage(const GrayImage &ref_image, const Vec2 &ref_pixel_uv, int32_t ex_ref_patch_rows,
                                                         int32_t ex_ref_patch_cols, std::vector<float> &ex_ref_patch,
                                                         std::vector<bool> &ex_ref_patch_pixel_valid) {
    // The reference appends to the two vectors (optical_flow.cpp:74-98); so does this.
    last_error_.clear();
    ftk_context *ctx = device::SharedContext(&last_error_);
    if (ctx == nullptr || ex_ref_patch_rows <= 0 || ex_ref_patch_cols <= 0) {
        ReportError("[OpticalFlow] ExtractExtendPatchInReferenceImage: " << (ctx ? "bad patch size" : last_error_));
        return 0;
    }
    std::unique_ptr<void, PyramidDeleter> guard(UploadSingle(ctx, ref_image, &last_error_));
    if (!guard) {
        ReportError("[OpticalFlow] ExtractExtendPatchInReferenceImage: " << last_error_);
        return 0;
    }
    const size_t n = static_cast<size_t>(ex_ref_patch_rows) * ex_ref_patch_cols;
    std::vector<float> patch(n);
    std::vector<uint8_t> valid(n);
    uint32_t count = 0;
    if (ftk_extract_extend_patch(ctx, static_cast<ftk_pyramid *>(guard.get()), 0, ref_pixel_uv.x(), ref_pixel_uv.y(), ex_ref_patch_rows,
                                 ex_ref_patch_cols, patch.data(), valid.data(), &count) != FTK_OK) {
        last_error_ = ftk_last_error(ctx);
        ReportError("[OpticalFlow] ExtractExtendPatchInReferenceImage: " << last_error_);
        return 0;
    }
    for (size_t i = 0; i < n; ++i) {
        ex_ref_patch.emplace_back(patch[i]);
        ex_ref_patch_pixel_valid.emplace_back(valid[i] != 0);
    }
    return count;
}

}  // namespace feature_tracker
