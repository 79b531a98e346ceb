// direct_method_tracker.cpp — marshals DirectMethod::TrackFeatures into the C ABI.  Replaces the level
// loop and TrackAllFeaturesDirect of the reference (direct_method_tracker.cpp:35-86, :115-192) with one
// device launch; the world-frame overload (:8-33) stays host-side quaternion algebra.
#include "direct_method_tracker.h"

#include "device_runtime.h"
#include "ftk.h"
#include "slam_log_reporter.h"
#include "slam_operations.h"

namespace feature_tracker {

DirectMethod::DirectMethod() { device::WarmUp(FTK_WARM_DIRECT); }

bool DirectMethod::TrackFeatures(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::array<float, 4> &K, const Quat ref_q_wc,
                                 const Vec3 ref_p_wc, const std::vector<Vec3> &p_w, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv,
                                 Quat &cur_q_wc, Vec3 &cur_p_wc, std::vector<uint8_t> &status) {
    // Lift all points in world frame to reference camera frame.
    p_c_in_ref_.clear();
    p_c_in_ref_.reserve(p_w.size());
    const Quat ref_q_cw = ref_q_wc.inverse();
    for (const auto &pos_w : p_w) {
        p_c_in_ref_.emplace_back(ref_q_cw * (pos_w - ref_p_wc));
    }
    // T_rc = T_wr.inverse() * T_wc
    q_rc_ = ref_q_cw * cur_q_wc;
    p_rc_ = ref_q_cw * (cur_p_wc - ref_p_wc);

    RETURN_FALSE_IF_FALSE(TrackFeatures(ref_pyramid, cur_pyramid, K, p_c_in_ref_, ref_pixel_uv, cur_pixel_uv, q_rc_, p_rc_, status));

    cur_q_wc = ref_q_wc * q_rc_;
    cur_p_wc = ref_q_wc * p_rc_ + ref_p_wc;
    return true;
}

bool DirectMethod::TrackFeatures(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::array<float, 4> &K,
                                 const std::vector<Vec3> &p_c_in_ref, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv, Quat &q_rc,
                                 Vec3 &p_rc, std::vector<uint8_t> &status) {
    RETURN_FALSE_IF(ref_pixel_uv.empty());
    RETURN_FALSE_IF(cur_pyramid.level() != ref_pyramid.level());

    // If sizeof ref_pixel_uv is not equal to cur_pixel_uv, view it as no prediction.
    if (ref_pixel_uv.size() != cur_pixel_uv.size()) {
        cur_pixel_uv = ref_pixel_uv;
    }
    const uint32_t needed = ref_pixel_uv.size() < options_.kMaxTrackPointsNumber ? static_cast<uint32_t>(ref_pixel_uv.size()) : options_.kMaxTrackPointsNumber;
    if (p_c_in_ref.size() < needed) {
        last_error_ = "p_c_in_ref holds fewer points than features to track";  // the reference would read past the end
        ReportError("[DirectMethod] " << last_error_);
        return false;
    }

    last_error_.clear();
    ftk_context *ctx = device::SharedContext(&last_error_);
    if (ctx == nullptr) {
        ReportError("[DirectMethod] " << last_error_);
        return false;
    }
    ftk_pyramid *ref_dev = device::PyramidTwin(ctx, ref_pyramid, &last_error_);
    ftk_pyramid *cur_dev = ref_dev ? device::PyramidTwin(ctx, cur_pyramid, &last_error_) : nullptr;
    if (ref_dev == nullptr || cur_dev == nullptr) {
        ReportError("[DirectMethod] " << last_error_);
        return false;
    }

    ftk_direct_options opt;
    opt.max_track_points = options_.kMaxTrackPointsNumber;
    opt.max_iteration = options_.kMaxIteration;
    opt.half_rows = options_.kPatchRowHalfSize;
    opt.half_cols = options_.kPatchColHalfSize;
    opt.max_converge_step = options_.kMaxConvergeStep;
    opt.max_converge_residual = options_.kMaxConvergeResidual;
    opt.method = static_cast<int32_t>(options_.kMethod);

    const int32_t n = static_cast<int32_t>(ref_pixel_uv.size());
    // the device entry point reads 3 floats per feature: pad a shorter point list (entries beyond the cap are never used)
    const float *points = p_c_in_ref[0].data();
    std::vector<Vec3> padded;
    if (p_c_in_ref.size() < ref_pixel_uv.size()) {
        padded = p_c_in_ref;
        padded.resize(ref_pixel_uv.size());
        points = padded[0].data();
    }
    const int status_valid = status.size() == ref_pixel_uv.size() ? 1 : 0;
    if (!status_valid) {
        status.assign(ref_pixel_uv.size(), static_cast<uint8_t>(TrackStatus::kTracked));  // :73-75 (the kernel writes the same value)
    }
    float q[4] = {q_rc.w(), q_rc.x(), q_rc.y(), q_rc.z()};
    float p[3] = {p_rc.x(), p_rc.y(), p_rc.z()};
    const int rc = ftk_direct_track(ctx, &opt, ref_dev, cur_dev, K.data(), points, ref_pixel_uv[0].data(), cur_pixel_uv[0].data(), n, q, p, status.data(),
                                    status_valid, &last_iterations_);
    if (rc != FTK_OK) {
        last_error_ = ftk_last_error(ctx);
        ReportError("[DirectMethod] " << last_error_);
        return false;
    }
    q_rc = Quat(q[0], q[1], q[2], q[3]);
    p_rc = Vec3(p[0], p[1], p[2]);
    return true;
}

}  // namespace feature_tracker
