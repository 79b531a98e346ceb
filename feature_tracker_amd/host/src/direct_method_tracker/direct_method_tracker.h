// direct_method_tracker.h — feature_tracker::DirectMethod with the reference's surface
// (src/direct_method_tracker/direct_method_tracker.h:14-80): options(), the world-frame and the
// camera-frame TrackFeatures overloads, the private virtual TrackSingleLevel plug-in point.
//
// Where the work runs: the whole coarse-to-fine Gauss-Newton alignment of the camera-frame overload is
// ONE launch on the MI355X (ftk_direct_track, include/ftk.h); the world-frame overload is the
// reference's quaternion algebra around it (direct_method_tracker.cpp:8-33).  A subclass that
// overrides TrackSingleLevel keeps the reference's level loop on the host and is called per level.
#ifndef _DIRECT_METHOD_TRACKER_H_
#define _DIRECT_METHOD_TRACKER_H_

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "basic_type.h"
#include "datatype_image.h"
#include "datatype_image_pyramid.h"
#include "feature_tracker.h"
#include "slam_basic_math.h"

namespace feature_tracker {

enum DirectMethodMethod : uint8_t {
    kInverse = 0,
    kDirect = 1,
    kFast = 2,
};

struct DirectMethodOptions {
    uint32_t kMaxTrackPointsNumber = 500;
    uint32_t kMaxIteration = 15;
    int32_t kPatchRowHalfSize = 6;
    int32_t kPatchColHalfSize = 6;
    float kMaxConvergeStep = 1e-6f;
    float kMaxConvergeResidual = 2.0f;
    DirectMethodMethod kMethod = kDirect;
};

class DirectMethod {

public:
    DirectMethod();  // = default in the reference; also the first-use device cost (device_runtime.h, WarmUp)
    virtual ~DirectMethod() = default;
    DirectMethod(const DirectMethod &direct_method) = delete;

    bool TrackFeatures(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::array<float, 4> &K, const Quat ref_q_wc,
                       const Vec3 ref_p_wc, const std::vector<Vec3> &p_w, const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv,
                       Quat &cur_q_wc, Vec3 &cur_p_wc, std::vector<uint8_t> &status);

    bool TrackFeatures(const ImagePyramid &ref_pyramid, const ImagePyramid &cur_pyramid, const std::array<float, 4> &K, const std::vector<Vec3> &p_c_in_ref,
                       const std::vector<Vec2> &ref_pixel_uv, std::vector<Vec2> &cur_pixel_uv, Quat &q_rc, Vec3 &p_rc, std::vector<uint8_t> &status);

    // Reference for member variables.
    DirectMethodOptions &options() { return options_; }

    // Const reference for member variables.
    const DirectMethodOptions &options() const { return options_; }

    // Not part of the reference API: Gauss-Newton iterations of the last call (all levels), last failure text.
    uint32_t last_iterations() const { return last_iterations_; }
    const std::string &last_error() const { return last_error_; }

private:
    DirectMethodOptions options_;

    // Points position in ref frame.
    std::vector<Vec3> p_c_in_ref_ = {};

    // Current frame pose in reference frame.
    Quat q_rc_ = Quat::Identity();
    Vec3 p_rc_ = Vec3::Zero();

    uint32_t last_iterations_ = 0;
    std::string last_error_;
};

}  // namespace feature_tracker

#endif  // end of _DIRECT_METHOD_TRACKER_H_
