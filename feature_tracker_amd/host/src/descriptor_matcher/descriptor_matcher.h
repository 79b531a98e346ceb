// descriptor_matcher.h — DescriptorMatcher<DescriptorType> with the reference's surface
// (src/descriptor_matcher/descriptor_matcher.h:12-52): ForceMatch / NearbyMatch in index- and
// pixel-returning overloads over a caller-supplied virtual ComputeDistance.
//
// Where the work runs:
//   * DescriptorType == std::vector<bool> (the per-bit BRIEF container of
//     test/test_descriptor_matcher_brief.cpp) and the caller's ComputeDistance IS the Hamming
//     distance (probed on a few pairs at call time): the all-pairs loop runs on the MI355X
//     (ftk_hamming_match: bit-packed descriptors, popcount kernel).  A device failure is reported
//     and the call returns false; the host loop is NOT used as a fallback for this case.
//   * DescriptorType is a fixed-size float vector (the SuperPoint-256 / DISK-128 descriptors of
//     test/test_descriptor_matcher_{superpoint,disk}.cpp) and the caller's ComputeDistance IS the
//     cosine distance 0.5f - a.dot(b) / a.norm() / b.norm() * 0.5f (probed bit for bit on a few
//     pairs): ftk_cosine_match — fp16 MFMA shortlist, exact fp32 decision on the device.
//   * any other descriptor type / distance: the distance is arbitrary caller code behind a virtual,
//     so the double loop below runs on the host exactly as written in the reference.  The recognition
//     is checked again on the pairs the device returned (Options::kAllowDeviceOffload explains it and
//     is the switch to turn the offload off).
#ifndef _DESCRIPTOR_MATCHER_H_
#define _DESCRIPTOR_MATCHER_H_

#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>

#include "basic_type.h"
#include "feature_tracker.h"
#include "slam_basic_math.h"
#include "slam_operations.h"

namespace feature_tracker {

namespace device {
// Defined in descriptor_matcher.cpp (lib_descriptor_matcher).  pred_uv == nullptr selects ForceMatch.
// Returns false on a device failure (already reported).
bool HammingMatch(const std::vector<std::vector<bool>> &descriptors_ref, const std::vector<std::vector<bool>> &descriptors_cur,
                  const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur, float max_distance,
                  int32_t max_col_distance, int32_t max_row_distance, std::vector<int32_t> &index_pairs_in_cur);
// First-use device cost (context, code objects, workspaces) at construction time instead of inside the first match
// (device_runtime.h, WarmUp): hamming != 0 for per-bit descriptors, else the float (cosine) matcher's kernels.
void WarmUpMatcher(bool hamming);
// Plain Hamming distance of two per-bit descriptors, used for the call-time probe.
float HammingDistance(const std::vector<bool> &a, const std::vector<bool> &b);
// Float descriptors stored back to back (n x dim floats).  pred_uv == nullptr selects ForceMatch.
bool CosineMatch(const float *descriptors_ref, int32_t n_ref, const float *descriptors_cur, int32_t n_cur, int32_t dim,
                 const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur, float max_distance,
                 int32_t max_col_distance, int32_t max_row_distance, std::vector<int32_t> &index_pairs_in_cur);
}  // namespace device

namespace detail {
// FixedMat<N, 1>-like: contiguous floats, nothing else in the object, dot() / norm() members.
template <typename T, typename = void>
struct IsFloatVector : std::false_type {};
template <typename T>
struct IsFloatVector<T, std::void_t<decltype(std::declval<const T &>().data()), decltype(std::declval<const T &>().dot(std::declval<const T &>())),
                                    decltype(std::declval<const T &>().norm()), decltype(T::size())>>
    : std::integral_constant<bool, std::is_same<decltype(std::declval<const T &>().data()), const float *>::value &&
                                       sizeof(T) == sizeof(float) * static_cast<size_t>(T::size())> {};
}  // namespace detail

/* Class Descriptor Matcher Declaration. */
template <typename DescriptorType>
class DescriptorMatcher {

public:
    struct Options {
        int32_t kMaxValidPredictRowDistance = 40;
        int32_t kMaxValidPredictColDistance = 40;
        float kMaxValidDescriptorDistance = 0.0f;
        // Not in the reference.  The distance is caller code behind a virtual, which the device cannot run: the all-pairs
        // scan is offloaded only when that code is RECOGNISED as the Hamming / cosine distance — it must agree, bit for bit,
        // with the built-in definition on a few probe pairs before the call and on a sample of the pairs the device
        // returned (plus random ones) after it; any disagreement restores the indices and runs the host loop over the
        // virtual, as the reference does.  A distance that differs from the built-in one only on inputs none of those
        // checks meets is not detected: set this to false to always honour the virtual on the host.
        bool kAllowDeviceOffload = true;
    };

public:
    DescriptorMatcher() {  // = default in the reference
        if (std::is_same<DescriptorType, std::vector<bool>>::value) {
            device::WarmUpMatcher(true);
        } else if (detail::IsFloatVector<DescriptorType>::value) {
            device::WarmUpMatcher(false);
        }
    }
    virtual ~DescriptorMatcher() = default;

    bool ForceMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                    std::vector<int32_t> &index_pairs_in_cur);

    bool ForceMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                    const std::vector<Vec2> &pixel_uv_cur, std::vector<Vec2> &matched_pixel_uv_cur, std::vector<uint8_t> &status);

    bool NearbyMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                     const std::vector<Vec2> &pixel_uv_pred_in_cur, const std::vector<Vec2> &pixel_uv_cur, std::vector<int32_t> &index_pairs_in_cur);

    bool NearbyMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                     const std::vector<Vec2> &pixel_uv_pred_in_cur, const std::vector<Vec2> &pixel_uv_cur, std::vector<Vec2> &matched_pixel_uv_cur,
                     std::vector<uint8_t> &status);

    // Reference for member variables.
    Options &options() { return options_; }
    // Const reference for member variables.
    const Options &options() const { return options_; }

private:
    virtual float ComputeDistance(const DescriptorType &descriptor_ref, const DescriptorType &descriptor_cur) = 0;

    bool FillMatchedPixelByPairIndices(const std::vector<int32_t> &index_pairs_in_cur, const std::vector<Vec2> &pixel_uv_cur,
                                       std::vector<Vec2> &matched_pixel_uv_cur, std::vector<uint8_t> &status);

    // True when DescriptorType is the per-bit container and ComputeDistance agrees with Hamming on a
    // handful of pairs spread over the inputs.
    bool DistanceIsHamming(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur);

    // True when DescriptorType is a packed float vector and ComputeDistance agrees bit for bit with
    // the cosine distance of the reference's SuperPoint / DISK matchers on a handful of pairs.
    bool DistanceIsCosine(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur);

    // After an offloaded call: the caller's ComputeDistance must equal `builtin` (bitwise, or both NaN) on up to 48 of the
    // returned pairs, evenly spread, and on 16 pseudo-random pairs.
    template <typename Builtin>
    bool OffloadAgreesWithVirtual(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                                  const std::vector<int32_t> &index_pairs_in_cur, Builtin &&builtin);

    // The reference's double loop over the virtual distance (pred == nullptr: no window test).
    void HostLoop(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                  const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur, std::vector<int32_t> &index_pairs_in_cur);

    // Shared body of ForceMatch / NearbyMatch (pred == nullptr: no window test).
    bool MatchIndices(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                      const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur,
                      std::vector<int32_t> &index_pairs_in_cur);

private:
    Options options_;
};

/* Class Descriptor Matcher Definition. */
template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::DistanceIsHamming(const std::vector<DescriptorType> &descriptors_ref,
                                                          const std::vector<DescriptorType> &descriptors_cur) {
    if constexpr (std::is_same<DescriptorType, std::vector<bool>>::value) {
        const size_t n_ref = descriptors_ref.size(), n_cur = descriptors_cur.size();
        if (n_ref == 0 || n_cur == 0) {
            return true;
        }
        const size_t bits = descriptors_ref[0].size();
        for (const auto &d : descriptors_ref) {
            RETURN_FALSE_IF(d.size() != bits);
        }
        for (const auto &d : descriptors_cur) {
            RETURN_FALSE_IF(d.size() != bits);
        }
        const size_t probes = 6;
        for (size_t k = 0; k < probes; ++k) {
            const auto &a = descriptors_ref[(k * 7919u + 1u) % n_ref];
            const auto &b = descriptors_cur[(k * 104729u + 3u) % n_cur];
            RETURN_FALSE_IF(ComputeDistance(a, b) != device::HammingDistance(a, b));
        }
        return true;
    } else {
        (void)descriptors_ref;
        (void)descriptors_cur;
        return false;
    }
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::DistanceIsCosine(const std::vector<DescriptorType> &descriptors_ref,
                                                         const std::vector<DescriptorType> &descriptors_cur) {
    if constexpr (detail::IsFloatVector<DescriptorType>::value) {
        const size_t n_ref = descriptors_ref.size(), n_cur = descriptors_cur.size();
        if (n_ref == 0 || n_cur == 0) {
            return true;
        }
        const size_t probes = 6;
        for (size_t k = 0; k < probes; ++k) {
            const DescriptorType &a = descriptors_ref[(k * 7919u + 1u) % n_ref];
            const DescriptorType &b = descriptors_cur[(k * 104729u + 3u) % n_cur];
            const float expect = 0.5f - a.dot(b) / a.norm() / b.norm() * 0.5f;
            const float got = ComputeDistance(a, b);
            // bitwise equal, or both NaN (a zero descriptor)
            RETURN_FALSE_IF(!(got == expect || (got != got && expect != expect)));
        }
        return true;
    } else {
        (void)descriptors_ref;
        (void)descriptors_cur;
        return false;
    }
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::MatchIndices(const std::vector<DescriptorType> &descriptors_ref,
                                                     const std::vector<DescriptorType> &descriptors_cur,
                                                     const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur,
                                                     std::vector<int32_t> &index_pairs_in_cur) {
    // Entries are reset only when the caller's vector has the wrong size, so stale indices survive a
    // call in which nothing beats the threshold (reference behaviour, descriptor_matcher.h:60-62).
    if (descriptors_ref.size() != index_pairs_in_cur.size()) {
        index_pairs_in_cur.assign(descriptors_ref.size(), -1);
    }

    if constexpr (std::is_same<DescriptorType, std::vector<bool>>::value) {
        if (options_.kAllowDeviceOffload && DistanceIsHamming(descriptors_ref, descriptors_cur)) {
            const std::vector<int32_t> incoming = index_pairs_in_cur;
            RETURN_FALSE_IF_FALSE(device::HammingMatch(descriptors_ref, descriptors_cur, pixel_uv_pred_in_cur, pixel_uv_cur,
                                                       options_.kMaxValidDescriptorDistance, options_.kMaxValidPredictColDistance,
                                                       options_.kMaxValidPredictRowDistance, index_pairs_in_cur));
            if (OffloadAgreesWithVirtual(descriptors_ref, descriptors_cur, index_pairs_in_cur,
                                         [](const DescriptorType &a, const DescriptorType &b) { return device::HammingDistance(a, b); })) {
                return true;
            }
            index_pairs_in_cur = incoming;  // the virtual is not the Hamming distance after all: honour it on the host
        }
    }

    if constexpr (detail::IsFloatVector<DescriptorType>::value) {
        if (options_.kAllowDeviceOffload && DistanceIsCosine(descriptors_ref, descriptors_cur)) {
            if (descriptors_ref.empty()) {
                return true;
            }
            const std::vector<int32_t> incoming = index_pairs_in_cur;
            RETURN_FALSE_IF_FALSE(device::CosineMatch(descriptors_ref[0].data(), static_cast<int32_t>(descriptors_ref.size()), descriptors_cur[0].data(),
                                                      static_cast<int32_t>(descriptors_cur.size()), static_cast<int32_t>(DescriptorType::size()),
                                                      pixel_uv_pred_in_cur, pixel_uv_cur, options_.kMaxValidDescriptorDistance,
                                                      options_.kMaxValidPredictColDistance, options_.kMaxValidPredictRowDistance, index_pairs_in_cur));
            if (OffloadAgreesWithVirtual(descriptors_ref, descriptors_cur, index_pairs_in_cur, [](const DescriptorType &a, const DescriptorType &b) {
                    return 0.5f - a.dot(b) / a.norm() / b.norm() * 0.5f;
                })) {
                return true;
            }
            index_pairs_in_cur = incoming;
        }
    }

    HostLoop(descriptors_ref, descriptors_cur, pixel_uv_pred_in_cur, pixel_uv_cur, index_pairs_in_cur);
    return true;
}

template <typename DescriptorType>
template <typename Builtin>
bool DescriptorMatcher<DescriptorType>::OffloadAgreesWithVirtual(const std::vector<DescriptorType> &descriptors_ref,
                                                                 const std::vector<DescriptorType> &descriptors_cur,
                                                                 const std::vector<int32_t> &index_pairs_in_cur, Builtin &&builtin) {
    const size_t n_ref = descriptors_ref.size(), n_cur = descriptors_cur.size();
    if (n_ref == 0 || n_cur == 0) {
        return true;
    }
    auto same = [&](size_t i, size_t j) {
        const float got = ComputeDistance(descriptors_ref[i], descriptors_cur[j]);
        const float expect = builtin(descriptors_ref[i], descriptors_cur[j]);
        return got == expect || (got != got && expect != expect);
    };
    const size_t stride = (n_ref + 47) / 48;  // at most 48 rows
    for (size_t i = 0; i < n_ref; i += stride) {
        const int32_t j = index_pairs_in_cur[i];
        if (j >= 0 && static_cast<size_t>(j) < n_cur) {
            RETURN_FALSE_IF(!same(i, static_cast<size_t>(j)));
        }
    }
    uint32_t state = 0x9E3779B9u ^ static_cast<uint32_t>(n_ref * 2654435761u + n_cur);
    for (int k = 0; k < 16; ++k) {
        state = state * 1664525u + 1013904223u;
        const size_t i = (state >> 8) % n_ref;
        state = state * 1664525u + 1013904223u;
        const size_t j = (state >> 8) % n_cur;
        RETURN_FALSE_IF(!same(i, j));
    }
    return true;
}

template <typename DescriptorType>
void DescriptorMatcher<DescriptorType>::HostLoop(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                                                 const std::vector<Vec2> *pixel_uv_pred_in_cur, const std::vector<Vec2> *pixel_uv_cur,
                                                 std::vector<int32_t> &index_pairs_in_cur) {
    // Generic host loop over the caller's virtual distance: strict '<' against a running minimum
    // that starts at the threshold, so the lowest index wins ties.
    const size_t n_ref = descriptors_ref.size(), n_cur = descriptors_cur.size();
    const float threshold = options_.kMaxValidDescriptorDistance;
    for (size_t i = 0; i < n_ref; ++i) {
        float best = threshold;
        for (size_t j = 0; j < n_cur; ++j) {
            if (pixel_uv_pred_in_cur != nullptr) {
                const Vec2 &p = (*pixel_uv_pred_in_cur)[i];
                const Vec2 &c = (*pixel_uv_cur)[j];
                CONTINUE_IF(std::fabs(p.x() - c.x()) > options_.kMaxValidPredictColDistance ||
                            std::fabs(p.y() - c.y()) > options_.kMaxValidPredictRowDistance);
            }
            const float distance = ComputeDistance(descriptors_ref[i], descriptors_cur[j]);
            if (distance < best && distance < threshold) {
                best = distance;
                index_pairs_in_cur[i] = static_cast<int32_t>(j);
            }
            BREAK_IF(pixel_uv_pred_in_cur != nullptr && distance == 0);
        }
    }
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::ForceMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                                                   std::vector<int32_t> &index_pairs_in_cur) {
    RETURN_FALSE_IF(descriptors_cur.empty());
    return MatchIndices(descriptors_ref, descriptors_cur, nullptr, nullptr, index_pairs_in_cur);
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::ForceMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                                                   const std::vector<Vec2> &pixel_uv_cur, std::vector<Vec2> &matched_pixel_uv_cur,
                                                   std::vector<uint8_t> &status) {
    std::vector<int32_t> index_pairs_in_cur;
    RETURN_FALSE_IF_FALSE(ForceMatch(descriptors_ref, descriptors_cur, index_pairs_in_cur));
    return FillMatchedPixelByPairIndices(index_pairs_in_cur, pixel_uv_cur, matched_pixel_uv_cur, status);
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::NearbyMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                                                    const std::vector<Vec2> &pixel_uv_pred_in_cur, const std::vector<Vec2> &pixel_uv_cur,
                                                    std::vector<int32_t> &index_pairs_in_cur) {
    RETURN_FALSE_IF(descriptors_cur.empty());
    RETURN_FALSE_IF(descriptors_ref.size() != pixel_uv_pred_in_cur.size());
    RETURN_FALSE_IF(descriptors_cur.size() != pixel_uv_cur.size());
    return MatchIndices(descriptors_ref, descriptors_cur, &pixel_uv_pred_in_cur, &pixel_uv_cur, index_pairs_in_cur);
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::NearbyMatch(const std::vector<DescriptorType> &descriptors_ref, const std::vector<DescriptorType> &descriptors_cur,
                                                    const std::vector<Vec2> &pixel_uv_pred_in_cur, const std::vector<Vec2> &pixel_uv_cur,
                                                    std::vector<Vec2> &matched_pixel_uv_cur, std::vector<uint8_t> &status) {
    std::vector<int32_t> index_pairs_in_cur;
    RETURN_FALSE_IF_FALSE(NearbyMatch(descriptors_ref, descriptors_cur, pixel_uv_pred_in_cur, pixel_uv_cur, index_pairs_in_cur));
    return FillMatchedPixelByPairIndices(index_pairs_in_cur, pixel_uv_cur, matched_pixel_uv_cur, status);
}

template <typename DescriptorType>
bool DescriptorMatcher<DescriptorType>::FillMatchedPixelByPairIndices(const std::vector<int32_t> &index_pairs_in_cur, const std::vector<Vec2> &pixel_uv_cur,
                                                                      std::vector<Vec2> &matched_pixel_uv_cur, std::vector<uint8_t> &status) {
    // index -> pixel gather (reference behaviour: descriptor_matcher.h:135-157)
    if (index_pairs_in_cur.size() != status.size()) {
        status.assign(index_pairs_in_cur.size(), static_cast<uint8_t>(TrackStatus::kNotTracked));
    }
    matched_pixel_uv_cur.resize(index_pairs_in_cur.size());
    for (size_t ref_id = 0; ref_id < index_pairs_in_cur.size(); ++ref_id) {
        CONTINUE_IF(status[ref_id] > static_cast<uint8_t>(TrackStatus::kTracked));
        const int32_t cur_id = index_pairs_in_cur[ref_id];
        if (cur_id >= 0 && static_cast<size_t>(cur_id) < pixel_uv_cur.size()) {
            matched_pixel_uv_cur[ref_id] = pixel_uv_cur[cur_id];
            status[ref_id] = static_cast<uint8_t>(TrackStatus::kTracked);
        } else {
            status[ref_id] = static_cast<uint8_t>(TrackStatus::kLargeResidual);
        }
    }
    return true;
}

}  // namespace feature_tracker

#endif  // _DESCRIPTOR_MATCHER_H_
