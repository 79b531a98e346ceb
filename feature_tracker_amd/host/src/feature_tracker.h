// feature_tracker.h — status codes shared by every tracker (mirrors src/feature_tracker.h:8-14 of the
// reference so that callers compile unchanged).
#ifndef _FEATURE_TRACKER_H_
#define _FEATURE_TRACKER_H_

#include "basic_type.h"

namespace feature_tracker {

enum class TrackStatus : uint8_t {
    kNotTracked = 0,
    kTracked = 1,
    kLargeResidual = 2,
    kOutside = 3,
    kNumericError = 4,
};

}  // namespace feature_tracker

#endif  // _FEATURE_TRACKER_H_
