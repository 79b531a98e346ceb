// device_runtime.h — process-wide access to the MI355X runtime behind the host-side classes.
// Not part of the reference API.  One ftk_context per process (device from $FTK_DEVICE, default 0);
// the classes above are not thread-safe, exactly like the reference's.
#ifndef _FEATURE_TRACKER_DEVICE_RUNTIME_H_
#define _FEATURE_TRACKER_DEVICE_RUNTIME_H_

#include <string>

#include "datatype_image_pyramid.h"
#include "ftk.h"

namespace feature_tracker {
namespace device {

// Returns the shared context, creating it on first use; nullptr (and *error filled) when no HIP
// device is usable.  There is no CPU fallback: callers report the error and return false.
ftk_context *SharedContext(std::string *error);
// Text of the last failure on the shared context.
std::string LastError();
// Device twin of a host ImagePyramid: uploaded once per generation of the host object, then reused by
// every tracker that is handed the same pyramid (owned by the pyramid, released with it).
ftk_pyramid *PyramidTwin(ftk_context *ctx, const ImagePyramid &pyramid, std::string *error);

}  // namespace device
}  // namespace feature_tracker

#endif
