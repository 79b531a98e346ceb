// device_runtime.h — process-wide access to the MI355X runtime behind the host-side classes.
// Not part of the reference API.  One ftk_context per process (device from $FTK_DEVICE, default 0).
// Threading: a tracker / matcher OBJECT is not thread-safe (neither is the reference's: optical_flow.h:91-111 keeps
// mutable scratch in the object), but separate objects may be used from separate threads at once, as in the reference —
// every ftk_* call on the shared context is serialised inside the C ABI (include/ftk.h, Conventions) and the device-twin
// bookkeeping of a pyramid that several trackers read is guarded here.
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
// Hooks of ImagePyramid (compat/datatype_image_pyramid.h, declared weak there): CreateImagePyramid builds levels >= 1 in HBM
// from ONE upload of level 0; the host copies of those levels are downloaded only if a caller reads them.
bool BuildPyramidOnDevice(const ImagePyramid &pyramid);
bool DownloadPyramidLevels(const ImagePyramid &pyramid);

}  // namespace device
}  // namespace feature_tracker

#endif
