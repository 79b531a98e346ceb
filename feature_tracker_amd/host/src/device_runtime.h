// device_runtime.h — process-wide access to the MI355X runtime behind the host-side classes.
// Not part of the reference API.  One ftk_context per process (device from $FTK_DEVICE, default 0);
// the classes above are not thread-safe, exactly like the reference's.
#ifndef _FEATURE_TRACKER_DEVICE_RUNTIME_H_
#define _FEATURE_TRACKER_DEVICE_RUNTIME_H_

#include <string>

#include "ftk.h"

namespace feature_tracker {
namespace device {

// Returns the shared context, creating it on first use; nullptr (and *error filled) when no HIP
// device is usable.  There is no CPU fallback: callers report the error and return false.
ftk_context *SharedContext(std::string *error);
// Text of the last failure on the shared context.
std::string LastError();

}  // namespace device
}  // namespace feature_tracker

#endif
