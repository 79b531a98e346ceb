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
// One process per GPU: when FTK_WORLD_SIZE (default 1) says this process is one rank of several — or FTK_COMM_ID_FILE is set,
// which exercises the same path at world size 1 — the trackers shard their feature list over the ranks and every process
// receives the complete result (ftk_klt_track_sharded: RCCL all-gather issued by libftk_hip.so).  Environment: FTK_RANK,
// FTK_WORLD_SIZE (RANK / WORLD_SIZE of a torchrun-style launcher are honoured too), FTK_DEVICE (default: LOCAL_RANK, else 0) and
// FTK_COMM_ID_FILE, a path all ranks can read: rank 0 writes the 128-byte RCCL unique id there, the others wait for it.
// Returns nullptr (and no error) for a single process; nullptr with *error filled when the communicator cannot be made.
ftk_comm *SharedComm(ftk_context *ctx, std::string *error);
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
