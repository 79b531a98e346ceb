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
// First-use cost out of the callers' timed regions: creating the shared context (the first device use of the process — in the
// reference's programs Harris detection or CreateImagePyramid, before any timer) also loads every kernel family's code object and
// makes the first staging allocations (ftk_warmup(FTK_WARM_ALL), a few milliseconds once per process; FTK_NO_WARMUP=1 skips it).
// The constructors of the classes of this layer call WarmUp so that a program whose first device use IS a tracker / matcher
// object still pays at construction (test/test_optical_flow.cpp:64 constructs before its timer starts at :69).
// Never fails: without a device it does nothing and the first real call reports the missing device.
void WarmUp(unsigned what);
// One process per GPU, EXPLICIT opt-in: when FTK_COMM_ID_FILE is set (world size 1 included: same path through RCCL) — or
// FTK_WORLD_SIZE > 1, which then requires it — the trackers shard their feature list over the ranks and every process receives
// the complete result (ftk_klt_track_sharded: RCCL all-gather issued by libftk_hip.so).  That is only valid when EVERY rank passes
// the same pyramids and the same feature list, which a launcher's generic WORLD_SIZE / RANK cannot tell (the ranks of a
// data-parallel job track different frames): those two are therefore read only as DEFAULTS for FTK_WORLD_SIZE / FTK_RANK once
// the opt-in is present.  FTK_DEVICE (default: LOCAL_RANK, else 0) picks the GPU.
// FTK_COMM_ID_FILE is a path all ranks can read and that is UNIQUE PER LAUNCH: rank 0 removes whatever is there, writes
// {magic, launch nonce, the 128-byte RCCL unique id} under a temporary name, renames it in, and removes it again once the
// communicator exists; the others wait (up to two minutes) for a file carrying THEIR nonce, so a file left by a crashed or
// earlier run under the same path is not mistaken for this run's.  The nonce is $FTK_COMM_NONCE, else $TORCHELASTIC_RUN_ID, else
// $MASTER_PORT, else empty (then only the path's uniqueness protects the launch).
// One rendezvous per process: a failure is reported by every later call, not retried (it can block for minutes).
// Returns nullptr (and no error) for a single process; nullptr with *error filled when the communicator cannot be made.
ftk_comm *SharedComm(ftk_context *ctx, std::string *error);
// The rendezvous' two halves and its nonce (no device needed; tests/test_host_logic_cpu.py drives them through comm_id_cli).
// 0: the environment does not opt in (plain single-process calls); 1: it does, *rank / *world filled; -1: it does but is inconsistent (*error).
int CommOptIn(int *rank, int *world, std::string *error);
std::string CommLaunchNonce();
bool PublishCommId(const std::string &path, const std::string &nonce, const unsigned char id[FTK_UNIQUE_ID_BYTES], std::string *error);
bool AwaitCommId(const std::string &path, const std::string &nonce, int timeout_ms, unsigned char id[FTK_UNIQUE_ID_BYTES], std::string *error);
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
