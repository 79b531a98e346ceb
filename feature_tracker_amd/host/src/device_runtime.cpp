#include "device_runtime.h"

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace feature_tracker {
namespace device {

namespace {
std::mutex g_mutex;
ftk_context *g_ctx = nullptr;
bool g_tried = false;  // a failed creation (no device) is not retried on every call
std::string g_error;

ftk_comm *g_comm = nullptr;
bool g_comm_tried = false;  // one rendezvous per process: a failed one is reported on every call, not repeated (it can block for minutes)
std::string g_comm_error;

int EnvInt(const char *primary, const char *secondary, int fallback) {
    for (const char *name : {primary, secondary}) {
        if (name != nullptr) {
            const char *v = std::getenv(name);
            if (v != nullptr && v[0] != '\0') {
                return std::atoi(v);
            }
        }
    }
    return fallback;
}

struct ContextReaper {
    ~ContextReaper() {
        if (g_comm != nullptr) {
            ftk_comm_destroy(g_comm);
            g_comm = nullptr;
        }
        if (g_ctx != nullptr) {
            ftk_context_destroy(g_ctx);
            g_ctx = nullptr;
        }
    }
} g_reaper;
}  // namespace

ftk_context *SharedContext(std::string *error) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_ctx == nullptr && !g_tried) {
        g_tried = true;
        const int dev = EnvInt("FTK_DEVICE", "LOCAL_RANK", 0);
        // Bringing up the HIP runtime must not disturb the caller's std::rand() / random() stream (the runtime's own
        // initialisation draws from it): CreateImagePyramid now reaches this point BEFORE a caller like
        // test/test_direct_method.cpp:45-49 picks its features with std::rand(), and the reference's own pyramid
        // code consumes no random numbers.  glibc keeps rand() and random() in one state: park it, restore it.
        char scratch_state[256];
        char *callers_state = initstate(1u, scratch_state, sizeof(scratch_state));
        const int rc = ftk_context_create(dev, nullptr, &g_ctx);
        if (callers_state != nullptr) {
            setstate(callers_state);
        }
        if (rc != FTK_OK) {
            g_ctx = nullptr;
            g_error = ftk_last_error(nullptr);
        } else {
            // First-use cost, once per process and HERE: loading the kernels' code objects and the first staging allocations take
            // milliseconds, and the reference's programs start their timers at different points relative to the construction of
            // their objects (test_optical_flow.cpp:64 vs :69 constructs first; test_descriptor_matcher_brief.cpp:69 vs :79 starts
            // the timer first) — but all of them touch the device (Harris corners, CreateImagePyramid) before any timer runs.
            static const bool no_warmup = std::getenv("FTK_NO_WARMUP") != nullptr && std::atoi(std::getenv("FTK_NO_WARMUP")) != 0;  // experiment switch
            if (!no_warmup) {
                callers_state = initstate(1u, scratch_state, sizeof(scratch_state));
                (void)ftk_warmup(g_ctx, FTK_WARM_ALL);
                if (callers_state != nullptr) {
                    setstate(callers_state);
                }
            }
        }
    }
    if (g_ctx == nullptr && error != nullptr) {
        *error = g_error;
    }
    return g_ctx;
}

namespace {
// The id file: magic, a per-launch nonce, the 128-byte RCCL unique id.  The nonce is what tells this launch's file from one a
// previous (or crashed) run left under the same path: readers wait until a file with THEIR nonce appears.
constexpr char kIdMagic[8] = {'F', 'T', 'K', 'I', 'D', '0', '0', '2'};
constexpr size_t kNonceBytes = 64;
constexpr int kMaxRankSkewSeconds = 120;  // how far apart the ranks of one launch may start (AwaitCommId waits this long for rank 0)
struct IdFile {
    char magic[8];
    char nonce[kNonceBytes];
    unsigned char id[FTK_UNIQUE_ID_BYTES];
};

// The nonce as the id file stores it: as is while it fits, otherwise its head + '#' + a 64-bit FNV-1a hash of the WHOLE string — so
// that what follows the 63rd character (";parent=...;restart=..." behind a UUID run id) still tells two launches apart.
void PadNonce(const std::string &nonce, char out[kNonceBytes]) {
    std::memset(out, 0, kNonceBytes);
    if (nonce.size() < kNonceBytes) {
        std::memcpy(out, nonce.data(), nonce.size());
        return;
    }
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char ch : nonce) {
        h = (h ^ ch) * 1099511628211ull;
    }
    char tail[20];
    std::snprintf(tail, sizeof(tail), "#%016llx", h);
    std::memcpy(out, nonce.data(), kNonceBytes - 1 - 17);
    std::memcpy(out + kNonceBytes - 1 - 17, tail, 17);
}

std::string ShownNonce(const char stored[kNonceBytes]) { return std::string(stored, strnlen(stored, kNonceBytes)); }
}  // namespace

namespace {
// /proc/<pid>/stat: the parent's pid (field 4) and the start time in clock ticks since boot (field 22); false when unreadable.
bool ProcStat(long pid, long *ppid, std::string *start) {
    const std::string stat_path = "/proc/" + std::to_string(pid) + "/stat";
    std::string line;
    if (FILE *f = std::fopen(stat_path.c_str(), "rb")) {
        char buf[1024];
        const size_t got = std::fread(buf, 1, sizeof(buf) - 1, f);
        std::fclose(f);
        line.assign(buf, got);
    }
    // the command name (field 2) is in parentheses and may contain spaces: count fields after the LAST ')'
    const size_t close = line.rfind(')');
    if (close == std::string::npos) {
        return false;
    }
    size_t pos = close + 1;
    std::string field;
    for (int index = 3; index <= 22; ++index) {
        while (pos < line.size() && line[pos] == ' ') {
            ++pos;
        }
        const size_t end = line.find(' ', pos);
        field = line.substr(pos, end == std::string::npos ? std::string::npos : end - pos);
        if (index == 4) {
            *ppid = std::atol(field.c_str());
        }
        if (end == std::string::npos && index < 22) {
            return false;
        }
        pos = end == std::string::npos ? line.size() : end;
    }
    if (field.empty()) {
        return false;
    }
    *start = field;
    return true;
}

// Whether the environment a process was STARTED with holds `name` (/proc/<pid>/environ; unreadable: false).
bool StartedWith(long pid, const char *name) {
    const std::string path = "/proc/" + std::to_string(pid) + "/environ";
    std::string env;
    if (FILE *f = std::fopen(path.c_str(), "rb")) {
        char buf[4096];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0 && env.size() < (1u << 20)) {
            env.append(buf, got);
        }
        std::fclose(f);
    }
    const std::string key = std::string(name) + "=";
    for (size_t pos = 0; pos < env.size();) {
        if (env.compare(pos, key.size(), key) == 0) {
            return true;
        }
        const size_t end = env.find('\0', pos);
        if (end == std::string::npos) {
            break;
        }
        pos = end + 1;
    }
    return false;
}

// "<pid>@<start time>" of the process that launched the ranks: the nearest ancestor that was NOT itself started as a rank (no
// LOCAL_RANK in the environment it was started with).  Under plain torchrun that is the parent (the elastic agent); with a per-rank
// wrapper in between (`torchrun --no-python wrapper.sh`, a per-rank `rocprofv3 -- python3 ...`: the wrapper inherits LOCAL_RANK from
// the agent) it is still the agent, the same for every rank — round 4 took the parent and gave every wrapped rank its own nonce
// (ADVICE r4).  The start time makes the pair unique for the life of the machine even when pids are reused.  "" when unreadable.
std::string LauncherIdentity() {
    long pid = static_cast<long>(getppid());
    for (int depth = 0; depth < 8 && pid > 1; ++depth) {
        long ppid = 0;
        std::string start;
        if (!ProcStat(pid, &ppid, &start)) {
            return std::string();
        }
        if (!StartedWith(pid, "LOCAL_RANK")) {
            return std::to_string(pid) + "@" + start;
        }
        pid = ppid;
    }
    return std::string();
}

const std::chrono::system_clock::time_point g_process_start = std::chrono::system_clock::now();
}  // namespace

// What tells THIS launch's id file from one an earlier (or crashed) run left under the same path.  FTK_COMM_NONCE, when the caller
// sets it, is taken as is — REQUIRED when the ranks do not descend from one launcher process that exports LOCAL_RANK to them (two
// terminals, an MPI launcher whose per-rank shell sets LOCAL_RANK itself): their automatic nonces would differ, and the reader's
// time-out message then names both nonces.
// Otherwise: the launcher's variables are NOT unique per launch under default torchrun (TORCHELASTIC_RUN_ID is "none",
// MASTER_PORT 29500, every time), so under a launcher that forks all ranks of a node from one agent process (it exports
// LOCAL_RANK / TORCHELASTIC_RUN_ID; this library shards over the GPUs of ONE node) the agent's identity — pid and start time —
// and the restart count of an elastic agent are part of the nonce: the same for every rank of a launch, different for the next.
std::string CommLaunchNonce() {
    if (const char *v = std::getenv("FTK_COMM_NONCE")) {
        if (v[0] != '\0') {
            return std::string("FTK_COMM_NONCE=") + v;
        }
    }
    std::string nonce;
    for (const char *name : {"TORCHELASTIC_RUN_ID", "MASTER_PORT"}) {
        const char *v = std::getenv(name);
        if (v != nullptr && v[0] != '\0') {
            nonce = std::string(name) + "=" + v;
            break;
        }
    }
    const char *local_rank = std::getenv("LOCAL_RANK");
    const bool common_parent = (local_rank != nullptr && local_rank[0] != '\0') || std::getenv("TORCHELASTIC_RUN_ID") != nullptr;
    if (common_parent) {
        const std::string parent = LauncherIdentity();
        if (!parent.empty()) {
            nonce += (nonce.empty() ? "" : ";") + std::string("parent=") + parent;
        }
        if (const char *restarts = std::getenv("TORCHELASTIC_RESTART_COUNT")) {
            nonce += std::string(";restart=") + restarts;
        }
    }
    return nonce;  // (longer than the id file's field: stored as head + hash of the whole string, PadNonce)
}

bool PublishCommId(const std::string &path, const std::string &nonce, const unsigned char id[FTK_UNIQUE_ID_BYTES], std::string *error) {
    // A file of an earlier run must never be read as this run's: it is removed BEFORE the new one exists (callers remove it
    // before they even generate the id); the new one is written under a temporary name and renamed in, so that a reader never
    // sees a partial file.
    std::remove(path.c_str());
    IdFile file;
    std::memcpy(file.magic, kIdMagic, sizeof(kIdMagic));
    PadNonce(nonce, file.nonce);
    std::memcpy(file.id, id, FTK_UNIQUE_ID_BYTES);
    const std::string tmp = path + ".tmp." + std::to_string(static_cast<long>(getpid()));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    const bool ok = f != nullptr && std::fwrite(&file, 1, sizeof(file), f) == sizeof(file);
    if ((f != nullptr && std::fclose(f) != 0) || !ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
        std::remove(tmp.c_str());
        if (error != nullptr) {
            *error = "cannot write the RCCL unique id to " + path;
        }
        return false;
    }
    return true;
}

bool AwaitCommId(const std::string &path, const std::string &nonce, int timeout_ms, unsigned char id[FTK_UNIQUE_ID_BYTES], std::string *error) {
    char want[kNonceBytes];
    PadNonce(nonce, want);
    bool stale_seen = false;
    std::string found;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    for (;;) {
        if (FILE *f = std::fopen(path.c_str(), "rb")) {
            IdFile file;
            const bool whole = std::fread(&file, 1, sizeof(file), f) == sizeof(file) && std::memcmp(file.magic, kIdMagic, sizeof(kIdMagic)) == 0;
            // ... and whatever the nonce says, a file written long before this process started is not this launch's: the ranks of a
            // launch come up within the time a reader waits for rank 0 (kMaxRankSkewSeconds), so an older file is a leftover
            struct stat st;
            const bool fresh = fstat(fileno(f), &st) == 0 &&
                               std::chrono::system_clock::from_time_t(st.st_mtime) + std::chrono::seconds(kMaxRankSkewSeconds) >= g_process_start;
            std::fclose(f);
            if (whole && fresh && std::memcmp(file.nonce, want, kNonceBytes) == 0) {
                std::memcpy(id, file.id, FTK_UNIQUE_ID_BYTES);
                return true;
            }
            stale_seen = true;  // another launch's file, an old format or a foreign file: rank 0 of THIS launch replaces it
            found = whole ? ShownNonce(file.nonce) + (fresh ? "" : " (written before this launch)") : std::string("<not an id file>");
        }
        if (std::chrono::steady_clock::now() >= deadline) {
            break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    if (error != nullptr) {
        *error = "timed out waiting for the RCCL unique id in " + path +
                 (stale_seen ? " (the file there belongs to another launch: this rank expects the nonce \"" + ShownNonce(want) + "\", the file holds \"" + found +
                                   "\"; ranks that do not share one launcher process must be given the same FTK_COMM_NONCE)"
                             : "");
    }
    return false;
}

void WarmUp(unsigned what) {
    (void)what;  // the shared context prepares every family when it is created (SharedContext)
    std::string error;
    (void)SharedContext(&error);
}

int CommOptIn(int *rank, int *world, std::string *error) {
    // Opt-in is EXPLICIT: the sharded mode is only valid when every rank passes identical pyramids and feature lists, which the
    // generic launcher variables cannot establish (a data-parallel job under torchrun tracks DIFFERENT frames per rank).  So only
    // FTK_* variables switch it on — FTK_COMM_ID_FILE, or FTK_WORLD_SIZE > 1, which then demands the file — and RANK / WORLD_SIZE
    // are read as defaults only once that opt-in is present.  LOCAL_RANK stays the device default (SharedContext).
    const char *id_file = std::getenv("FTK_COMM_ID_FILE");
    if (id_file != nullptr && id_file[0] == '\0') {
        id_file = nullptr;
    }
    const int ftk_world = EnvInt("FTK_WORLD_SIZE", nullptr, 0);
    if (id_file == nullptr && ftk_world <= 1) {
        return 0;
    }
    *world = ftk_world > 0 ? ftk_world : EnvInt("WORLD_SIZE", nullptr, 1);
    *rank = EnvInt("FTK_RANK", "RANK", 0);
    if (id_file == nullptr) {
        *error = "FTK_WORLD_SIZE > 1 needs FTK_COMM_ID_FILE (a path every rank can read, unique per launch) to hand out the RCCL unique id";
        return -1;
    }
    if (*world < 1 || *rank < 0 || *rank >= *world) {
        *error = "FTK_RANK / FTK_WORLD_SIZE: rank " + std::to_string(*rank) + " is not in [0, " + std::to_string(*world) + ")";
        return -1;
    }
    return 1;
}

ftk_comm *SharedComm(ftk_context *ctx, std::string *error) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_comm != nullptr || g_comm_tried) {
        if (g_comm == nullptr && !g_comm_error.empty() && error != nullptr) {
            *error = g_comm_error;
        }
        return g_comm;
    }
    g_comm_tried = true;
    int rank = 0, world = 1;
    const int opt_in = CommOptIn(&rank, &world, &g_comm_error);
    if (opt_in == 0) {
        return nullptr;  // a single process, or the ranks of somebody else's job: the plain calls
    }
    const char *id_file = std::getenv("FTK_COMM_ID_FILE");
    const std::string nonce = CommLaunchNonce();
    unsigned char id[FTK_UNIQUE_ID_BYTES];
    if (opt_in < 0) {
        id_file = nullptr;  // g_comm_error says why
    } else if (rank == 0) {
        std::remove(id_file);  // before the new id exists, so that the window in which a stale file can be read is as short as possible
        if (ftk_comm_unique_id(id) != FTK_OK) {
            g_comm_error = ftk_last_error(nullptr);
        } else {
            PublishCommId(id_file, nonce, id, &g_comm_error);
        }
    } else {
        AwaitCommId(id_file, nonce, kMaxRankSkewSeconds * 1000, id, &g_comm_error);  // up to two minutes for rank 0 to come up
    }
    if (g_comm_error.empty() && ftk_comm_create(ctx, rank, world, id, &g_comm) != FTK_OK) {
        g_comm = nullptr;
        g_comm_error = ftk_last_error(ctx);
    }
    // ncclCommInitRank is collective: once it has returned on rank 0 every rank has read the id, and the file has done its job.
    if (rank == 0 && id_file != nullptr) {
        std::remove(id_file);
    }
    if (g_comm == nullptr && error != nullptr) {
        *error = g_comm_error;
    }
    return g_comm;
}

namespace {
struct PyramidDeleter {
    void operator()(void *p) const { ftk_pyramid_destroy(static_cast<ftk_pyramid *>(p)); }
};
}  // namespace

namespace {
// Guards the twin bookkeeping of every pyramid: the reference lets several trackers (threads) read one const ImagePyramid.
std::recursive_mutex g_twin_mutex;

void AdoptTwin(const ImagePyramid &pyramid, ftk_pyramid *dev) {
    pyramid.device_twin() = std::shared_ptr<void>(dev, PyramidDeleter());
    pyramid.device_twin_generation() = pyramid.generation();
    pyramid.device_twin_stamp() = pyramid.ContentStamp();
}

bool TwinMatchesGeneration(const ImagePyramid &pyramid) {
    return pyramid.device_twin() && pyramid.device_twin_generation() == pyramid.generation();
}
}  // namespace

ftk_pyramid *PyramidTwin(ftk_context *ctx, const ImagePyramid &pyramid, std::string *error) {
    std::lock_guard<std::recursive_mutex> lock(g_twin_mutex);
    // Reused while the host object is unchanged AND level 0 (a caller-owned buffer) still holds the bytes the twin was made from.
    if (TwinMatchesGeneration(pyramid) && pyramid.device_twin_stamp() == pyramid.ContentStamp()) {
        return static_cast<ftk_pyramid *>(pyramid.device_twin().get());
    }
    ftk_image levels[FTK_MAX_LEVELS];
    const int32_t n = static_cast<int32_t>(pyramid.level());
    if (n < 1 || n > FTK_MAX_LEVELS) {
        *error = "image pyramid has no levels (CreateImagePyramid was not called)";
        return nullptr;
    }
    // What the reference would read now is whatever the host buffers hold: complete them (levels >= 1 may still live only
    // in the outdated twin) and upload them as they are.
    pyramid.EnsureHostLevels();
    for (int32_t i = 0; i < n; ++i) {
        pyramid.LevelGeometry(static_cast<uint32_t>(i), &levels[i].data, &levels[i].rows, &levels[i].cols);
    }
    ftk_pyramid *dev = nullptr;
    if (ftk_pyramid_upload(ctx, levels, n, &dev) != FTK_OK) {
        *error = ftk_last_error(ctx);
        return nullptr;
    }
    AdoptTwin(pyramid, dev);
    return dev;
}

// ImagePyramid::CreateImagePyramid (datatype_image_pyramid.h): level 0 goes up once, levels >= 1 are built in HBM.
bool BuildPyramidOnDevice(const ImagePyramid &pyramid) {
    static const bool host_pyramid = std::getenv("FTK_HOST_PYRAMID") != nullptr && std::atoi(std::getenv("FTK_HOST_PYRAMID")) != 0;
    if (host_pyramid) {
        return false;  // experiment switch: levels >= 1 by the host loop, all levels uploaded at the first TrackFeatures
    }
    std::string error;
    ftk_context *ctx = SharedContext(&error);
    if (ctx == nullptr) {
        return false;  // no device: the pyramid keeps its host loop; the trackers will report the missing device themselves
    }
    std::lock_guard<std::recursive_mutex> lock(g_twin_mutex);
    const uint8_t *data = nullptr;
    int32_t rows = 0, cols = 0;
    pyramid.LevelGeometry(0, &data, &rows, &cols);
    const int32_t n = static_cast<int32_t>(pyramid.level());
    if (data == nullptr || n < 1 || n > FTK_MAX_LEVELS) {
        return false;
    }
    ftk_pyramid *dev = nullptr;
    if (ftk_pyramid_build(ctx, data, rows, cols, n, 0, &dev) != FTK_OK) {
        return false;
    }
    AdoptTwin(pyramid, dev);
    return true;
}

bool DownloadPyramidLevels(const ImagePyramid &pyramid) {
    std::lock_guard<std::recursive_mutex> lock(g_twin_mutex);
    if (!TwinMatchesGeneration(pyramid)) {
        return false;
    }
    std::string error;
    ftk_context *ctx = SharedContext(&error);
    if (ctx == nullptr) {
        return false;
    }
    const ftk_pyramid *dev = static_cast<const ftk_pyramid *>(pyramid.device_twin().get());
    for (uint32_t i = 1; i < pyramid.level(); ++i) {
        const uint8_t *data = nullptr;
        int32_t rows = 0, cols = 0;
        pyramid.LevelGeometry(i, &data, &rows, &cols);
        if (ftk_pyramid_download_level(ctx, dev, static_cast<int32_t>(i), const_cast<uint8_t *>(data)) != FTK_OK) {
            return false;
        }
    }
    return true;
}

std::string LastError() {
    std::lock_guard<std::mutex> lock(g_mutex);
    return g_ctx != nullptr ? std::string(ftk_last_error(g_ctx)) : g_error;
}

}  // namespace device
}  // namespace feature_tracker
