#include "device_runtime.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
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
bool g_comm_tried = false;

int EnvInt(const char *primary, const char *secondary, int fallback) {
    for (const char *name : {primary, secondary}) {
        if (name != nullptr) {
            if (const char *v = std::getenv(name)) {
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
        }
    }
    if (g_ctx == nullptr && error != nullptr) {
        *error = g_error;
    }
    return g_ctx;
}

ftk_comm *SharedComm(ftk_context *ctx, std::string *error) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_comm != nullptr || g_comm_tried) {
        if (g_comm == nullptr && !g_error.empty() && error != nullptr) {
            *error = g_error;
        }
        return g_comm;
    }
    g_comm_tried = true;
    const int world = EnvInt("FTK_WORLD_SIZE", "WORLD_SIZE", 1), rank = EnvInt("FTK_RANK", "RANK", 0);
    const char *id_file = std::getenv("FTK_COMM_ID_FILE");
    if (world <= 1 && id_file == nullptr) {
        return nullptr;  // a single process: the plain calls
    }
    unsigned char id[FTK_UNIQUE_ID_BYTES];
    if (id_file == nullptr) {
        g_error = "FTK_WORLD_SIZE > 1 needs FTK_COMM_ID_FILE (a path every rank can read) to hand out the RCCL unique id";
    } else if (rank == 0) {
        // written under a temporary name and renamed, so that a reader never sees a partial id
        const std::string tmp = std::string(id_file) + ".tmp";
        FILE *f = nullptr;
        if (ftk_comm_unique_id(id) != FTK_OK) {
            g_error = ftk_last_error(nullptr);
        } else if ((f = std::fopen(tmp.c_str(), "wb")) == nullptr || std::fwrite(id, 1, sizeof(id), f) != sizeof(id) || std::fclose(f) != 0 ||
                   std::rename(tmp.c_str(), id_file) != 0) {
            g_error = std::string("cannot write the RCCL unique id to ") + id_file;
        }
    } else {
        bool got = false;
        for (int attempt = 0; attempt < 1200 && !got; ++attempt) {  // up to two minutes for rank 0 to come up
            if (FILE *f = std::fopen(id_file, "rb")) {
                got = std::fread(id, 1, sizeof(id), f) == sizeof(id);
                std::fclose(f);
            }
            if (!got) {
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
        }
        if (!got) {
            g_error = std::string("timed out waiting for the RCCL unique id in ") + id_file;
        }
    }
    if (g_error.empty() && ftk_comm_create(ctx, rank, world, id, &g_comm) != FTK_OK) {
        g_comm = nullptr;
        g_error = ftk_last_error(ctx);
    }
    if (g_comm == nullptr && error != nullptr) {
        *error = g_error;
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
