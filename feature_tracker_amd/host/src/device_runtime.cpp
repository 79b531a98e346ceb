#include "device_runtime.h"

#include <cstdlib>
#include <memory>
#include <mutex>

namespace feature_tracker {
namespace device {

namespace {
std::mutex g_mutex;
ftk_context *g_ctx = nullptr;
bool g_tried = false;  // a failed creation (no device) is not retried on every call
std::string g_error;

struct ContextReaper {
    ~ContextReaper() {
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
        int dev = 0;
        if (const char *env = std::getenv("FTK_DEVICE")) {
            dev = std::atoi(env);
        }
        const int rc = ftk_context_create(dev, nullptr, &g_ctx);
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
