#include "device_runtime.h"

#include <cstdlib>
#include <memory>
#include <mutex>

namespace feature_tracker {
namespace device {

namespace {
std::mutex g_mutex;
ftk_context *g_ctx = nullptr;
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
    if (g_ctx == nullptr) {
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

ftk_pyramid *PyramidTwin(ftk_context *ctx, const ImagePyramid &pyramid, std::string *error) {
    std::shared_ptr<void> &twin = pyramid.device_twin();
    if (twin && pyramid.device_twin_generation() == pyramid.generation()) {
        return static_cast<ftk_pyramid *>(twin.get());
    }
    ftk_image levels[FTK_MAX_LEVELS];
    const int32_t n = static_cast<int32_t>(pyramid.level());
    if (n < 1 || n > FTK_MAX_LEVELS) {
        *error = "image pyramid has no levels (CreateImagePyramid was not called)";
        return nullptr;
    }
    for (int32_t i = 0; i < n; ++i) {
        const GrayImage &im = pyramid.GetImageConst(i);
        levels[i].data = im.data();
        levels[i].rows = im.rows();
        levels[i].cols = im.cols();
    }
    ftk_pyramid *dev = nullptr;
    if (ftk_pyramid_upload(ctx, levels, n, &dev) != FTK_OK) {
        *error = ftk_last_error(ctx);
        return nullptr;
    }
    twin = std::shared_ptr<void>(dev, PyramidDeleter());
    pyramid.device_twin_generation() = pyramid.generation();
    return dev;
}

std::string LastError() {
    std::lock_guard<std::mutex> lock(g_mutex);
    return g_ctx != nullptr ? std::string(ftk_last_error(g_ctx)) : g_error;
}

}  // namespace device
}  // namespace feature_tracker
