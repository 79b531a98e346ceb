#include "device_runtime.h"

#include <cstdlib>
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

std::string LastError() {
    std::lock_guard<std::mutex> lock(g_mutex);
    return g_ctx != nullptr ? std::string(ftk_last_error(g_ctx)) : g_error;
}

}  // namespace device
}  // namespace feature_tracker
