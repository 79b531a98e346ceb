// ftk_internal.h — definitions shared by the translation units behind the C ABI (ftk_api.cpp, ftk_comm.cpp).
// Not installed, not part of the boundary: include/ftk.h is.
#pragma once

#include <stdio.h>
#include <stdlib.h>

#include <chrono>
#include <mutex>
#include <string>

#include "ftk_device.h"

using ftk::DevImage;

// The FTK_* experiment switches of the library, read ONCE per context (ftk_context_create; ftk_context_refresh_env re-reads them for
// a test or a sweep that flips one): no getenv on any call path.  None of them changes a result — they pick launch shapes and kernels.
#define FTK_ENV_SWITCHES(X)                                                                                                        \
    X(klt_waves, "FTK_KLT_WAVES") X(klt_group, "FTK_KLT_GROUP") X(klt_pipelined, "FTK_KLT_PIPELINED") X(klt_fast_kernel, "FTK_KLT_FAST_KERNEL") \
    X(lssd_chunked, "FTK_LSSD_CHUNKED") X(klt_spill, "FTK_KLT_SPILL") X(klt_spill_budget_mb, "FTK_KLT_SPILL_BUDGET_MB") X(klt_sched, "FTK_KLT_SCHED") \
    X(klt_swap, "FTK_KLT_SWAP") X(klt_order, "FTK_KLT_ORDER") X(klt_position_order, "FTK_KLT_POSITION_ORDER") X(klt_swap_dump, "FTK_KLT_SWAP_DUMP") \
    X(klt_sched_dump, "FTK_KLT_SCHED_DUMP") X(stamps_dump, "FTK_STAMPS_DUMP") X(klt_zerocopy, "FTK_KLT_ZEROCOPY") X(pyramid_zerocopy, "FTK_PYRAMID_ZEROCOPY") \
    X(match_wgs, "FTK_MATCH_WGS") X(match_splits, "FTK_MATCH_SPLITS") X(match_any_per, "FTK_MATCH_ANY_PER") X(match_kernel, "FTK_MATCH_KERNEL") \
    X(match_boxes, "FTK_MATCH_BOXES") X(match_stamps_dump, "FTK_MATCH_STAMPS_DUMP") X(match_small, "FTK_MATCH_SMALL") \
    X(direct_spread, "FTK_DIRECT_SPREAD") X(direct_spread_min_terms, "FTK_DIRECT_SPREAD_MIN_TERMS") X(direct_spread_resident, "FTK_DIRECT_SPREAD_RESIDENT") X(direct_spread_poison, "FTK_DIRECT_SPREAD_POISON") X(direct_spread_max_problems, "FTK_DIRECT_SPREAD_MAX_PROBLEMS") \
    X(cosine_kernel, "FTK_COSINE_KERNEL") X(cosine_chunked, "FTK_COSINE_CHUNKED") X(cosine_splits, "FTK_COSINE_SPLITS") X(cosine_two_pass, "FTK_COSINE_TWO_PASS") \
    X(cosine_small, "FTK_COSINE_SMALL") X(cosine_small_any, "FTK_COSINE_SMALL_ANY") X(reduction, "FTK_REDUCTION") X(klt_policy, "FTK_KLT_POLICY") X(klt_quad, "FTK_KLT_QUAD") X(klt_tail, "FTK_KLT_TAIL") X(klt_sched_min, "FTK_KLT_SCHED_MIN") X(klt_tail_class, "FTK_KLT_TAIL_CLASS") X(pinned_noncoherent, "FTK_PINNED_NONCOHERENT")

struct ftk_env {
#define X(field, name) const char *field = nullptr;
    FTK_ENV_SWITCHES(X)
#undef X
    enum { kCount = 0
#define X(field, name) +1
    FTK_ENV_SWITCHES(X)
#undef X
    };
    std::string keep[kCount];  // the values' storage: the environment may change under a pointer getenv returned
    void read() {
        int k = 0;
#define X(field, name)                     \
    if (const char *v = getenv(name)) {    \
        keep[k] = v;                       \
        field = keep[k].c_str();           \
    } else {                               \
        field = nullptr;                   \
    }                                      \
    ++k;
        FTK_ENV_SWITCHES(X)
#undef X
    }
    static bool off(const char *v) { return v && atoi(v) == 0; }  // "FTK_X=0" switches a default-on feature off
    static bool on(const char *v) { return v && atoi(v) != 0; }
};

#define FTK_ENV(ctx, field) ((ctx) ? (ctx)->env.field : nullptr)

struct ftk_context {
    // Every entry point that takes a context holds this lock for its whole duration: the scratch / pinned / workspace
    // buffers below are reused (and regrown) by every call, so calls on ONE context from several threads — e.g. the
    // left and right tracker objects of a stereo front end, which share the process-wide context of the C++ classes —
    // are serialised here instead of racing on them.  Recursive: the host-buffer wrappers call the *_device entries.
    std::recursive_mutex lock;
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string error;
    ftk_env env;  // the experiment switches as they were when the context was made (or last refreshed)
    // cached device scratch for the host-buffer entry points
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long *match_keys = nullptr;
    size_t match_keys_count = 0;
    float *match_boxes = nullptr;  // NearbyMatch bounding boxes (4 floats each)
    size_t match_boxes_count = 0;
    // workspace of the float-descriptor matcher (fp16 copies, norms, candidate lists)
    void *cosine_ws = nullptr;
    size_t cosine_ws_bytes = 0;
    // problem table of the direct-method launches
    void *direct_table = nullptr;
    size_t direct_table_bytes = 0;
    // per-feature projection tables of direct-method problems too large for LDS (16 B per tracked feature)
    void *direct_feat = nullptr;
    size_t direct_feat_bytes = 0;
    // zero-padded copies of descriptors whose width is not a power of two (device-resident matcher entry)
    // launch order of the trackers (ftk_klt_track_device): iteration counts of the last two calls and the permutations the
    // sort block of the tracker launches makes from them, double-buffered
    uint32_t *sched_iters[2] = {nullptr, nullptr};
    int32_t *sched_order[2] = {nullptr, nullptr};
    // position-keyed slot swaps (klt_common.h sched_resolve_slot): iteration counts by position (two hash tables), one claim word per launch slot
    uint32_t *sched_grid = nullptr;
    uint32_t *sched_claim = nullptr;
    uint8_t *sched_pred = nullptr;    // predicted iteration count of every feature of the call in hand (position-keyed launch order)
    uint32_t sched_recorded = 0;      // sched_call of the last call that left its counts in the position table (0: none yet)
    uint32_t sched_call = 0;     // calls that used the grid so far (tags its entries and the claims)
    size_t sched_capacity = 0;   // features each buffer holds
    int32_t sched_n = 0;         // feature count of the calls counted in sched_calls
    uint32_t sched_calls = 0;    // consecutive calls with that feature count so far
    // Tail-aware wave policy (round 5): the trackers' kernels report the iteration count of a call's LONGEST feature (features below
    // kTailReportFrom stay silent) into a device word per variant (atomicMax) whose raisers forward it to `tail_host`, device-visible host words
    // a later call's policy reads without any synchronisation: {call number << 8 | iterations}; feature 0 always reports, so every launch refreshes its word.  A heuristic input, never a result.
    uint32_t *tail_host = nullptr;
    uint32_t *tail_dev = nullptr;
    uint32_t tail_call = 0;                 // tracker launches of this context so far (tags the reports)
    struct TailState {
        uint32_t launches = 0;              // launches of this variant so far
        uint32_t long_until = 0;            // "this variant's calls have a long tail" while launches < long_until
        uint32_t longest = 0;               // the last long report's iteration count
    } tail[3][3];                           // [model][inverse, direct, fast]
    void *match_pad = nullptr;
    size_t match_pad_bytes = 0;
    // per-workgroup slices of the trackers' large-patch form (ftk_device.h KltParams::spill)
    void *klt_spill = nullptr;
    size_t klt_spill_bytes = 0;
    // hand-off workspace of the spread direct-method kernel (header, chunk flags, products)
    void *direct_spread = nullptr;
    size_t direct_spread_bytes = 0;
    int direct_spread_resident = -1;            // workgroups of the spread kernel this device holds at once (-1: not asked yet)
    uint32_t direct_spread_resident_features = 0;
    int direct_spread_launched = 0;             // problems the LAST ftk_direct_track_batch_device call spread over the chip (0: one workgroup each)
    size_t direct_spread_stride = 0;            // bytes of workspace per problem of that launch (header word 1 != 0: its waits ran out)
    bool direct_spread_off = false;             // set around the re-run of a poisoned spread launch
    uint32_t direct_spread_reruns = 0;          // such re-runs so far (tests)
    // pinned host staging for the host-buffer entry points (one H2D + one D2H per call)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    // BRIEF sampling pattern resident on the device, cached per (n_bits, half)
    int8_t *brief_pattern = nullptr;
    int32_t brief_bits = 0, brief_half = 0;
    // FTK_REDUCTION_EXACT (default) or FTK_REDUCTION_TREE: how the trackers' normal-equation sums are formed (ftk_set_reduction_mode)
    int32_t reduction = 0;
    // Two pinned, device-visible slots through which host images reach the pyramid launches (ftk_pyramid_build / _update of a
    // pageable image: a CPU copy into the slot, then the launch reads the slot over PCIe — no staged hipMemcpy, no stream
    // synchronisation; a slot is reused only after the event recorded behind its last reader has passed)
    struct ImageStage {
        uint8_t *host = nullptr;
        const uint8_t *device_view = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    } image_stage[2];
    int image_stage_next = 0;
};

struct ftk_pyramid {
    int device = 0;
    int32_t n_levels = 0;
    DevImage levels[FTK_MAX_LEVELS];
    uint8_t *owned = nullptr;  // single allocation holding every owned level
};


// Records the message on the context (or, with ctx == nullptr, for ftk_last_error(NULL)) and returns `code`.
int ftk_fail(ftk_context *ctx, int code, const char *fmt, ...);
#define fail ftk_fail

#define FTK_HIP(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            return ftk_fail(ctx, e_ == hipErrorOutOfMemory ? FTK_E_OUT_OF_MEMORY : FTK_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
        }                                                                                                    \
    } while (0)

#define FTK_LOCK(ctx) std::lock_guard<std::recursive_mutex> ftk_lock_guard_((ctx)->lock)

inline size_t ftk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// FTK_TRACE=1 in the environment: every host-buffer entry point of the C ABI prints its wall time to stderr when it returns
// ("[ftk trace] ftk_klt_track 83.1 us") — for finding out where a caller's timed region goes; costs one getenv per process.
struct ftk_trace_scope {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    bool on;
    explicit ftk_trace_scope(const char *n) : name(n), on(enabled()) {
        if (on) {
            t0 = std::chrono::steady_clock::now();
        }
    }
    ~ftk_trace_scope() {
        if (on) {
            fprintf(stderr, "[ftk trace] %s %.1f us\n", name, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
        }
    }
    static bool enabled() {
        static const bool e = getenv("FTK_TRACE") != nullptr && atoi(getenv("FTK_TRACE")) != 0;
        return e;
    }
};
#define FTK_TRACE_SCOPE(name) ftk_trace_scope ftk_trace_scope_(name)
// Grows a context-owned device buffer (stream-synchronising first: earlier launches may still read the old one).
int ftk_ensure_device_buffer(ftk_context *ctx, void **buf, size_t *have, size_t bytes);
