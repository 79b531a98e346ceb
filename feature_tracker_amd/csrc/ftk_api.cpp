// ftk_api.cpp — the C ABI of libftk_hip.so (declared in include/ftk.h).
//
// Host-side plumbing only: argument validation, device buffers, stream ordering, launches.
// All numerics live in the kernels.  There is deliberately no CPU fallback: if HIP is not
// usable every compute entry point fails with FTK_E_NO_DEVICE / FTK_E_HIP.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ftk_device.h"
#include "ftk_internal.h"

using ftk::DevImage;

thread_local std::string g_create_error;

int ftk_fail(ftk_context *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        ctx->error = buf;
    } else {
        g_create_error = buf;
    }
    return code;
}

namespace {

size_t align_up(size_t x, size_t a) { return ftk_align_up(x, a); }

int ensure_scratch(ftk_context *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) {
        return FTK_OK;
    }
    if (ctx->scratch) {
        FTK_HIP(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    const size_t want = align_up(bytes + bytes / 2, 4096);
    FTK_HIP(ctx, hipMalloc(&ctx->scratch, want));
    ctx->scratch_bytes = want;
    return FTK_OK;
}

int ensure_pinned(ftk_context *ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) {
        return FTK_OK;
    }
    if (ctx->pinned) {
        FTK_HIP(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    const size_t want = align_up(bytes + bytes / 2, 4096);
    // Coarse-grained (non-coherent) host memory: cacheable in the device's L2, coherent at kernel boundaries — which is all the
    // host-buffer entry points need (the host writes the block before the launch and reads it after the synchronisation).  The 2 000
    // workgroups of a zero-copy tracker call then share 64-byte lines instead of each crossing PCIe for its own 17 bytes, and their
    // results leave the chip as whole lines when the kernel ends: same box, 2 000-feature call 68.1 / 71.2 -> 66.2 / 65.6 us, small
    // calls unchanged (round 5).  FTK_PINNED_NONCOHERENT=0: fine-grained memory as before.
    const unsigned flags = ftk_env::off(FTK_ENV(ctx, pinned_noncoherent)) ? hipHostMallocDefault : hipHostMallocNonCoherent;
    FTK_HIP(ctx, hipHostMalloc(&ctx->pinned, want, flags));
    ctx->pinned_bytes = want;
    return FTK_OK;
}

// The next image-staging slot, free and at least `bytes` large (ftk_internal.h ImageStage); *out stays null when the device cannot
// address pinned host memory (the callers then take their copy paths).
int acquire_image_stage(ftk_context *ctx, size_t bytes, ftk_context::ImageStage **out) {
    *out = nullptr;
    ftk_context::ImageStage &st = ctx->image_stage[ctx->image_stage_next];
    if (st.busy) {
        FTK_HIP(ctx, hipEventSynchronize(st.done));  // normally long past: two frames per tracker call
        st.busy = false;
    }
    if (st.bytes < bytes) {
        if (st.host) {
            FTK_HIP(ctx, hipHostFree(st.host));
            st.host = nullptr;
            st.bytes = 0;
        }
        const size_t want = align_up(bytes, 1u << 20);
        void *h = nullptr, *d = nullptr;
        FTK_HIP(ctx, hipHostMalloc(&h, want, hipHostMallocDefault));
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || d == nullptr) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            return FTK_OK;  // *out == nullptr
        }
        st.host = static_cast<uint8_t *>(h);
        st.device_view = static_cast<const uint8_t *>(d);
        st.bytes = want;
    }
    if (!st.done) {
        FTK_HIP(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    }
    ctx->image_stage_next ^= 1;
    *out = &st;
    return FTK_OK;
}

int ensure_match_keys(ftk_context *ctx, size_t count) {
    if (count <= ctx->match_keys_count) {
        return FTK_OK;
    }
    if (ctx->match_keys) {
        FTK_HIP(ctx, hipFree(ctx->match_keys));
        ctx->match_keys = nullptr;
        ctx->match_keys_count = 0;
    }
    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->match_keys), sizeof(unsigned long long) * count));
    // all-ones = "no match yet"; the epilogue kernel restores this state after every call
    FTK_HIP(ctx, hipMemsetAsync(ctx->match_keys, 0xFF, sizeof(unsigned long long) * count, ctx->stream));
    ctx->match_keys_count = count;
    return FTK_OK;
}

constexpr uint32_t kSchedMinFeatures = 4096;  // below this (nearly) every feature is resident from the start: nothing to order ...
constexpr uint32_t kSchedMinLongTail = 1024;  // ... unless the calls have a long tail (see ftk_klt_track_device)
constexpr size_t kSchedTableWords = (2u << 16) + 2;  // two position tables of 2^16 entries (klt_common.h kSchedTableSize) + the two "no tail" flags behind them
constexpr size_t kSchedOrderWords = 512;             // behind them: histogram + cursors of the position-keyed launch order (klt_position_order_launch)
constexpr int32_t kSchedMaxFeatures = 1 << 18;  // the sort block walks the list alone; beyond this it could outlast the launch

int ensure_match_boxes(ftk_context *ctx, size_t count) {
    if (count <= ctx->match_boxes_count) {
        return FTK_OK;
    }
    if (ctx->match_boxes) {
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FTK_HIP(ctx, hipFree(ctx->match_boxes));
        ctx->match_boxes = nullptr;
        ctx->match_boxes_count = 0;
    }
    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->match_boxes), sizeof(float) * 4 * count));
    ctx->match_boxes_count = count;
    return FTK_OK;
}

int ensure_cosine_ws(ftk_context *ctx, size_t bytes) {
    if (bytes <= ctx->cosine_ws_bytes) {
        return FTK_OK;
    }
    if (ctx->cosine_ws) {
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FTK_HIP(ctx, hipFree(ctx->cosine_ws));
        ctx->cosine_ws = nullptr;
        ctx->cosine_ws_bytes = 0;
    }
    const size_t want = align_up(bytes + bytes / 4, 4096);
    FTK_HIP(ctx, hipMalloc(&ctx->cosine_ws, want));
    ctx->cosine_ws_bytes = want;
    return FTK_OK;
}

}  // namespace

// Grows a context-owned device buffer (stream-synchronising first: earlier launches may still read the old one).
int ftk_ensure_device_buffer(ftk_context *ctx, void **buf, size_t *have, size_t bytes) {
    if (bytes <= *have) {
        return FTK_OK;
    }
    if (*buf) {
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FTK_HIP(ctx, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    const size_t want = align_up(bytes + bytes / 4, 4096);
    FTK_HIP(ctx, hipMalloc(buf, want));
    *have = want;
    return FTK_OK;
}

namespace ftk {
namespace {
#include "klt_wave_policy.inc"

// nearest bucket centre on a log scale (the centres ascend)
int policy_bucket(const int *centres, int count, int x) {
    int best = 0;
    for (int i = 1; i < count; ++i) {
        // x is nearer to centres[i] than to centres[i - 1] when x * x > centres[i - 1] * centres[i]
        if ((long long)x * x > (long long)centres[i - 1] * centres[i]) {
            best = i;
        }
    }
    return best;
}
}  // namespace

// Waves per feature for one call (klt_wave_policy.inc: measured, generated): method_class 0 inverse, 1 direct, 2 fast-like.
// Patches beyond the table's largest bucket by more than a factor of two (from about 30 x 30) are outside what was swept: the
// lanes-per-pixel rule serves them (they run four waves: the pixel loops dominate).
int klt_policy_waves(int model, int method_class, int consider_luminance, int long_tail, int pixels, int n) {
    if (pixels > 2 * kPolicyPixels[kPolicyPixelBuckets - 1]) {
        return std::min(4, std::max(1, (pixels + 63) / 64));
    }
    int variant = model * 3 + method_class;
    if (model == FTK_MODEL_LSSD && method_class == 2 && consider_luminance) {
        variant = 9;
    }
    const int w = kWavePolicy[variant][long_tail ? 1 : 0][policy_bucket(kPolicyPixels, kPolicyPixelBuckets, pixels)][policy_bucket(kPolicyFeatures, kPolicyFeatureBuckets, n)];
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}
}  // namespace ftk

namespace {

// Direct method: a batch is spread over the chip (1 + NP workgroups per problem) while at least two producer workgroups per problem
// fit beside the others (NP = 32 for up to six problems, then what the 224 usable workgroups of a whole MI355X allow).  Round 5, same box,
// scripts/direct_batch_time.py (300 points x 13 x 13 x 4 levels), spread / one workgroup per problem, ms: 1 problem 1.01 / 1.81,
// 6: 1.05 / 1.85, 12: 1.11 / 1.85, 24 (NP 8): 1.16 / 1.86, 32 (6): 1.21 / 1.86, 44 (4): 1.30 / 1.87, 56 (3): 1.41 / 1.89, 64 (2): 1.39 /
// 1.89, 74 (2): 1.54 / 1.88, 100 (1): 2.03 / 1.93 — one producer workgroup does not keep up with its consumer's chain, two do
// (profiles/r5_direct_spread_consumer.txt).  Beyond that one workgroup per problem IS the fast form, and its time is one problem's.
constexpr int kDirectSpreadMaxProblems = 112;  // (two producers each no longer fit from 75 problems on a whole device: the fit decides)
constexpr int kDirectSpreadMinProducers = 2;
constexpr uint32_t kTailLongFrom = 24;  // iterations of a call's longest feature from which the call counts as tail-bound
constexpr uint32_t kTailHold = 8;       // launches of the variant for which one such report holds
constexpr uint32_t kTailFresh = 256;    // launches of the context a report may lag behind (the host enqueues far ahead of the device)

int ensure_device_buffer(ftk_context *ctx, void **buf, size_t *have, size_t bytes) { return ftk_ensure_device_buffer(ctx, buf, have, bytes); }

int fill_klt_params(ftk_context *ctx, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur, int32_t n,
                    const float *prior, int consider_luminance, int single_level, ftk::KltParams *out) {
    if (!opt || !ref || !cur) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt: null options or pyramid");
    }
    if (model < FTK_MODEL_BASIC || model > FTK_MODEL_LSSD) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt: unknown model %d", model);
    }
    if (opt->method < FTK_METHOD_INVERSE || opt->method > FTK_METHOD_NEON) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt: unknown method %d", opt->method);
    }
    if (ref->n_levels != cur->n_levels) {
        // OpticalFlow::TrackFeatures returns false here (optical_flow.cpp:9); callers above the ABI handle it
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt: pyramid level mismatch (%d vs %d)", ref->n_levels, cur->n_levels);
    }
    if (ref->n_levels < 1) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt: empty pyramid");
    }
    if (ref->device != ctx->device || cur->device != ctx->device) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt: pyramid lives on another device");
    }
    // The reference takes any int32 half size (optical_flow.h:24-25).  Here: up to 1023 (a 2047 x 2047 patch; pixel indices stay
    // below 2^23 for the 24-bit multiplier); patches beyond a workgroup's LDS run the large-patch form below.
    if (opt->half_rows < 0 || opt->half_cols < 0 || opt->half_rows > 1023 || opt->half_cols > 1023) {
        return fail(ctx, FTK_E_UNSUPPORTED, "klt: half patch size (%d, %d) outside [0, 1023]", opt->half_rows, opt->half_cols);
    }
    ftk::KltParams &p = *out;
    memset(&p, 0, sizeof(p));
    p.tree = (ctx && ctx->reduction == FTK_REDUCTION_TREE) ? 1 : 0;  // before the LDS size is computed: the throughput mode of the pipelined kernel keeps one table per wave
    p.n_levels = single_level ? 1 : ref->n_levels;
    p.single_level = single_level ? 1 : 0;
    for (int i = 0; i < p.n_levels; ++i) {
        p.ref[i] = ref->levels[i];
        p.cur[i] = cur->levels[i];
    }
    p.n = n;
    p.order = nullptr;        // list order unless the caller of this function installs a permutation
    p.sched_iters = nullptr;
    p.sort_iters = nullptr;
    p.sort_order_out = nullptr;
    p.n_track = ((uint32_t)n < opt->max_track_points) ? (uint32_t)n : opt->max_track_points;
    p.max_iteration = opt->max_iteration;
    p.max_large_step = opt->max_tolerance_large_step;
    p.half_rows = opt->half_rows;
    p.half_cols = opt->half_cols;
    p.converge = opt->max_converge_step;
    static const float identity[4] = {1.0f, 0.0f, 0.0f, 1.0f};
    const float *pr = prior ? prior : identity;
    for (int i = 0; i < 4; ++i) {
        p.prior[i] = pr[i];
    }
    p.consider_luminance = consider_luminance ? 1 : 0;
    ftk::klt_fill_geometry(p);  // patch / window / lattice geometry: everything that follows from the half sizes alone (ftk_device.h)
    // Wavefronts per feature: DATA, not rules (round 5).  csrc/klt_wave_policy.inc is generated by
    // scripts/make_wave_policy.py from one committed sweep (scripts/wave_policy_sweep.py -> profiles/r5_wave_policy_sweep.jsonl:
    // every variant x 9 x 9 ... 21 x 21 x 300 ... 16 000 features x one to four waves, on the synthetic scene and on the reference's
    // example pair) and holds, per (variant, tail class, pixel bucket, feature bucket), the wave count that measured fastest.  The
    // kernel forms follow from the count (below): one wave = the one-wave kernels (klt_fast_kernels.hip, the chunked LSSD levels, the
    // pipelined Basic kernel's solo form), more = the multi-wave forms.  tests/test_wave_policy_gpu.py times cells of every variant
    // and fails when the table's column is more than 10 % off the best.  Whatever is chosen changes the launch shape only, never a
    // result.
    //
    // Tail class: the table was swept on two kinds of scene — every feature done after a handful of iterations, and real frames in
    // which a few features never converge and run kMaxIteration iterations on every level, so that a call of a few thousand features
    // lasts as long as its slowest one.  The kernels report each call's longest feature (klt_common.h tail_report); a variant whose
    // recent calls had one of kTailLongFrom iterations or more is looked up in the long-tail half of the table.
    p.long_tail = 0;
    if (ctx && ctx->tail_host && !ftk_env::off(FTK_ENV(ctx, klt_tail)) && model >= 0 && model < 3) {
        const int mi = opt->method == FTK_METHOD_INVERSE ? 0 : (opt->method == FTK_METHOD_DIRECT ? 1 : 2);
        ftk_context::TailState &ts = ctx->tail[model][mi];
        // this variant's own word: {call number << 8 | iterations} of the longest feature of its most recent launch that has got that
        // far.  The host may be many launches ahead of the device (back-to-back calls), so a report counts while it is at most
        // kTailFresh launches of the context old, and one long report holds for kTailHold launches of the variant.
        const uint32_t seen = reinterpret_cast<volatile uint32_t *>(ctx->tail_host)[model * 3 + mi];
        const uint32_t age = (ctx->tail_call - (seen >> 8)) & 0xFFFFFFu;
        if (seen != 0 && age <= kTailFresh && (seen & 0xFFu) >= kTailLongFrom) {
            ts.long_until = ts.launches + kTailHold;
            ts.longest = seen & 0xFFu;
        }
        p.long_tail = ts.launches < ts.long_until ? 1 : 0;
        if (FTK_ENV(ctx, klt_tail) && atoi(FTK_ENV(ctx, klt_tail)) == 2 && ts.launches < 12) {  // diagnostic
            fprintf(stderr, "[ftk tail] model %d method %d launch %u (context launch %u): host word call %u iterations %u, long_until %u -> long_tail %d\n", model, mi,
                    ts.launches, ctx->tail_call, seen >> 8, seen & 0xFFu, ts.long_until, p.long_tail);
        }
    }
    if (const char *env = FTK_ENV(ctx, klt_tail_class)) {
        p.long_tail = atoi(env) != 0 ? 1 : 0;  // experiment override (the sweep and the policy test pin the class)
    }
    const bool fast_like = opt->method != FTK_METHOD_INVERSE && opt->method != FTK_METHOD_DIRECT;
    int waves = ftk::klt_policy_waves(model, fast_like ? 2 : (opt->method == FTK_METHOD_DIRECT ? 1 : 0), p.consider_luminance, p.long_tail, p.P, n);
    if (p.tree && waves == 1 && fast_like && model != FTK_MODEL_LSSD) {
        waves = std::min(4, std::max(2, (p.P + 63) / 64));  // the throughput mode has no one-wave fast kernel: the generic kernel's waves
    }
    if (const char *env = FTK_ENV(ctx, klt_waves)) {
        waves = atoi(env);  // experiment override
    }
    p.waves_per_feature = waves < 1 ? 1 : (waves > 4 ? 4 : waves);
    // One-wave features are packed several to a workgroup (no barrier between them): the 16 workgroups a CU admits would
    // otherwise cap it at 16 resident features = 4 waves per SIMD, below what the registers allow.
    p.features_per_group = 1;
    const bool pipelined_candidate = model == FTK_MODEL_BASIC && opt->method == FTK_METHOD_INVERSE && p.patch_rows <= 64 && p.patch_cols <= 64;
    if (p.waves_per_feature == 1 && pipelined_candidate) {
        // measured (config 5 shard, 25 000 features): 1 / 2 / 3 / 4 per workgroup = 191 / 175 / 180 / 172 us.  (Round 2 measured no gain:
        // that was before the compile-time geometry removed the SGPR spill traffic — the one-wave kernel is bound by how many
        // features are in flight, not by vector issue: the throughput mode, which drops 20 % of its VALU work, runs no faster.)
        int group = 4;
        if (const char *env = FTK_ENV(ctx, klt_group)) {
            group = atoi(env);  // experiment override
        }
        p.features_per_group = group < 1 ? 1 : (group > 4 ? 4 : group);
    }
    // Basic KLT inverse runs the pipelined kernel (klt_basic_kernels.hip) when its single-wave table
    // builders can hold the patch (<= 64 rows / columns) and coordinates stay exact integers in fp32.
    p.pb_enabled = 0;
    if (model == FTK_MODEL_BASIC && opt->method == FTK_METHOD_INVERSE && p.patch_rows <= 64 && p.patch_cols <= 64) {
        bool small = true;
        for (int i = 0; i < p.n_levels; ++i) {
            small = small && p.ref[i].rows < (1 << 23) && p.ref[i].cols < (1 << 23) && p.cur[i].rows < (1 << 23) && p.cur[i].cols < (1 << 23);
        }
        const char *env = FTK_ENV(ctx, klt_pipelined);
        p.pb_enabled = (small && !(env && atoi(env) == 0)) ? 1 : 0;
    }
    // The `fast` method (the reference's default) of Basic KLT runs the one-wave kernel of klt_fast_kernels.hip at every feature
    // count: with 1 - 2 iterations per level a feature's life is its level entries, which that kernel walks without a barrier and with
    // the next level's windows in flight (2 000 x 13 x 13: 28.8 us on the generic two-wave kernel).  Not in the throughput mode (the
    // generic kernel's instantiations serve it), not for patches whose per-pixel records would crowd the LDS.
    // (the chunked one-wave LSSD levels choose their chain form at run time: quads while every feature of the call is resident at once)
    p.quad_chain = n <= 4096 ? 1 : 0;
    if (const char *env = FTK_ENV(ctx, klt_quad)) {
        p.quad_chain = atoi(env) != 0 ? 1 : 0;  // experiment override
    }
    p.fk_enabled = 0;
    // One wave walks all P pixels of every pass: up to 15 x 15 that beats the generic kernel's 2 - 4 waves at every feature count
    // (2 000 x 13 x 13: 22.6 vs 27.5 us; 10 000: 58.6 vs 82.9 us); larger patches only where the call is throughput-bound anyway
    // (2 000 x 21 x 21: 77.8 vs 54.4 us on two waves).
    // The affine tracker's fast method runs on the same skeleton (the first iteration of a level sweeps and chains 64-pixel chunks
    // through a ring, the later ones six whole rows) from 512 features on: 13 x 13, same box, one wave / generic kernel: 1 000
    // features 34.7 / 36.9 us, 2 000: 77.0 / 83.6, 3 000: 66.0 / 108.3, 5 000: 80.5 / 128.9 — and 100: 83.8 / 75.0, 300: 56.9 / 53.4:
    // a small call IS its slowest feature (31 iterations here), and that one runs 10 % faster on the generic kernel's three waves.
    const bool fk_model = model == FTK_MODEL_BASIC || model == FTK_MODEL_AFFINE;
    if (fk_model && fast_like && !p.tree && p.waves_per_feature == 1 && p.P <= 1024) {  // (the policy table chose ONE wave: the one-wave kernel)
        bool small = true;
        for (int i = 0; i < p.n_levels; ++i) {
            small = small && p.ref[i].rows < (1 << 23) && p.ref[i].cols < (1 << 23) && p.cur[i].rows < (1 << 23) && p.cur[i].cols < (1 << 23);
        }
        const char *env = FTK_ENV(ctx, klt_fast_kernel);
        ftk::KltParams one = p;
        one.features_per_group = 1;
        if (small && !(env && atoi(env) == 0) && ftk::klt_fast_lds_bytes(model, one) <= 40 * 1024) {
            p.fk_enabled = 1;
            p.waves_per_feature = 1;
        }
    }
    // LSSD fast, one wave per feature, no luminance scaling: the chunked sweep / chain variant (a 64-pixel ring instead of all
    // P products of all nine chains in LDS; config 4: 304 -> 242 us)
    p.terms_floats = 0;
    p.px_floats = 3;
    p.lssd_chunked = 0;
    if (model == FTK_MODEL_AFFINE && (opt->method == FTK_METHOD_INVERSE || opt->method == FTK_METHOD_DIRECT)) {
        p.px_floats = 4;
        p.terms_floats = ((p.Ppad / 4 + ftk::kAffineTermsRoundGroups - 1) / ftk::kAffineTermsRoundGroups) * ftk::kAffineTermsRoundGroups * ftk::kAffineTermsGroupFloats;  // products grouped by four pixels (klt_kernels.hip affine_all_terms)
        // The axis tables of the level setup are dead once the iterations start, and the head of the product groups is rewritten
        // by every iteration before it is read: the tables live THERE.  1.2 KB less per feature at 13 x 13 — 22.5 instead of
        // 23.8 KB, i.e. seven instead of six features per CU.  Not for patches of fewer than four pixels (1x1, 1x3, 3x1): their
        // FIRST group holds zero padding (written once per launch) that the tables would overwrite; from 1x5 / 3x3 on the first
        // group with padding starts behind the tables (group g at 100 g floats, tables 12 (rows + cols) floats).
        if (p.P >= 4) {
            p.a0_floats = 0;
        }
    }
    const char *chunk_env = FTK_ENV(ctx, lssd_chunked);
    // (with consider_patch_luminance: the variant that keeps the sampled values in registers — patches up to 512 pixels, klt_kernels.hip kLumChunks)
    if (model == FTK_MODEL_LSSD && fast_like && p.waves_per_feature == 1 && (!p.consider_luminance || (p.P <= 512 && !p.tree)) && !(chunk_env && atoi(chunk_env) == 0)) {
        p.lssd_chunked = 1;
        p.px_floats = 6;
        const int32_t epad = (p.E + 3) & ~3;
        p.terms_floats = epad > 9 * 68 ? epad : 9 * 68;  // the ring, or the extended patch that shares its space at level entry
        p.a0_floats = 0;                                  // (klt_kernels.hip lssd_level_fast_chunked)
    }
    if (p.fk_enabled) {
        int group = 4;  // one-wave features that never meet, several to a workgroup (the 16-workgroups-per-CU cap)
        if (const char *env = FTK_ENV(ctx, klt_group)) {
            group = atoi(env);  // experiment override
        }
        p.features_per_group = group < 1 ? 1 : (group > 4 ? 4 : group);
        ftk::KltParams one = p;
        one.features_per_group = 1;
        const int fit = (int)((size_t)(64 * 1024) / ftk::klt_fast_lds_bytes(model, one));  // a workgroup's LDS stays below 64 KB
        if (p.features_per_group > fit) {
            p.features_per_group = fit < 1 ? 1 : fit;
        }
    } else if (!p.pb_enabled) {
        p.features_per_group = 1;  // the generic kernel's workgroup is one feature ...
        if (p.waves_per_feature == 1) {
            int group = 2;  // ... or, one-wave variants, a few features that never meet (config 4, 10 000 features: 153 / 140.5 / 141 us at 1 / 2 / 4)
            if (const char *env = FTK_ENV(ctx, klt_group)) {
                group = atoi(env);  // experiment override
            }
            p.features_per_group = group < 1 ? 1 : (group > 4 ? 4 : group);
        }
    }
#ifdef FTK_PB_EXTRAS_TIMING_ONLY  // diagnostic builds only (scripts/build_variant.sh): a capacity below the provable maximum gives wrong results
    p.pb_cap_r = p.patch_rows + 2 + (FTK_PB_EXTRAS_TIMING_ONLY);
    p.pb_cap_c = p.patch_cols + 2 + (FTK_PB_EXTRAS_TIMING_ONLY);
#endif
    size_t lds = ftk::klt_lds_bytes(model, opt->method, p);
    if (lds == 0) {
        return fail(ctx, FTK_E_UNSUPPORTED, "klt: unknown variant (model %d, method %d)", model, opt->method);
    }
    // The large-patch form: a patch whose per-pixel arrays exceed a workgroup's 160 KB of LDS (from 41 x 41 for Basic KLT, 37 x 37
    // for the non-fast affine variants) runs the generic multi-wave kernel with those arrays in a per-workgroup slice of device
    // memory (ftk_device.h KltParams::spill) — same code, same sums.  Exact mode only.  FTK_KLT_SPILL=1 forces the form for any
    // patch, 2 also drops the LDS image windows (what happens by itself from about 280 x 280): the tests walk every variant through both.
    const char *spill_env = FTK_ENV(ctx, klt_spill);
    const int spill_force = spill_env ? atoi(spill_env) : 0;
    if (lds > 160 * 1024 || spill_force > 0) {
        p.spill = 1;
        p.tree = 0;
        p.pb_enabled = p.fk_enabled = p.lssd_chunked = 0;
        p.features_per_group = 1;
        int w = (p.P + 63) / 64;
        p.waves_per_feature = w < 2 ? 2 : (w > 4 ? 4 : w);
        ftk::KltParams geometry = p;
        ftk::klt_fill_geometry(geometry);
        p.a0_floats = geometry.a0_floats;
        p.px_floats = 3;
        p.terms_floats = 0;
        if (model == FTK_MODEL_AFFINE && (opt->method == FTK_METHOD_INVERSE || opt->method == FTK_METHOD_DIRECT)) {
            p.px_floats = 4;
            p.terms_floats = ((p.Ppad / 4 + ftk::kAffineTermsRoundGroups - 1) / ftk::kAffineTermsRoundGroups) * ftk::kAffineTermsRoundGroups * ftk::kAffineTermsGroupFloats;
        }
        lds = ftk::klt_lds_bytes(model, opt->method, p);
        if (lds > 150 * 1024 || spill_force > 1) {
            // not even the windows fit: every tap takes the samplers' global-memory path (same arithmetic); a disabled window has one
            // row and no column, which every covered-by-the-window test refuses
            p.rwin_rows = p.cwin_rows = 1;
            p.rwin_cols = p.cwin_cols = 0;
            p.magic_rwc = p.magic_cwc = p.magic_rwq = p.magic_cwq = 0;
            lds = ftk::klt_lds_bytes(model, opt->method, p);
        }
        const size_t floats = ftk::klt_spill_floats(model, p);
        if (floats == 0 || floats > 0xFFFFFFFFull) {
            return fail(ctx, FTK_E_UNSUPPORTED, "klt: patch %dx%d needs %zu floats of device memory per feature", p.patch_rows, p.patch_cols, floats);
        }
        p.spill_stride_floats = (uint32_t)floats;
    }
    return FTK_OK;
}

int make_pyramid(ftk_context *ctx, ftk_pyramid **out) {
    if (!ctx || !out) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid: null context or output");
    }
    ftk_pyramid *pyr = new (std::nothrow) ftk_pyramid();
    if (!pyr) {
        return fail(ctx, FTK_E_OUT_OF_MEMORY, "pyramid: host allocation failed");
    }
    pyr->device = ctx->device;
    *out = pyr;
    return FTK_OK;
}

// The trackers index a level with 32-bit pixel offsets formed on the 24-bit multiplier (klt_common.h px()).
bool level_addressable(int32_t rows, int32_t cols) { return rows < (1 << 24) && cols < (1 << 24) && (long long)rows * cols < (1ll << 32); }

int check_levels(ftk_context *ctx, const ftk_image *levels, int32_t n_levels) {
    if (!levels || n_levels < 1 || n_levels > FTK_MAX_LEVELS) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid: n_levels %d outside [1, %d]", n_levels, FTK_MAX_LEVELS);
    }
    for (int i = 0; i < n_levels; ++i) {
        if (!levels[i].data || levels[i].rows <= 0 || levels[i].cols <= 0) {
            return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid: level %d is empty", i);
        }
        if (!level_addressable(levels[i].rows, levels[i].cols)) {
            return fail(ctx, FTK_E_UNSUPPORTED, "pyramid: level %d (%d x %d) exceeds 2^24 on a side or 2^32 pixels", i, levels[i].rows, levels[i].cols);
        }
    }
    return FTK_OK;
}

}  // namespace

extern "C" {

int ftk_abi_version(void) { return FTK_ABI_VERSION; }

int ftk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

void ftk_default_klt_options(ftk_klt_options *opt) {
    if (!opt) {
        return;
    }
    opt->max_track_points = 500;
    opt->max_iteration = 15;
    opt->max_tolerance_large_step = 3;
    opt->half_rows = 6;
    opt->half_cols = 6;
    opt->max_converge_step = 4e-2f;
    opt->method = FTK_METHOD_FAST;
}

int ftk_context_create(int device, void *stream, ftk_context **out) {
    FTK_TRACE_SCOPE("ftk_context_create");
    if (!out) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "context: null output pointer");
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        return fail(nullptr, FTK_E_NO_DEVICE, "context: no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    }
    if (device < 0) {
        FTK_HIP(nullptr, hipGetDevice(&device));
    }
    if (device >= count) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "context: device %d out of range (have %d)", device, count);
    }
    FTK_HIP(nullptr, hipSetDevice(device));
    ftk_context *ctx = new (std::nothrow) ftk_context();
    if (!ctx) {
        return fail(nullptr, FTK_E_OUT_OF_MEMORY, "context: host allocation failed");
    }
    ctx->device = device;
    if (stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(stream);
        ctx->owns_stream = false;
    } else {
        hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (se != hipSuccess) {
            delete ctx;
            return fail(nullptr, FTK_E_HIP, "context: hipStreamCreate failed: %s", hipGetErrorString(se));
        }
        ctx->owns_stream = true;
    }
    ctx->env.read();
    if (const char *env = FTK_ENV(ctx, reduction)) {  // experiment switch: contexts start in the throughput mode ("tree"); default exact
        ctx->reduction = (strcmp(env, "tree") == 0) ? FTK_REDUCTION_TREE : FTK_REDUCTION_EXACT;
    }
    *out = ctx;
    return FTK_OK;
}

void ftk_context_destroy(ftk_context *ctx) {
    if (!ctx) {
        return;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) {
        (void)hipFree(ctx->scratch);
    }
    if (ctx->match_boxes) {
        (void)hipFree(ctx->match_boxes);
    }
    if (ctx->match_keys) {
        (void)hipFree(ctx->match_keys);
    }
    if (ctx->cosine_ws) {
        (void)hipFree(ctx->cosine_ws);
    }
    if (ctx->direct_table) {
        (void)hipFree(ctx->direct_table);
    }
    if (ctx->direct_feat) {
        (void)hipFree(ctx->direct_feat);
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->sched_iters[k]) {
            (void)hipFree(ctx->sched_iters[k]);
            (void)hipFree(ctx->sched_order[k]);
        }
    }
    if (ctx->klt_spill) {
        (void)hipFree(ctx->klt_spill);
    }
    if (ctx->direct_spread) {
        (void)hipFree(ctx->direct_spread);
    }
    if (ctx->match_pad) {
        (void)hipFree(ctx->match_pad);
    }
    if (ctx->tail_host) {
        (void)hipHostFree(ctx->tail_host);
    }
    if (ctx->tail_dev) {
        (void)hipFree(ctx->tail_dev);
    }
    if (ctx->sched_grid) {
        (void)hipFree(ctx->sched_grid);
    }
    if (ctx->sched_pred) {
        (void)hipFree(ctx->sched_pred);
    }
    if (ctx->sched_claim) {
        (void)hipFree(ctx->sched_claim);
    }
    for (auto &st : ctx->image_stage) {
        if (st.done) {
            (void)hipEventDestroy(st.done);
        }
        if (st.host) {
            (void)hipHostFree(st.host);
        }
    }
    if (ctx->pinned) {
        (void)hipHostFree(ctx->pinned);
    }
    if (ctx->brief_pattern) {
        (void)hipFree(ctx->brief_pattern);
    }
    if (ctx->owns_stream) {
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char *ftk_last_error(const ftk_context *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int ftk_synchronize(ftk_context *ctx) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "synchronize: null context");
    }
    FTK_LOCK(ctx);
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

static int ensure_brief_pattern(ftk_context *ctx, int32_t n_bits, int32_t half);

int ftk_context_refresh_env(ftk_context *ctx) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "context_refresh_env: null context");
    }
    FTK_LOCK(ctx);
    ctx->env.read();
    return FTK_OK;
}

int ftk_set_reduction_mode(ftk_context *ctx, int mode) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "set_reduction_mode: null context");
    }
    FTK_LOCK(ctx);
    if (mode != FTK_REDUCTION_EXACT && mode != FTK_REDUCTION_TREE) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "set_reduction_mode: unknown mode %d", mode);
    }
    ctx->reduction = mode;
    return FTK_OK;
}

int ftk_warmup(ftk_context *ctx, unsigned what) {
    FTK_TRACE_SCOPE("ftk_warmup");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "warmup: null context");
    }
    FTK_LOCK(ctx);
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    if (what & FTK_WARM_KLT) {
        FTK_HIP(ctx, ftk::klt_warm(ctx->stream));
        FTK_HIP(ctx, ftk::klt_basic_warm(ctx->stream));
        FTK_HIP(ctx, ftk::klt_fast_warm(ctx->stream));
        FTK_HIP(ctx, ftk::pyramid_warm(ctx->stream));
        // the staging blocks of the host-buffer entry points, at the size a few thousand features need ...
        // ... and what the upload of one 1080p pyramid stages (ftk_pyramid_upload gathers the levels in the pinned block)
        int rc = ensure_scratch(ctx, 4u << 20);
        if (rc == FTK_OK) {
            rc = ensure_pinned(ctx, 4u << 20);
        }
        // ... and the two pinned slots host images pass through on their way into a pyramid (1 MB each: up to 1024 x 1024)
        for (int k = 0; k < 2 && rc == FTK_OK; ++k) {
            ftk_context::ImageStage *stage = nullptr;
            rc = acquire_image_stage(ctx, 1u << 20, &stage);
        }
        if (rc != FTK_OK) {
            return rc;
        }
    }
    if (what & FTK_WARM_HAMMING) {
        FTK_HIP(ctx, ftk::matcher_warm(ctx->stream));
        FTK_HIP(ctx, ftk::feature_warm(ctx->stream));  // BRIEF descriptors sit in front of the matcher
        int rc = ensure_match_keys(ctx, 4096);
        if (rc == FTK_OK) {
            rc = ensure_scratch(ctx, 4u << 20);
        }
        if (rc == FTK_OK) {
            rc = ensure_pinned(ctx, 4u << 20);
        }
        if (rc == FTK_OK) {
            rc = ensure_brief_pattern(ctx, 256, 8);  // kLength / kHalfPatchSize of the reference's caller (test_descriptor_matcher_brief.cpp:71-72)
        }
        if (rc != FTK_OK) {
            return rc;
        }
    }
    if (what & FTK_WARM_COSINE) {
        FTK_HIP(ctx, ftk::cosine_warm(ctx->stream));
    }
    if (what & FTK_WARM_DIRECT) {
        FTK_HIP(ctx, ftk::direct_warm(ctx->stream));
        FTK_HIP(ctx, ftk::pyramid_warm(ctx->stream));
    }
    if (what & FTK_WARM_FEATURES) {
        FTK_HIP(ctx, ftk::feature_warm(ctx->stream));
    }
    // ... and one REAL launch of every kernel a default-configured object of the family would launch first: besides its code
    // object a kernel's very first launch costs 0.2 - 1 ms of its own (measured: the first TrackFeatures of a process 0.41 / 1.33 ms
    // on two boxes against 0.17 ms for the second tracker).  A 64 x 64 all-zero image, one feature / descriptor / point; results
    // are discarded, failures ignored (warm-up is best effort and must not leave an error behind).
    {
        const std::string saved_error = ctx->error;
        uint8_t *dummy = nullptr;
        constexpr size_t kImg = 64 * 64, kOff = 8192;  // image | feature block | descriptors
        if (hipMalloc(reinterpret_cast<void **>(&dummy), kOff + 8192) == hipSuccess &&
            hipMemsetAsync(dummy, 0, kOff + 8192, ctx->stream) == hipSuccess) {
            static_assert(kImg <= kOff, "dummy image fits in front of the feature block");
            ftk_image level = {dummy, 64, 64};
            ftk_pyramid *pyr = nullptr;
            float *d_uv = reinterpret_cast<float *>(dummy + kOff);          // ref (u, v) = (0, 0): never dereferenced out of range
            float *d_cur = d_uv + 2, *d_out = d_uv + 4;
            uint8_t *d_st = dummy + kOff + 64, *d_sto = dummy + kOff + 128;
            uint32_t *d_desc_ref = reinterpret_cast<uint32_t *>(dummy + kOff + 256), *d_desc_cur = d_desc_ref + 16;
            int32_t *d_idx = reinterpret_cast<int32_t *>(dummy + kOff + 512);
            float *d_fref = reinterpret_cast<float *>(dummy + kOff + 1024), *d_fcur = d_fref + 256;
            if (ftk_pyramid_wrap_device(ctx, &level, 1, &pyr) == FTK_OK) {
                if (what & FTK_WARM_KLT) {
                    ftk_klt_options opt;
                    ftk_default_klt_options(&opt);
                    for (int model = FTK_MODEL_BASIC; model <= FTK_MODEL_LSSD; ++model) {
                        for (int method = FTK_METHOD_INVERSE; method <= FTK_METHOD_FAST; ++method) {
                            opt.method = method;
                            (void)ftk_klt_track_device(ctx, model, &opt, pyr, pyr, d_uv, d_cur, d_out, d_st, d_sto, 1, nullptr, 0, 0, nullptr);
                        }
                    }
                }
                if (what & (FTK_WARM_HAMMING | FTK_WARM_FEATURES)) {
                    (void)ftk_brief_compute_device(ctx, pyr, 0, d_uv, 1, 256, 8, d_desc_ref);
                }
                if (what & FTK_WARM_FEATURES) {
                    float corner[2];
                    int32_t found = 0;
                    (void)ftk_harris_detect(ctx, pyr, 0, 1, 25, 40.0f, corner, &found);
                }
                if (what & FTK_WARM_DIRECT) {
                    ftk_direct_options dopt;
                    ftk_default_direct_options(&dopt);
                    const float K[4] = {64.0f, 64.0f, 32.0f, 32.0f}, point[3] = {0.0f, 0.0f, 1.0f}, ruv[2] = {32.0f, 32.0f};
                    float cuv[2] = {32.0f, 32.0f}, q[4] = {1.0f, 0.0f, 0.0f, 0.0f}, t[3] = {0.0f, 0.0f, 0.0f};
                    uint8_t st = 0;
                    (void)ftk_direct_track(ctx, &dopt, pyr, pyr, K, point, ruv, cuv, 1, q, t, &st, 0, nullptr);
                }
                ftk_pyramid_destroy(pyr);
            }
            if (what & FTK_WARM_HAMMING) {
                (void)ftk_hamming_match_device(ctx, d_desc_ref, 1, d_desc_cur, 1, 8, 256, 60.0f, nullptr, nullptr, 40, 40, d_idx, nullptr);
                (void)ftk_hamming_match_device(ctx, d_desc_ref, 1, d_desc_cur, 1, 8, 256, 60.0f, d_uv, d_cur, 40, 40, d_idx, nullptr);
            }
            if (what & FTK_WARM_COSINE) {
                for (int dim : {256, 128}) {
                    (void)ftk_cosine_match_device(ctx, d_fref, 1, d_fcur, 1, dim, 0.5f, nullptr, nullptr, 40, 40, d_idx);
                    (void)ftk_cosine_match_device(ctx, d_fref, 1, d_fcur, 1, dim, 0.5f, d_uv, d_cur, 40, 40, d_idx);
                }
            }
            (void)hipStreamSynchronize(ctx->stream);
        }
        if (dummy) {
            (void)hipFree(dummy);
        }
        ctx->error = saved_error;
    }
    if (ctx->pinned && ctx->scratch) {
        // first copies in both directions between the staging blocks (the copy path's first use is not free either)
        // (an image-sized one: copies beyond a few KB take another path in the runtime than small ones, and the first 361 KB
        // upload of a process was measured at 5.8 - 7.9 ms)
        const size_t probe = ctx->pinned_bytes < ctx->scratch_bytes ? ctx->pinned_bytes : ctx->scratch_bytes;
        FTK_HIP(ctx, hipMemcpyAsync(ctx->scratch, ctx->pinned, probe, hipMemcpyHostToDevice, ctx->stream));
        FTK_HIP(ctx, hipMemcpyAsync(ctx->pinned, ctx->scratch, probe, hipMemcpyDeviceToHost, ctx->stream));
        void *tmp = nullptr;  // and one image-sized allocation: what every ftk_pyramid_upload / ftk_pyramid_build makes
        if (hipMalloc(&tmp, 4u << 20) == hipSuccess) {
            (void)hipFree(tmp);
        }
    }
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

/* ---- pyramids ------------------------------------------------------------------------------ */

int ftk_pyramid_upload(ftk_context *ctx, const ftk_image *host_levels, int32_t n_levels, ftk_pyramid **out) {
    FTK_TRACE_SCOPE("ftk_pyramid_upload");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "pyramid_upload: null context");
    }
    FTK_LOCK(ctx);
    int rc = check_levels(ctx, host_levels, n_levels);
    if (rc != FTK_OK) {
        return rc;
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    size_t offsets[FTK_MAX_LEVELS];
    size_t total = 0;
    for (int i = 0; i < n_levels; ++i) {
        offsets[i] = total;
        total += align_up((size_t)host_levels[i].rows * host_levels[i].cols, 256);
    }
    ftk_pyramid *pyr = nullptr;
    rc = make_pyramid(ctx, &pyr);
    if (rc != FTK_OK) {
        return rc;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&pyr->owned), total);
    if (e != hipSuccess) {
        delete pyr;
        return fail(ctx, FTK_E_OUT_OF_MEMORY, "pyramid_upload: hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
    }
    pyr->n_levels = n_levels;
    // gather the levels in pinned staging, then ONE H2D copy of the whole pyramid
    if (ensure_pinned(ctx, total) != FTK_OK) {
        ftk_pyramid_destroy(pyr);
        return FTK_E_OUT_OF_MEMORY;
    }
    uint8_t *staging = static_cast<uint8_t *>(ctx->pinned);
    for (int i = 0; i < n_levels; ++i) {
        const size_t bytes = (size_t)host_levels[i].rows * host_levels[i].cols;
        memcpy(staging + offsets[i], host_levels[i].data, bytes);
        pyr->levels[i].data = pyr->owned + offsets[i];
        pyr->levels[i].rows = host_levels[i].rows;
        pyr->levels[i].cols = host_levels[i].cols;
    }
    e = hipMemcpyAsync(pyr->owned, staging, total, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        e = hipStreamSynchronize(ctx->stream);  // staging and the caller's buffers are free again on return
    }
    if (e != hipSuccess) {
        ftk_pyramid_destroy(pyr);
        return fail(ctx, FTK_E_HIP, "pyramid_upload: copy failed: %s", hipGetErrorString(e));
    }
    *out = pyr;
    return FTK_OK;
}

int ftk_pyramid_wrap_device(ftk_context *ctx, const ftk_image *device_levels, int32_t n_levels, ftk_pyramid **out) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "pyramid_wrap_device: null context");
    }
    FTK_LOCK(ctx);
    int rc = check_levels(ctx, device_levels, n_levels);
    if (rc != FTK_OK) {
        return rc;
    }
    ftk_pyramid *pyr = nullptr;
    rc = make_pyramid(ctx, &pyr);
    if (rc != FTK_OK) {
        return rc;
    }
    pyr->n_levels = n_levels;
    for (int i = 0; i < n_levels; ++i) {
        pyr->levels[i].data = device_levels[i].data;
        pyr->levels[i].rows = device_levels[i].rows;
        pyr->levels[i].cols = device_levels[i].cols;
    }
    *out = pyr;
    return FTK_OK;
}

int ftk_pyramid_build(ftk_context *ctx, const uint8_t *image, int32_t rows, int32_t cols, int32_t n_levels, int image_on_device,
                      ftk_pyramid **out) {
    FTK_TRACE_SCOPE("ftk_pyramid_build");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "pyramid_build: null context");
    }
    FTK_LOCK(ctx);
    if (!image || rows <= 0 || cols <= 0 || n_levels < 1 || n_levels > FTK_MAX_LEVELS || !out) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid_build: bad image or level count");
    }
    if (!level_addressable(rows, cols)) {
        return fail(ctx, FTK_E_UNSUPPORTED, "pyramid_build: image %d x %d exceeds 2^24 on a side or 2^32 pixels", rows, cols);
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    int32_t lrows[FTK_MAX_LEVELS], lcols[FTK_MAX_LEVELS];
    size_t offsets[FTK_MAX_LEVELS];
    size_t total = 0;
    lrows[0] = rows;
    lcols[0] = cols;
    for (int i = 0; i < n_levels; ++i) {
        if (i > 0) {
            lrows[i] = lrows[i - 1] / 2;
            lcols[i] = lcols[i - 1] / 2;
            if (lrows[i] <= 0 || lcols[i] <= 0) {
                return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid_build: level %d would be empty", i);
            }
        }
        offsets[i] = total;
        if (i > 0 || !image_on_device) {
            total += align_up((size_t)lrows[i] * lcols[i], 256);
        }
    }
    ftk_pyramid *pyr = nullptr;
    int rc = make_pyramid(ctx, &pyr);
    if (rc != FTK_OK) {
        return rc;
    }
    if (total > 0) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&pyr->owned), total);
        if (e != hipSuccess) {
            delete pyr;
            return fail(ctx, FTK_E_OUT_OF_MEMORY, "pyramid_build: hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
        }
    }
    pyr->n_levels = n_levels;
    hipError_t e = hipSuccess;
    // A host image goes through a pinned staging slot: the CPU copies it there (the caller's buffer is free on return), the
    // pyramid launch reads the slot over PCIe and keeps level 0 — no staged hipMemcpy of pageable memory, no stream
    // synchronisation (CreateImagePyramid x 2 sits inside the reference's timed region, test_optical_flow.cpp:69-73: 57 us per
    // build before).  One-level pyramids and FTK_PYRAMID_ZEROCOPY=0 / FTK_PYRAMID_FUSED=0 keep the copy.
    ftk_context::ImageStage *stage = nullptr;
    const bool stage_allowed = !(FTK_ENV(ctx, pyramid_zerocopy) && atoi(FTK_ENV(ctx, pyramid_zerocopy)) == 0);
    if (!image_on_device && stage_allowed && n_levels >= 2 && ftk::pyramid_fused_enabled()) {
        rc = acquire_image_stage(ctx, (size_t)rows * cols, &stage);
        if (rc != FTK_OK) {
            ftk_pyramid_destroy(pyr);
            return rc;
        }
    }
    if (image_on_device) {
        pyr->levels[0].data = image;
    } else {
        if (stage) {
            memcpy(stage->host, image, (size_t)rows * cols);
        } else {
            e = hipMemcpyAsync(pyr->owned, image, (size_t)rows * cols, hipMemcpyHostToDevice, ctx->stream);
        }
        pyr->levels[0].data = pyr->owned;
    }
    pyr->levels[0].rows = rows;
    pyr->levels[0].cols = cols;
    uint8_t *level_ptr[FTK_MAX_LEVELS] = {nullptr};
    for (int i = 1; i < n_levels; ++i) {
        level_ptr[i] = pyr->owned + offsets[i];
        pyr->levels[i].data = level_ptr[i];
        pyr->levels[i].rows = lrows[i];
        pyr->levels[i].cols = lcols[i];
    }
    if (e == hipSuccess) {
        if (stage) {
            e = ftk::pyramid_build_levels_launch(stage->device_view, rows, cols, level_ptr, n_levels, ctx->stream, pyr->owned);
            if (e == hipSuccess) {
                e = hipEventRecord(stage->done, ctx->stream);
                stage->busy = e == hipSuccess;
            }
        } else {
            e = ftk::pyramid_build_levels_launch(pyr->levels[0].data, rows, cols, level_ptr, n_levels, ctx->stream);  // one launch for all levels
        }
    }
    if (e == hipSuccess && !image_on_device && !stage) {
        e = hipStreamSynchronize(ctx->stream);  // host image may be released by the caller
    }
    if (e != hipSuccess) {
        ftk_pyramid_destroy(pyr);
        return fail(ctx, FTK_E_HIP, "pyramid_build: %s", hipGetErrorString(e));
    }
    *out = pyr;
    return FTK_OK;
}

int ftk_pyramid_update(ftk_context *ctx, ftk_pyramid *pyr, const uint8_t *image, int image_location) {
    FTK_TRACE_SCOPE("ftk_pyramid_update");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "pyramid_update: null context");
    }
    FTK_LOCK(ctx);
    if (!pyr || !image || image_location < FTK_IMAGE_HOST || image_location > FTK_IMAGE_HOST_ASYNC) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid_update: null pyramid / image or unknown image location %d", image_location);
    }
    if (pyr->device != ctx->device) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid_update: the pyramid lives on another device");
    }
    if (!pyr->owned || pyr->levels[0].data != pyr->owned) {
        return fail(ctx, FTK_E_UNSUPPORTED, "pyramid_update: only pyramids that own their level 0 (ftk_pyramid_upload, ftk_pyramid_build of a host image) can be refilled");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *level_ptr[FTK_MAX_LEVELS] = {nullptr};
    bool halves = true;  // every level is the floor-half of the one above it (true for every pyramid this library builds)
    for (int i = 1; i < pyr->n_levels; ++i) {
        level_ptr[i] = const_cast<uint8_t *>(pyr->levels[i].data);
        halves = halves && pyr->levels[i].rows == pyr->levels[i - 1].rows / 2 && pyr->levels[i].cols == pyr->levels[i - 1].cols / 2;
    }
    if (!halves) {
        return fail(ctx, FTK_E_UNSUPPORTED, "pyramid_update: the levels of this pyramid are not successive halves (uploaded with another geometry)");
    }
    // A frame in PINNED host memory (FTK_IMAGE_HOST_ASYNC) is read by the pyramid launch itself when the device can address it:
    // the copy engine takes ~20 us per 300 KB frame, the kernel's own PCIe read a third of that, and a launch gap goes with it.
    // FTK_PYRAMID_ZEROCOPY=0 keeps the copy (experiment switch); pageable or unmapped memory takes it anyway.
    const uint8_t *direct_src = nullptr;
    const bool zero_copy_allowed = !(FTK_ENV(ctx, pyramid_zerocopy) && atoi(FTK_ENV(ctx, pyramid_zerocopy)) == 0);
    if (image_location == FTK_IMAGE_HOST_ASYNC && zero_copy_allowed && pyr->n_levels >= 2 && ftk::pyramid_fused_enabled()) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, image) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer != nullptr) {
            direct_src = static_cast<const uint8_t *>(attr.devicePointer);
        } else {
            (void)hipGetLastError();  // not an error of this call: the copy path below serves the pointer
        }
    }
    ftk_context::ImageStage *stage = nullptr;
    if (image_location == FTK_IMAGE_HOST && zero_copy_allowed && pyr->n_levels >= 2 && ftk::pyramid_fused_enabled()) {
        // a pageable frame: CPU copy into a pinned slot (the caller's buffer is free on return), read by the launch; no synchronisation
        const int rc = acquire_image_stage(ctx, (size_t)pyr->levels[0].rows * pyr->levels[0].cols, &stage);
        if (rc != FTK_OK) {
            return rc;
        }
    }
    if (stage != nullptr) {
        memcpy(stage->host, image, (size_t)pyr->levels[0].rows * pyr->levels[0].cols);
        FTK_HIP(ctx, ftk::pyramid_build_levels_launch(stage->device_view, pyr->levels[0].rows, pyr->levels[0].cols, level_ptr, pyr->n_levels, ctx->stream, pyr->owned));
        FTK_HIP(ctx, hipEventRecord(stage->done, ctx->stream));
        stage->busy = true;
        return FTK_OK;
    }
    if (direct_src != nullptr) {
        FTK_HIP(ctx, ftk::pyramid_build_levels_launch(direct_src, pyr->levels[0].rows, pyr->levels[0].cols, level_ptr, pyr->n_levels, ctx->stream, pyr->owned));
    } else {
        const size_t bytes0 = (size_t)pyr->levels[0].rows * pyr->levels[0].cols;
        FTK_HIP(ctx, hipMemcpyAsync(pyr->owned, image, bytes0, image_location == FTK_IMAGE_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                    ctx->stream));
        FTK_HIP(ctx, ftk::pyramid_build_levels_launch(pyr->levels[0].data, pyr->levels[0].rows, pyr->levels[0].cols, level_ptr, pyr->n_levels, ctx->stream));
    }
    if (image_location == FTK_IMAGE_HOST) {
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller may release or rewrite the host image on return
    }
    return FTK_OK;
}

int ftk_pyramid_levels(const ftk_pyramid *pyr) { return pyr ? pyr->n_levels : 0; }

int ftk_pyramid_level(const ftk_pyramid *pyr, int32_t level, ftk_image *out) {
    if (!pyr || !out || level < 0 || level >= pyr->n_levels) {
        return FTK_E_INVALID_ARGUMENT;
    }
    out->data = pyr->levels[level].data;
    out->rows = pyr->levels[level].rows;
    out->cols = pyr->levels[level].cols;
    return FTK_OK;
}

int ftk_pyramid_download_level(ftk_context *ctx, const ftk_pyramid *pyr, int32_t level, uint8_t *host_out) {
    FTK_TRACE_SCOPE("ftk_pyramid_download_level");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "pyramid_download_level: null context");
    }
    FTK_LOCK(ctx);
    if (!pyr || !host_out || level < 0 || level >= pyr->n_levels) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "pyramid_download_level: bad arguments");
    }
    const size_t bytes = (size_t)pyr->levels[level].rows * pyr->levels[level].cols;
    FTK_HIP(ctx, hipMemcpyAsync(host_out, pyr->levels[level].data, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

void ftk_pyramid_destroy(ftk_pyramid *pyr) {
    FTK_TRACE_SCOPE("ftk_pyramid_destroy");
    if (!pyr) {
        return;
    }
    if (pyr->owned) {
        (void)hipSetDevice(pyr->device);
        (void)hipFree(pyr->owned);
    }
    delete pyr;
}

/* ---- KLT ----------------------------------------------------------------------------------- */

int ftk_klt_track_device(ftk_context *ctx, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur,
                         const float *d_ref_uv, const float *d_cur_uv_in, float *d_cur_uv_out, const uint8_t *d_status_in,
                         uint8_t *d_status_out, int32_t n, const float *prior, int consider_luminance, int single_level, uint32_t *d_iters) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "klt_track_device: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_device: negative feature count");
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!d_ref_uv || !d_cur_uv_in || !d_cur_uv_out || !d_status_in || !d_status_out) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_device: null buffer");
    }
    // The kernels read and write a feature's (u, v) as ONE 8-byte access (include/ftk.h: "8-byte aligned"): a pair array at an odd
    // float offset — legal through round 3 — is refused here instead of becoming misaligned 64-bit accesses on the device.
    if (((reinterpret_cast<uintptr_t>(d_ref_uv) | reinterpret_cast<uintptr_t>(d_cur_uv_in) | reinterpret_cast<uintptr_t>(d_cur_uv_out)) & 7u) != 0) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_device: the (u, v) arrays must be 8-byte aligned (ref %p, in %p, out %p)", (const void *)d_ref_uv,
                    (const void *)d_cur_uv_in, (const void *)d_cur_uv_out);
    }
    ftk::KltParams p;
    const int rc = fill_klt_params(ctx, model, opt, ref, cur, n, prior, consider_luminance, single_level, &p);
    if (rc != FTK_OK) {
        return rc;
    }
    p.tree = (ctx->reduction == FTK_REDUCTION_TREE && !p.spill) ? 1 : 0;
    {
        // this launch's number, for the report of its longest feature (tail-aware wave policy, fill_klt_params)
        if (!ctx->tail_host) {
            void *host = nullptr;
            if (hipHostMalloc(&host, 64, hipHostMallocDefault) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&ctx->tail_dev), 64) == hipSuccess) {
                memset(host, 0, 64);
                ctx->tail_host = static_cast<uint32_t *>(host);
                (void)hipMemsetAsync(ctx->tail_dev, 0, 64, ctx->stream);
            } else {
                (void)hipGetLastError();
                if (host) {
                    (void)hipHostFree(host);
                }
            }
        }
        if (ctx->tail_host && ctx->tail_dev) {
            ctx->tail_call = (ctx->tail_call + 1u) & 0xFFFFFFu;
            if (ctx->tail_call == 0u) {
                ctx->tail_call = 1u;  // (after 16 M launches the device word's running maximum starts over with the host's)
                (void)hipMemsetAsync(ctx->tail_dev, 0, 64, ctx->stream);
            }
            const int mi = opt->method == FTK_METHOD_INVERSE ? 0 : (opt->method == FTK_METHOD_DIRECT ? 1 : 2);
            ++ctx->tail[model][mi].launches;
            p.tail_dev = ctx->tail_dev + (model * 3 + mi);    // a word per variant
            p.tail_host = ctx->tail_host + (model * 3 + mi);  // (hipHostMalloc'ed memory is device-visible under the same address)
            p.tail_call = ctx->tail_call;
        }
    }
    p.ref_uv = d_ref_uv;
    p.cur_uv_in = d_cur_uv_in;
    p.cur_uv_out = d_cur_uv_out;
    p.status_in = d_status_in;
    p.status_out = d_status_out;
    p.iters = d_iters;
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    if (p.spill) {
        // Large patches: a slice of device memory per launch slot.  All features at once while that stays within a budget (4 GB;
        // FTK_KLT_SPILL_BUDGET_MB), otherwise in batches of consecutive features — a feature's result does not depend on the others.
        const size_t per = sizeof(float) * (size_t)p.spill_stride_floats;
        size_t budget = (size_t)4096 << 20;
        if (const char *env = FTK_ENV(ctx, klt_spill_budget_mb)) {
            budget = (size_t)(atoll(env) > 0 ? atoll(env) : 1) << 20;
        }
        size_t batch = budget / per;
        batch = batch < 1 ? 1 : (batch > (size_t)n ? (size_t)n : batch);
        if (batch * per > ctx->klt_spill_bytes) {
            // the slices would have to grow: a hipFree / hipMalloc (and a synchronisation) that a stream capture cannot contain
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
                return fail(ctx, FTK_E_UNSUPPORTED, "klt_track_device: a %d x %d patch needs %zu MB of device memory for its per-feature slices, which cannot be "
                            "allocated while the stream is being captured: make one such call before the capture (the buffer is kept)", p.patch_rows, p.patch_cols,
                            (batch * per) >> 20);
            }
        }
        const int rc_buf = ftk_ensure_device_buffer(ctx, &ctx->klt_spill, &ctx->klt_spill_bytes, batch * per);
        if (rc_buf != FTK_OK) {
            return rc_buf;
        }
        p.spill_base = static_cast<float *>(ctx->klt_spill);
        ctx->sched_calls = 0;  // no launch order for these calls; a later ordinary call starts its history over
        ctx->sched_n = 0;
        for (size_t b0 = 0; b0 < (size_t)n; b0 += batch) {
            const size_t nb = (size_t)n - b0 < batch ? (size_t)n - b0 : batch;
            ftk::KltParams q = p;
            q.n = (int32_t)nb;
            q.ref_uv = p.ref_uv + 2 * b0;
            q.cur_uv_in = p.cur_uv_in + 2 * b0;
            q.cur_uv_out = p.cur_uv_out + 2 * b0;
            q.status_in = p.status_in + b0;
            q.status_out = p.status_out + b0;
            q.iters = p.iters ? p.iters + b0 : nullptr;
            q.n_track = (size_t)p.n_track > b0 ? (uint32_t)((size_t)p.n_track - b0 < nb ? (size_t)p.n_track - b0 : nb) : 0u;  // kMaxTrackPointsNumber is a cap on the whole list
            const hipError_t e = ftk::klt_launch(model, opt->method, q, ctx->stream);
            if (e != hipSuccess) {
                return fail(ctx, e == hipErrorOutOfMemory ? FTK_E_OUT_OF_MEMORY : FTK_E_HIP, "klt launch (large patch) failed: %s", hipGetErrorString(e));
            }
        }
        return FTK_OK;
    }
    {
        // Launch order.  A call's time is bulk + tail: features run a data-dependent number of Gauss-Newton iterations
        // (config 3: mean 6.7, one feature 52), a launch in list order starts the long ones wherever they happen to sit,
        // and the grid drains while they finish.  Trackers are called frame after frame on (nearly) the same feature list
        // and a feature that needed many iterations tends to need many again, so the launch slots go through a permutation:
        // longest first by an EARLIER call's iteration counts.  No launch of its own: call k's tracker launch carries one
        // extra workgroup (block 0, klt_common.h klt_order_block) that sorts call k - 1's counts while the features of call k
        // run, and call k + 1 uses the result — so from the third call with the same feature count on, with a predictor two
        // calls old.  Which slot runs a feature changes nothing in its arithmetic.  Only for calls with more features than
        // fit the chip at once; FTK_KLT_SCHED=0 keeps list order.
        const bool sched_allowed = !(FTK_ENV(ctx, klt_sched) && atoi(FTK_ENV(ctx, klt_sched)) == 0);
        // From kSchedMinFeatures on — or, when this variant's recent calls had a long feature (p.long_tail: the kernels report it,
        // fill_klt_params), already from kSchedMinLongTail: multi-wave features of a few thousand do NOT all fit the chip at once, and
        // a 50-iteration feature that starts in the second round ends the launch that much later (the reference's example pair, same
        // box, order from 4 096 / from 1 024: affine inverse 2 000 features 165.9 / 138.8 us, affine direct 3 000: 174.7 / 136.0, LSSD
        // fast 3 000: 143.2 / 113.2, Basic fast 3 000: 60.9 / 52.3; the synthetic scene's LSSD / affine variants -4 ... -15 %).  Calls
        // without a tail keep list order there: the order costs every feature one more dependent load (Basic variants +3 ... 4 %).
        const uint32_t sched_min = FTK_ENV(ctx, klt_sched_min) ? (uint32_t)atoi(FTK_ENV(ctx, klt_sched_min))  // (experiment override)
                                                               : (p.long_tail ? kSchedMinLongTail : kSchedMinFeatures);
        if (sched_allowed && p.n_track >= sched_min && n <= kSchedMaxFeatures) {
            if ((size_t)n > ctx->sched_capacity) {
                FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                for (int k = 0; k < 2; ++k) {
                    if (ctx->sched_iters[k]) {
                        (void)hipFree(ctx->sched_iters[k]);
                        (void)hipFree(ctx->sched_order[k]);
                        ctx->sched_iters[k] = nullptr;
                        ctx->sched_order[k] = nullptr;
                    }
                }
                if (ctx->sched_claim) {
                    (void)hipFree(ctx->sched_claim);
                    ctx->sched_claim = nullptr;
                }
                if (ctx->sched_pred) {
                    (void)hipFree(ctx->sched_pred);
                    ctx->sched_pred = nullptr;
                }
                ctx->sched_capacity = 0;
                ctx->sched_n = 0;
                const size_t cap = ((size_t)n + 4095) / 4096 * 4096;
                // position-keyed slot swaps: a claim word per launch slot, and (once) the two tables of iteration counts by position
                FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sched_claim), sizeof(uint32_t) * cap));
                FTK_HIP(ctx, hipMemsetAsync(ctx->sched_claim, 0, sizeof(uint32_t) * cap, ctx->stream));
                FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sched_pred), cap));
                if (!ctx->sched_grid) {
                    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sched_grid), sizeof(uint32_t) * (kSchedTableWords + kSchedOrderWords)));
                    FTK_HIP(ctx, hipMemsetAsync(ctx->sched_grid, 0, sizeof(uint32_t) * kSchedTableWords, ctx->stream));
                }
                for (int k = 0; k < 2; ++k) {
                    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sched_iters[k]), sizeof(uint32_t) * cap));
                    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sched_order[k]), sizeof(int32_t) * cap));
                    // never-written entries must still be valid feature ids (0) and valid counts, whatever happens to a launch
                    FTK_HIP(ctx, hipMemsetAsync(ctx->sched_iters[k], 0, sizeof(uint32_t) * cap, ctx->stream));
                    FTK_HIP(ctx, hipMemsetAsync(ctx->sched_order[k], 0, sizeof(int32_t) * cap, ctx->stream));
                }
                ctx->sched_capacity = cap;
            }
            if (ctx->sched_n != n) {
                ctx->sched_n = n;
                ctx->sched_calls = 0;
            }
            const uint32_t k = ctx->sched_calls++;
            // Position-keyed swaps ride on every such call, whatever the list did since the last one (FTK_KLT_SWAP=0: off).  Call
            // numbers start at 4 (an all-zero grid / claim word is never "recent") and tag 23 bits of a claim word: the claims are
            // wiped before a tag could repeat.
            const bool swap_allowed = !(FTK_ENV(ctx, klt_swap) && atoi(FTK_ENV(ctx, klt_swap)) == 0);
            // (never inside a stream capture: a replayed launch would carry this call's number again and read its own old claims)
            hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(ctx->stream, &capture) != hipSuccess) {
                (void)hipGetLastError();
                capture = hipStreamCaptureStatusActive;  // unknown: be safe
            }
            // (nor when the results overwrite the reference positions: both sides of a trade must read the same positions)
            const char *ref_lo = reinterpret_cast<const char *>(p.ref_uv), *out_lo = reinterpret_cast<const char *>(p.cur_uv_out);
            const size_t uv_span = sizeof(float) * 2 * (size_t)n;
            const bool ref_untouched = ref_lo + uv_span <= out_lo || out_lo + uv_span <= ref_lo;
            // Multi-wave features only: there a feature is tens of microseconds long and iteration counts have heavy tails
            // (config 3: 192 / 207 -> 149 / 166 us with no / a stale launch order, +0.5 % with a fitting one); the one-wave kernels
            // run 10 000 - 25 000 cheap features, every late one of which would pay a table look-up for a 3 % gain at best
            // (config 4: +2.9 % with a fitting order, -3 % without; config 5: +1 %).
            // EVERY such call (outside a capture) leaves its iteration counts in the position table — one or two atomics per feature —
            // so that the next one can order or trade by position whatever kernel either of them runs.
            const bool recording = capture == hipStreamCaptureStatusNone && ctx->sched_grid && ctx->sched_claim;
            uint32_t last_recorded = 0;
            if (recording) {
                if (ctx->sched_call < 4u) {
                    ctx->sched_call = 4u;
                }
                ++ctx->sched_call;
                if ((ctx->sched_call & 0x7FFFFFu) < 4u) {
                    FTK_HIP(ctx, hipMemsetAsync(ctx->sched_claim, 0, sizeof(uint32_t) * ctx->sched_capacity, ctx->stream));
                    FTK_HIP(ctx, hipMemsetAsync(ctx->sched_grid, 0, sizeof(uint32_t) * kSchedTableWords, ctx->stream));
                    ctx->sched_call += 4u;
                    ctx->sched_recorded = 0;
                }
                p.sched_grid = ctx->sched_grid;
                p.sched_call = ctx->sched_call;
                last_recorded = ctx->sched_recorded;
                ctx->sched_recorded = ctx->sched_call;
            }
            if (recording && swap_allowed && p.waves_per_feature >= 2 && ref_untouched && n > 1024 + 512) {
                p.sched_flags = ctx->sched_grid + (2u << 16);
                p.sched_claim = ctx->sched_claim;
            }
            p.sched_iters = ctx->sched_iters[k & 1];          // this call's counts
            if (k >= 1) {                                     // sort the previous call's counts beside this call's features
                p.sort_iters = ctx->sched_iters[(k - 1) & 1];
                p.sort_order_out = ctx->sched_order[(k - 1) & 1];
                // The spatial (tile) order reads the reference positions in two passes while the feature workgroups of the same
                // launch write cur_uv_out: with one position buffer updated in place (ref == out, allowed by include/ftk.h) a
                // feature crossing a tile boundary in between would make the histogram and the scatter disagree — duplicates,
                // stale entries, a write past order[n - 1].  Such a call gets the iteration-count / identity order instead.
                p.sort_ref_uv = ref_untouched ? p.ref_uv : nullptr;
            }
            const int order_mode = FTK_ENV(ctx, klt_order) ? atoi(FTK_ENV(ctx, klt_order)) : -1;  // experiment: 0 = never, 1 = always
            const bool use_order = order_mode >= 0 ? order_mode != 0 : true;
            if (k >= 2 && use_order) {                        // made during the previous call from the counts before it
                p.order = ctx->sched_order[k & 1];
            } else if (use_order && recording && last_recorded != 0u && last_recorded + 1u == ctx->sched_call && ctx->sched_pred && model != FTK_MODEL_BASIC &&
                       p.sched_claim == nullptr) {
                // (LSSD and affine KLT: their iteration counts have tails — config 4 without history 206 -> 183 us, with luminance
                // 357 -> 315; Basic KLT's are flat on most scenes and the ~10 us of the two launches would buy nothing — config 5 shard
                // 181 -> 190; the multi-wave kernels trade slots by position inside the launch instead)
                // No index-keyed order (the feature count has just changed, or these are the first calls): order THIS call by what the
                // last call left at its features' positions — two small launches in front of the tracker's (klt_kernels.hip
                // klt_position_order_launch; FTK_KLT_POSITION_ORDER=0: list order as before).  The buffer is the one an index-keyed
                // order of this call would have used: nobody else writes it during this call.
                const bool position_order = !(FTK_ENV(ctx, klt_position_order) && atoi(FTK_ENV(ctx, klt_position_order)) == 0);
                if (position_order) {
                    const uint32_t *last_table = ctx->sched_grid + (((ctx->sched_call - 1u) & 1u) << 16);
                    FTK_HIP(ctx, ftk::klt_position_order_launch(p.ref_uv, n, last_table, ctx->sched_call - 1u, ctx->sched_pred, ctx->sched_grid + kSchedTableWords,
                                                                ctx->sched_order[k & 1], ctx->stream));
                    p.order = ctx->sched_order[k & 1];
                }
            }
            if (const char *dump = FTK_ENV(ctx, klt_swap_dump)) {  // diagnostic: how many trades the PREVIOUS launch of this context made
                if (p.sched_claim != nullptr && ctx->sched_call > 5u) {
                    std::vector<uint32_t> h((size_t)n);
                    FTK_HIP(ctx, hipMemcpyAsync(h.data(), ctx->sched_claim, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
                    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    const uint32_t last = (ctx->sched_call - 1u) & 0x7FFFFFu;
                    size_t trades = 0, own = 0;
                    for (uint32_t w : h) {
                        if ((w >> 9) == last) {
                            ((w & 0x1FFu) == 0x1FFu ? own : trades) += 1;
                        }
                    }
                    if (FILE *f = fopen(dump, "w")) {
                        fprintf(f, "%zu %zu\n", trades, own);
                        fclose(f);
                    }
                }
            }
            if (const char *dump = FTK_ENV(ctx, klt_sched_dump)) {  // diagnostic: the permutation in use and the counts it came from
                if (k >= 2) {
                    std::vector<int32_t> h((size_t)n * 2);
                    FTK_HIP(ctx, hipMemcpyAsync(h.data(), ctx->sched_order[k & 1], sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
                    FTK_HIP(ctx, hipMemcpyAsync(h.data() + n, ctx->sched_iters[k & 1], sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
                    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if (FILE *f = fopen(dump, "wb")) {
                        fwrite(h.data(), sizeof(int32_t), h.size(), f);
                        fclose(f);
                    }
                }
            }
        }
    }
#ifdef FTK_STAMPS
    // diagnostic build: per-phase cycle totals (s_memtime ticks at 100 MHz) averaged over features, to stderr
    unsigned long long *d_stamps = nullptr;
    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d_stamps), sizeof(unsigned long long) * 8 * (size_t)n));
    FTK_HIP(ctx, hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * 8 * (size_t)n, ctx->stream));
    p.stamps = d_stamps;
    FTK_HIP(ctx, ftk::klt_launch(model, opt->method, p, ctx->stream));
    {
        std::vector<unsigned long long> h(8 * (size_t)n);
        FTK_HIP(ctx, hipMemcpyAsync(h.data(), d_stamps, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double avg[8] = {0};
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 8; ++k) avg[k] += (double)h[(size_t)i * 8 + k];
        static int printed = 0;
        if (printed++ < 3) {
            fprintf(stderr, "[ftk stamps] memtime ticks/feature: ref_stage %.0f setup %.0f cur_stage %.0f phaseA %.0f count %.0f chain %.0f solve %.0f total %.0f\n",
                    avg[0] / n, avg[1] / n, avg[2] / n, avg[3] / n, avg[4] / n, avg[5] / n, avg[6] / n, avg[7] / n);
        }
        if (const char *dump = FTK_ENV(ctx, stamps_dump)) {
            if (FILE *f = fopen(dump, "wb")) {
                fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
                fclose(f);
            }
        }
        if (p.pb_enabled && printed <= 3) {
            // pipelined kernel: slots 6 / 4 hold s_memrealtime (100 MHz) at workgroup start / end
            unsigned long long t_min = ~0ull, t_max = 0, life = 0;
            for (int i = 0; i < n; ++i) {
                const unsigned long long t0 = h[(size_t)i * 8 + 6], t1 = h[(size_t)i * 8 + 4];
                if (t0 == 0) continue;
                t_min = t0 < t_min ? t0 : t_min;
                t_max = t1 > t_max ? t1 : t_max;
                life += t1 - t0;
            }
            unsigned long long last_start = 0, first_end = ~0ull;
            for (int i = 0; i < n; ++i) {
                const unsigned long long t0 = h[(size_t)i * 8 + 6], t1 = h[(size_t)i * 8 + 4];
                if (t0 == 0) continue;
                last_start = t0 > last_start ? t0 : last_start;
                first_end = t1 < first_end ? t1 : first_end;
            }
            fprintf(stderr, "[ftk stamps] realtime: span %.2f us, mean workgroup life %.2f us, last start +%.2f us, first end +%.2f us\n",
                    (double)(t_max - t_min) * 0.01, (double)life * 0.01 / n, (double)(last_start - t_min) * 0.01, (double)(first_end - t_min) * 0.01);
        }
        (void)hipFree(d_stamps);
    }
    return FTK_OK;
#else
    const hipError_t launch_rc = ftk::klt_launch(model, opt->method, p, ctx->stream);
    if (launch_rc != hipSuccess) {
        // The launch-order state advanced above assumed this launch would write its iteration counts and (from the second call
        // on) a permutation: it did neither, so the history starts over — the next call must not install an order nobody wrote.
        ctx->sched_calls = 0;
        ctx->sched_n = 0;
        return fail(ctx, launch_rc == hipErrorOutOfMemory ? FTK_E_OUT_OF_MEMORY : FTK_E_HIP, "klt launch failed: %s", hipGetErrorString(launch_rc));
    }
    return FTK_OK;
#endif
}

int ftk_klt_track(ftk_context *ctx, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur, const float *ref_uv,
                  float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance, int single_level, uint32_t *iters) {
    FTK_TRACE_SCOPE("ftk_klt_track");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "klt_track: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track: negative feature count");
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!ref_uv || !cur_uv || !status) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    // One contiguous block [ref_uv | cur_uv | status | iters], mirrored in pinned host memory:
    // a single H2D of (ref_uv, cur_uv, status) and a single D2H of (cur_uv, status, iters) per call.
    const size_t uv_bytes = align_up(sizeof(float) * 2 * (size_t)n, 256);
    const size_t st_bytes = align_up((size_t)n, 256);
    const size_t it_bytes = align_up(sizeof(uint32_t) * (size_t)n, 256);
    const size_t total = 2 * uv_bytes + st_bytes + it_bytes;
    int rc = ensure_scratch(ctx, total);
    if (rc != FTK_OK) {
        return rc;
    }
    rc = ensure_pinned(ctx, total);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *dbase = static_cast<uint8_t *>(ctx->scratch);
    uint8_t *hbase = static_cast<uint8_t *>(ctx->pinned);
    float *d_ref = reinterpret_cast<float *>(dbase);
    float *d_cur = reinterpret_cast<float *>(dbase + uv_bytes);
    uint8_t *d_st = dbase + 2 * uv_bytes;
    uint32_t *d_it = reinterpret_cast<uint32_t *>(dbase + 2 * uv_bytes + st_bytes);
    memcpy(hbase, ref_uv, sizeof(float) * 2 * (size_t)n);
    memcpy(hbase + uv_bytes, cur_uv, sizeof(float) * 2 * (size_t)n);
    memcpy(hbase + 2 * uv_bytes, status, (size_t)n);
    // Small calls (the reference's callers track a few hundred features) are dominated by the two staging copies and
    // their queue latency, not by bytes: the kernel then reads (ref_uv, cur_uv, status) from and writes its 9 B per
    // feature straight into the pinned host block over PCIe — no H2D / D2H at all (2 000 features: 89 -> ~60 us per
    // call).  Larger calls keep the bulk copies.  FTK_KLT_ZEROCOPY=0 disables it.
    const bool zero_copy_allowed = !(FTK_ENV(ctx, klt_zerocopy) && atoi(FTK_ENV(ctx, klt_zerocopy)) == 0);
    void *mapped = nullptr;
    if (zero_copy_allowed && n <= 16384 && hipHostGetDevicePointer(&mapped, ctx->pinned, 0) == hipSuccess && mapped != nullptr) {
        uint8_t *mbase = static_cast<uint8_t *>(mapped);
        float *m_ref = reinterpret_cast<float *>(mbase);
        float *m_cur = reinterpret_cast<float *>(mbase + uv_bytes);
        uint8_t *m_st = mbase + 2 * uv_bytes;
        uint32_t *m_it = reinterpret_cast<uint32_t *>(mbase + 2 * uv_bytes + st_bytes);
        rc = ftk_klt_track_device(ctx, model, opt, ref, cur, m_ref, m_cur, m_cur, m_st, m_st, n, prior, consider_luminance, single_level,
                                  iters ? m_it : nullptr);
        if (rc != FTK_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            return rc;
        }
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(cur_uv, hbase + uv_bytes, sizeof(float) * 2 * (size_t)n);
        memcpy(status, hbase + 2 * uv_bytes, (size_t)n);
        if (iters) {
            memcpy(iters, hbase + 2 * uv_bytes + st_bytes, sizeof(uint32_t) * (size_t)n);
        }
        return FTK_OK;
    }
    FTK_HIP(ctx, hipMemcpyAsync(dbase, hbase, 2 * uv_bytes + st_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = ftk_klt_track_device(ctx, model, opt, ref, cur, d_ref, d_cur, d_cur, d_st, d_st, n, prior, consider_luminance, single_level,
                              iters ? d_it : nullptr);
    if (rc != FTK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    const size_t back = uv_bytes + st_bytes + (iters ? it_bytes : 0);
    FTK_HIP(ctx, hipMemcpyAsync(hbase + uv_bytes, dbase + uv_bytes, back, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(cur_uv, hbase + uv_bytes, sizeof(float) * 2 * (size_t)n);
    memcpy(status, hbase + 2 * uv_bytes, (size_t)n);
    if (iters) {
        memcpy(iters, hbase + 2 * uv_bytes + st_bytes, sizeof(uint32_t) * (size_t)n);
    }
    return FTK_OK;
}

int ftk_extract_extend_patch(ftk_context *ctx, const ftk_pyramid *ref, int32_t level, float u, float v, int32_t ex_rows, int32_t ex_cols,
                             float *ex_patch, uint8_t *valid, uint32_t *valid_count) {
    FTK_TRACE_SCOPE("ftk_extract_extend_patch");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "extract_extend_patch: null context");
    }
    FTK_LOCK(ctx);
    if (!ref || level < 0 || level >= ref->n_levels || ex_rows <= 0 || ex_cols <= 0 || !ex_patch || !valid || !valid_count) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "extract_extend_patch: bad arguments");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ex_rows * ex_cols;
    const size_t patch_bytes = align_up(sizeof(float) * n, 256);
    const size_t valid_bytes = align_up(n, 256);
    int rc = ensure_scratch(ctx, patch_bytes + valid_bytes + 256);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch);
    float *d_patch = reinterpret_cast<float *>(base);
    uint8_t *d_valid = base + patch_bytes;
    uint32_t *d_count = reinterpret_cast<uint32_t *>(base + patch_bytes + valid_bytes);
    FTK_HIP(ctx, ftk::extract_patch_launch(ref->levels[level], u, v, ex_rows, ex_cols, d_patch, d_valid, d_count, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(ex_patch, d_patch, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(valid, d_valid, n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(valid_count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

/* ---- BRIEF descriptors (producer of the matcher's input) ----------------------------------- */

static int ensure_brief_pattern(ftk_context *ctx, int32_t n_bits, int32_t half) {
    if (ctx->brief_pattern && ctx->brief_bits == n_bits && ctx->brief_half == half) {
        return FTK_OK;
    }
    if (ctx->brief_pattern) {
        FTK_HIP(ctx, hipFree(ctx->brief_pattern));
        ctx->brief_pattern = nullptr;
    }
    // LCG pattern: x <- 1664525 x + 1013904223 (seed 0x2545F491), offset = ((x >> 8) mod (2 half + 1)) - half
    std::vector<int8_t> pattern((size_t)4 * n_bits);
    uint32_t state = 0x2545F491u;
    const uint32_t span = (uint32_t)(2 * half + 1);
    for (auto &v : pattern) {
        state = state * 1664525u + 1013904223u;
        v = (int8_t)((int32_t)((state >> 8) % span) - half);
    }
    FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->brief_pattern), pattern.size()));
    // through the pinned block on the context's stream: a pageable hipMemcpy on the null stream costs milliseconds the first time
    const int prc = ensure_pinned(ctx, pattern.size());
    if (prc != FTK_OK) {
        return prc;
    }
    memcpy(ctx->pinned, pattern.data(), pattern.size());
    FTK_HIP(ctx, hipMemcpyAsync(ctx->brief_pattern, ctx->pinned, pattern.size(), hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pinned block is reused by the caller right away
    ctx->brief_bits = n_bits;
    ctx->brief_half = half;
    return FTK_OK;
}

int ftk_brief_compute_device(ftk_context *ctx, const ftk_pyramid *image, int32_t level, const float *d_uv, int32_t n, int32_t n_bits,
                             int32_t half_patch, uint32_t *d_words) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "brief_compute_device: null context");
    }
    FTK_LOCK(ctx);
    if (!image || level < 0 || level >= image->n_levels || n < 0 || n_bits <= 0 || half_patch <= 0 || half_patch > 63) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "brief_compute_device: bad arguments (n %d, bits %d, half %d)", n, n_bits, half_patch);
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!d_uv || !d_words) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "brief_compute_device: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = ensure_brief_pattern(ctx, n_bits, half_patch);
    if (rc != FTK_OK) {
        return rc;
    }
    ftk::BriefParams p;
    p.img = image->levels[level];
    p.uv = d_uv;
    p.words = d_words;
    p.pattern = ctx->brief_pattern;
    p.n = n;
    p.n_bits = n_bits;
    p.n_words = (n_bits + 31) / 32;
    p.half = half_patch;
    FTK_HIP(ctx, ftk::brief_launch(p, ctx->stream));
    return FTK_OK;
}

int ftk_brief_compute(ftk_context *ctx, const ftk_pyramid *image, int32_t level, const float *uv, int32_t n, int32_t n_bits,
                      int32_t half_patch, uint32_t *words) {
    FTK_TRACE_SCOPE("ftk_brief_compute");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "brief_compute: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0 || n_bits <= 0) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "brief_compute: bad sizes");
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!uv || !words) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "brief_compute: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n_words = (size_t)(n_bits + 31) / 32;
    const size_t uv_bytes = align_up(sizeof(float) * 2 * (size_t)n, 256);
    const size_t w_bytes = align_up(sizeof(uint32_t) * n_words * (size_t)n, 256);
    int rc = ensure_scratch(ctx, uv_bytes + w_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    if (n_bits <= 0 || half_patch <= 0 || half_patch > 63) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "brief_compute: bad arguments (bits %d, half %d)", n_bits, half_patch);
    }
    rc = ensure_brief_pattern(ctx, n_bits, half_patch);  // before the pinned block is filled: it stages the pattern there
    if (rc == FTK_OK) {
        rc = ensure_pinned(ctx, uv_bytes + w_bytes);
    }
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch), *hbase = static_cast<uint8_t *>(ctx->pinned);
    float *d_uv = reinterpret_cast<float *>(base);
    uint32_t *d_words = reinterpret_cast<uint32_t *>(base + uv_bytes);
    memcpy(hbase, uv, sizeof(float) * 2 * (size_t)n);
    FTK_HIP(ctx, hipMemcpyAsync(d_uv, hbase, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    rc = ftk_brief_compute_device(ctx, image, level, d_uv, n, n_bits, half_patch, d_words);
    if (rc != FTK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    FTK_HIP(ctx, hipMemcpyAsync(hbase + uv_bytes, d_words, sizeof(uint32_t) * n_words * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(words, hbase + uv_bytes, sizeof(uint32_t) * n_words * (size_t)n);
    return FTK_OK;
}

/* ---- Harris corners (producer of the trackers' input) -------------------------------------- */

static int harris_run(ftk_context *ctx, const ftk_pyramid *image, int32_t level, int32_t min_distance, float min_response, float *response_out,
                      std::vector<unsigned long long> *survivors) {
    if (!image || level < 0 || level >= image->n_levels) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "harris: bad image / level");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const DevImage img = image->levels[level];
    const size_t px = (size_t)img.rows * img.cols;
    const size_t capacity = px;  // worst case (min_distance 1): every candidate is its own window maximum
    const size_t g_bytes = align_up(sizeof(short) * px, 256), f_bytes = align_up(sizeof(float) * px, 256);
    const size_t k_bytes = align_up(sizeof(unsigned long long) * px, 256), l_bytes = align_up(sizeof(unsigned long long) * capacity, 256);
    const int rc = ensure_scratch(ctx, 2 * g_bytes + f_bytes + 3 * k_bytes + l_bytes + 256);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch);
    ftk::HarrisParams p;
    p.img = img;
    p.gx = reinterpret_cast<short *>(base);
    p.gy = reinterpret_cast<short *>(base + g_bytes);
    p.response = response_out ? reinterpret_cast<float *>(base + 2 * g_bytes) : nullptr;
    p.key = reinterpret_cast<unsigned long long *>(base + 2 * g_bytes + f_bytes);
    p.tmp = p.key + k_bytes / sizeof(unsigned long long);
    p.wmax = p.tmp + k_bytes / sizeof(unsigned long long);
    p.list = survivors ? p.wmax + k_bytes / sizeof(unsigned long long) : nullptr;
    p.count = reinterpret_cast<unsigned *>(base + 2 * g_bytes + f_bytes + 3 * k_bytes + l_bytes);
    p.capacity = (unsigned)capacity;
    p.min_distance = min_distance;
    p.min_response = min_response;
    FTK_HIP(ctx, ftk::harris_launch(p, ctx->stream));
    if (response_out) {
        FTK_HIP(ctx, hipMemcpyAsync(response_out, p.response, sizeof(float) * px, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (survivors) {
        unsigned count = 0;
        FTK_HIP(ctx, hipMemcpyAsync(&count, p.count, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (count > p.capacity) {
            return fail(ctx, FTK_E_UNSUPPORTED, "harris: %u survivors exceed the list capacity %u", count, p.capacity);
        }
        survivors->resize(count);
        if (count > 0) {
            FTK_HIP(ctx, hipMemcpyAsync(survivors->data(), p.list, sizeof(unsigned long long) * count, hipMemcpyDeviceToHost, ctx->stream));
            FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    } else {
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FTK_OK;
}

int ftk_harris_response(ftk_context *ctx, const ftk_pyramid *image, int32_t level, float *response) {
    FTK_TRACE_SCOPE("ftk_harris_response");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "harris_response: null context");
    }
    FTK_LOCK(ctx);
    if (!response) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "harris_response: null buffer");
    }
    return harris_run(ctx, image, level, 1, 0.0f, response, nullptr);
}

int ftk_harris_detect(ftk_context *ctx, const ftk_pyramid *image, int32_t level, int32_t max_count, int32_t min_distance, float min_response,
                      float *uv, int32_t *n_out) {
    FTK_TRACE_SCOPE("ftk_harris_detect");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "harris_detect: null context");
    }
    FTK_LOCK(ctx);
    if (!n_out || max_count < 0 || (max_count > 0 && !uv)) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "harris_detect: bad output arguments");
    }
    *n_out = 0;
    if (!image || level < 0 || level >= image->n_levels) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "harris_detect: bad image / level");
    }
    const DevImage img = image->levels[level];
    if (max_count == 0 || img.rows < 23 || img.cols < 23) {
        return FTK_OK;
    }
    std::vector<unsigned long long> survivors;
    const int rc = harris_run(ctx, image, level, min_distance, min_response, nullptr, &survivors);
    if (rc != FTK_OK) {
        return rc;
    }
    // key order == (response descending, pixel index ascending): the final top-N selection is a sort of
    // a few thousand 64-bit keys on the host
    std::sort(survivors.begin(), survivors.end(), [](unsigned long long a, unsigned long long b) { return a > b; });
    const size_t n = survivors.size() < (size_t)max_count ? survivors.size() : (size_t)max_count;
    for (size_t i = 0; i < n; ++i) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(survivors[i] & 0xFFFFFFFFull);
        uv[2 * i] = (float)(idx % (unsigned)img.cols);
        uv[2 * i + 1] = (float)(idx / (unsigned)img.cols);
    }
    *n_out = (int32_t)n;
    return FTK_OK;
}

/* ---- matcher ------------------------------------------------------------------------------- */

static bool supported_words(int32_t n_words) { return n_words == 1 || n_words == 2 || n_words == 4 || n_words == 8 || n_words == 16; }

int ftk_hamming_match_device(ftk_context *ctx, const uint32_t *d_ref_words, int32_t n_ref, const uint32_t *d_cur_words, int32_t n_cur,
                             int32_t n_words, int32_t n_bits, float max_distance, const float *d_pred_uv, const float *d_cur_uv,
                             int32_t max_col_distance, int32_t max_row_distance, int32_t *d_index_pairs, uint64_t *d_workspace) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "hamming_match_device: null context");
    }
    FTK_LOCK(ctx);
    if (n_ref < 0 || n_cur < 0 || n_bits < 0) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_device: negative size");
    }
    if (n_ref == 0 || n_cur == 0) {
        return FTK_OK;
    }
    if (n_words < 1) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_device: n_words %d < 1", n_words);
    }
    if (n_bits > 32 * n_words) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_device: n_bits %d exceeds %d words", n_bits, n_words);
    }
    if (!d_ref_words || !d_cur_words || !d_index_pairs || (d_pred_uv && !d_cur_uv)) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_device: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    // The register-tiled scan exists for 1, 2, 4, 8 and 16 words per descriptor.  Other widths up to 16 words are
    // zero-padded on the device into a context-owned copy (equal pad bits in both sets: same distances); wider
    // descriptors take the generic scan (matcher_kernels.hip), which reads any width.  Same indices either way.
    if (!supported_words(n_words) && n_words < 16) {
        int dev_words = 1;
        while (dev_words < n_words) {
            dev_words *= 2;
        }
        const size_t ref_bytes = align_up(sizeof(uint32_t) * (size_t)n_ref * dev_words, 256);
        const size_t cur_bytes = align_up(sizeof(uint32_t) * (size_t)n_cur * dev_words, 256);
        const int rc = ensure_device_buffer(ctx, &ctx->match_pad, &ctx->match_pad_bytes, ref_bytes + cur_bytes);
        if (rc != FTK_OK) {
            return rc;
        }
        uint32_t *pad_ref = static_cast<uint32_t *>(ctx->match_pad);
        uint32_t *pad_cur = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(ctx->match_pad) + ref_bytes);
        FTK_HIP(ctx, hipMemsetAsync(ctx->match_pad, 0, ref_bytes + cur_bytes, ctx->stream));
        FTK_HIP(ctx, hipMemcpy2DAsync(pad_ref, sizeof(uint32_t) * dev_words, d_ref_words, sizeof(uint32_t) * n_words, sizeof(uint32_t) * n_words,
                                      (size_t)n_ref, hipMemcpyDeviceToDevice, ctx->stream));
        FTK_HIP(ctx, hipMemcpy2DAsync(pad_cur, sizeof(uint32_t) * dev_words, d_cur_words, sizeof(uint32_t) * n_words, sizeof(uint32_t) * n_words,
                                      (size_t)n_cur, hipMemcpyDeviceToDevice, ctx->stream));
        d_ref_words = pad_ref;
        d_cur_words = pad_cur;
        n_words = dev_words;
    }
    int p_keys_clean = 0;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(d_workspace);
    p_keys_clean = 0;
    if (!keys) {
        p_keys_clean = 1;
        const int rc = ensure_match_keys(ctx, (size_t)n_ref);
        if (rc != FTK_OK) {
            return rc;
        }
        keys = ctx->match_keys;
    }
    ftk::MatchParams p;
    p.keys_clean = p_keys_clean;
    p.small_off = ftk_env::off(FTK_ENV(ctx, match_small)) ? 1 : 0;
    p.ref_words = d_ref_words;
    p.cur_words = d_cur_words;
    p.pred_uv = d_pred_uv;
    p.cur_uv = d_cur_uv;
    p.index_pairs = d_index_pairs;
    p.keys = keys;
    p.n_ref = n_ref;
    p.n_cur = n_cur;
    p.n_words = n_words;
    p.n_bits = n_bits;
    p.max_distance = max_distance;
    p.max_col = (float)max_col_distance;
    p.max_row = (float)max_row_distance;
    // Split the candidate range finely (a workgroup covers 512 reference descriptors — two per thread,
    // matcher_kernels.hip — and as few as 64 candidates): measured at 10 000 x 10 000, 40 / 80 / 160
    // splits take 85 / 74 / 69 us; the scan is pure VALU work and small workgroups even out the tail.
    const int row_blocks = (n_ref + ftk::kMatchRowsPerBlock - 1) / ftk::kMatchRowsPerBlock;
    int splits = (4096 + row_blocks - 1) / row_blocks;
    if (const char *env = FTK_ENV(ctx, match_wgs)) {
        splits = (atoi(env) + row_blocks - 1) / row_blocks;  // experiment: target number of workgroups
    }
    if (const char *env = FTK_ENV(ctx, match_splits)) {
        splits = atoi(env);  // experiment override
    }
    const int max_splits = (n_cur + 63) / 64;
    if (splits > max_splits) {
        splits = max_splits;
    }
    if (splits < 1) {
        splits = 1;
    }
    int per = (n_cur + splits - 1) / splits;
    if (!(FTK_ENV(ctx, match_any_per) && atoi(FTK_ENV(ctx, match_any_per)) == 1)) {
        per = (per + 63) / 64 * 64;
    }
    p.matrix_cores = 0;
    {
        // Which scan: 256- and 512-bit descriptors go to the matrix cores (matcher_kernels.hip, hamming_match_mfma_kernel:
        // faster at every size measured, 300 x 300 to 10 000 x 10 000, scripts/match_shapes.py); other widths to the popcount
        // scan with the candidates on the scalar path.  FTK_MATCH_KERNEL=mfma|scalar|lds forces one (experiment switch).
        const char *env = FTK_ENV(ctx, match_kernel);
        p.lds_tiles = (env && !strcmp(env, "lds")) ? 1 : 0;
        // (the matrix-core scan addresses the candidates with 32-bit byte offsets)
        bool mfma = n_bits > 0 && (n_words == 8 || n_words == 16) && (long long)n_cur * n_words * 4 < (1ll << 31);
        if (env) {
            mfma = mfma && !strcmp(env, "mfma");
        }
        if (mfma) {
            p.matrix_cores = 1;
            // One wave per workgroup: 64 rows and one split of the candidates, in 32-candidate tiles.  Two waves fit a SIMD
            // (registers): one round of at most 2048 waves, the splits whole tiles and as even as the tile count allows.
            const int mfma_row_blocks = (n_ref + 63) / 64;
            int target = 2048;
            if (const char *wgs = FTK_ENV(ctx, match_wgs)) {
                target = atoi(wgs);
            }
            int mfma_splits = target / mfma_row_blocks;
            mfma_splits = mfma_splits < 1 ? 1 : mfma_splits;
            const int n_tiles = (n_cur + 31) / 32;
            int tiles_per_split = (n_tiles + mfma_splits - 1) / mfma_splits;
            if (tiles_per_split > 1024) {
                tiles_per_split = 1024;  // the position field of the running keys
            }
            per = tiles_per_split * 32;
        }
    }
    p.cur_per_block = per;
    // NearbyMatch from a few thousand candidates on: bounding boxes for the early exit of workgroups whose candidates
    // cannot reach any window of their rows (matcher_kernels.hip)
    p.boxes = nullptr;
    const bool boxes_off = FTK_ENV(ctx, match_boxes) && atoi(FTK_ENV(ctx, match_boxes)) == 0;  // experiment switch
    if (d_pred_uv && n_bits > 0 && n_cur >= 2048 && !boxes_off) {
        const size_t n_boxes = (size_t)row_blocks + (size_t)((n_cur + per - 1) / per);
        const int rc = ensure_match_boxes(ctx, n_boxes);
        if (rc != FTK_OK) {
            return rc;
        }
        p.boxes = reinterpret_cast<float4 *>(ctx->match_boxes);
    }
#ifdef FTK_MATCH_STAMPS
    {
        // diagnostic build: per-workgroup {start, loaded, end} (s_memrealtime, 100 MHz) + HW_ID, dumped to $FTK_MATCH_STAMPS_DUMP
        const int n_splits = (n_cur + per - 1) / per;
        // (the matrix-core scan has 64-row workgroups and writes 8 words per workgroup)
        const size_t n_wg = p.matrix_cores ? (size_t)((n_ref + 63) / 64) * n_splits * 2 : (size_t)row_blocks * n_splits;
        unsigned long long *d_st = nullptr;
        FTK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d_st), sizeof(unsigned long long) * 4 * n_wg));
        FTK_HIP(ctx, hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * 4 * n_wg, ctx->stream));
        p.stamps = d_st;
        FTK_HIP(ctx, ftk::match_launch(p, ctx->stream));
        std::vector<unsigned long long> h(4 * n_wg);
        FTK_HIP(ctx, hipMemcpyAsync(h.data(), d_st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (const char *dump = FTK_ENV(ctx, match_stamps_dump)) {
            if (FILE *f = fopen(dump, "wb")) {
                fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
                fclose(f);
            }
        }
        (void)hipFree(d_st);
        return FTK_OK;
    }
#else
    p.stamps = nullptr;
    FTK_HIP(ctx, ftk::match_launch(p, ctx->stream));
    return FTK_OK;
#endif
}

int ftk_hamming_match(ftk_context *ctx, const uint32_t *ref_words, int32_t n_ref, const uint32_t *cur_words, int32_t n_cur, int32_t n_words,
                      int32_t n_bits, float max_distance, const float *pred_uv, const float *cur_uv, int32_t max_col_distance,
                      int32_t max_row_distance, int32_t *index_pairs, int *matched_ok) {
    FTK_TRACE_SCOPE("ftk_hamming_match");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "hamming_match: null context");
    }
    FTK_LOCK(ctx);
    if (matched_ok) {
        *matched_ok = 0;
    }
    if (n_ref < 0 || n_cur < 0 || n_words < 1 || n_bits < 0 || n_bits > 32 * n_words) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match: bad sizes (n_ref %d, n_cur %d, n_words %d, n_bits %d)", n_ref, n_cur, n_words, n_bits);
    }
    if (n_cur == 0) {
        return FTK_OK;  // descriptor_matcher.h:58 — `return false`
    }
    if (matched_ok) {
        *matched_ok = 1;
    }
    if (n_ref == 0) {
        return FTK_OK;
    }
    if (!ref_words || !cur_words || !index_pairs || (pred_uv && !cur_uv)) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    int dev_words = 1;
    while (dev_words < n_words && dev_words < 16) {
        dev_words *= 2;
    }
    if (n_words > 16) {
        dev_words = n_words;  // wider than any register-tiled instantiation: the generic scan reads the width as it is
    }
    const size_t ref_bytes = align_up(sizeof(uint32_t) * (size_t)n_ref * dev_words, 256);
    const size_t cur_bytes = align_up(sizeof(uint32_t) * (size_t)n_cur * dev_words, 256);
    const size_t pred_bytes = pred_uv ? align_up(sizeof(float) * 2 * (size_t)n_ref, 256) : 0;
    const size_t cuv_bytes = pred_uv ? align_up(sizeof(float) * 2 * (size_t)n_cur, 256) : 0;
    const size_t idx_bytes = align_up(sizeof(int32_t) * (size_t)n_ref, 256);
    int rc = ensure_scratch(ctx, ref_bytes + cur_bytes + pred_bytes + cuv_bytes + idx_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch);
    uint32_t *d_ref = reinterpret_cast<uint32_t *>(base);
    uint32_t *d_cur = reinterpret_cast<uint32_t *>(base + ref_bytes);
    float *d_pred = pred_uv ? reinterpret_cast<float *>(base + ref_bytes + cur_bytes) : nullptr;
    float *d_cuv = pred_uv ? reinterpret_cast<float *>(base + ref_bytes + cur_bytes + pred_bytes) : nullptr;
    int32_t *d_idx = reinterpret_cast<int32_t *>(base + ref_bytes + cur_bytes + pred_bytes + cuv_bytes);
    // One H2D per call: the inputs are gathered in the context's pinned block, laid out like the device scratch (pageable
    // hipMemcpyAsync calls are staged one by one by the runtime, ~10 us each; the reference's callers time this call).
    const size_t in_bytes = ref_bytes + cur_bytes + pred_bytes + cuv_bytes + idx_bytes;
    rc = ensure_pinned(ctx, in_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *hbase = static_cast<uint8_t *>(ctx->pinned);
    if (dev_words == n_words) {
        memcpy(hbase, ref_words, sizeof(uint32_t) * (size_t)n_ref * n_words);
        memcpy(hbase + ref_bytes, cur_words, sizeof(uint32_t) * (size_t)n_cur * n_words);
    } else {
        // zero-pad each descriptor to the next supported width (pad bits are equal in both sets -> distance unchanged)
        memset(hbase, 0, ref_bytes + cur_bytes);
        for (int32_t i = 0; i < n_ref; ++i) {
            memcpy(hbase + sizeof(uint32_t) * (size_t)i * dev_words, ref_words + (size_t)i * n_words, sizeof(uint32_t) * n_words);
        }
        for (int32_t i = 0; i < n_cur; ++i) {
            memcpy(hbase + ref_bytes + sizeof(uint32_t) * (size_t)i * dev_words, cur_words + (size_t)i * n_words, sizeof(uint32_t) * n_words);
        }
    }
    if (pred_uv) {
        memcpy(hbase + ref_bytes + cur_bytes, pred_uv, sizeof(float) * 2 * (size_t)n_ref);
        memcpy(hbase + ref_bytes + cur_bytes + pred_bytes, cur_uv, sizeof(float) * 2 * (size_t)n_cur);
    }
    uint8_t *h_idx = hbase + ref_bytes + cur_bytes + pred_bytes + cuv_bytes;
    memcpy(h_idx, index_pairs, sizeof(int32_t) * (size_t)n_ref);
    FTK_HIP(ctx, hipMemcpyAsync(base, hbase, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = ftk_hamming_match_device(ctx, d_ref, n_ref, d_cur, n_cur, dev_words, n_bits, max_distance, d_pred, d_cuv, max_col_distance,
                                  max_row_distance, d_idx, nullptr);
    if (rc != FTK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    FTK_HIP(ctx, hipMemcpyAsync(h_idx, d_idx, sizeof(int32_t) * (size_t)n_ref, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(index_pairs, h_idx, sizeof(int32_t) * (size_t)n_ref);
    return FTK_OK;
}

/* ---- diagnostics ---------------------------------------------------------------------------- */

int ftk_ldlt6_solve(ftk_context *ctx, const float *a, const float *b, float *x, int32_t n) {
    FTK_TRACE_SCOPE("ftk_ldlt6_solve");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "ldlt6_solve: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0 || (n > 0 && (!a || !b || !x))) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "ldlt6_solve: bad arguments");
    }
    if (n == 0) {
        return FTK_OK;
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t a_bytes = align_up(sizeof(float) * 36 * (size_t)n, 256), b_bytes = align_up(sizeof(float) * 6 * (size_t)n, 256);
    const int rc = ensure_scratch(ctx, a_bytes + 2 * b_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch);
    float *d_a = reinterpret_cast<float *>(base), *d_b = reinterpret_cast<float *>(base + a_bytes), *d_x = reinterpret_cast<float *>(base + a_bytes + b_bytes);
    FTK_HIP(ctx, hipMemcpyAsync(d_a, a, sizeof(float) * 36 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_b, b, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, ftk::ldlt6_launch(d_a, d_b, d_x, n, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(x, d_x, sizeof(float) * 6 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

/* ---- direct method ------------------------------------------------------------------------ */

void ftk_default_direct_options(ftk_direct_options *opt) {
    if (!opt) {
        return;
    }
    opt->max_track_points = 500;
    opt->max_iteration = 15;
    opt->half_rows = 6;
    opt->half_cols = 6;
    opt->max_converge_step = 1e-6f;
    opt->max_converge_residual = 2.0f;
    opt->method = FTK_METHOD_DIRECT;
}

int ftk_direct_track_batch_device(ftk_context *ctx, const ftk_direct_options *opt, const ftk_direct_problem *problems, int32_t n_problems) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "direct_track: null context");
    }
    FTK_LOCK(ctx);
    if (!opt || n_problems < 0 || (n_problems > 0 && !problems)) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "direct_track: null options / problems");
    }
    if (n_problems == 0) {
        return FTK_OK;
    }
    if (opt->half_rows < 0 || opt->half_cols < 0 || opt->half_rows > 63 || opt->half_cols > 63) {
        return fail(ctx, FTK_E_UNSUPPORTED, "direct_track: half patch size (%d, %d) outside [0, 63]", opt->half_rows, opt->half_cols);
    }
    std::vector<ftk::DirectProblem> host((size_t)n_problems);
    uint32_t max_features = 0;
    int32_t n_levels = 0;
    for (int32_t k = 0; k < n_problems; ++k) {
        const ftk_direct_problem &in = problems[k];
        if (!in.ref || !in.cur || in.n < 0 || (in.n > 0 && (!in.d_p_c_in_ref || !in.d_ref_uv || !in.d_cur_uv || !in.d_status)) || !in.d_pose) {
            return fail(ctx, FTK_E_INVALID_ARGUMENT, "direct_track: problem %d has a null buffer", k);
        }
        if (in.ref->n_levels != in.cur->n_levels || in.ref->n_levels < 1) {
            return fail(ctx, FTK_E_INVALID_ARGUMENT, "direct_track: problem %d pyramid level mismatch (%d vs %d)", k, in.ref->n_levels, in.cur->n_levels);
        }
        if (k == 0) {
            n_levels = in.ref->n_levels;
        } else if (in.ref->n_levels != n_levels) {
            return fail(ctx, FTK_E_UNSUPPORTED, "direct_track: all problems of a batch must share the pyramid depth");
        }
        if (in.ref->device != ctx->device || in.cur->device != ctx->device) {
            return fail(ctx, FTK_E_INVALID_ARGUMENT, "direct_track: pyramid lives on another device");
        }
        ftk::DirectProblem &out = host[(size_t)k];
        memset(&out, 0, sizeof(out));
        for (int i = 0; i < n_levels; ++i) {
            out.ref[i] = in.ref->levels[i];
            out.cur[i] = in.cur->levels[i];
        }
        for (int i = 0; i < 4; ++i) {
            out.K[i] = in.K[i];
        }
        out.p_ref = in.d_p_c_in_ref;
        out.ref_uv = in.d_ref_uv;
        out.cur_uv = in.d_cur_uv;
        out.pose = in.d_pose;
        out.status = in.d_status;
        out.iterations = in.d_iterations;
        out.n = in.n;
        out.status_valid = in.status_valid ? 1 : 0;
        const uint32_t tracked = ((uint32_t)in.n < opt->max_track_points) ? (uint32_t)in.n : opt->max_track_points;
        max_features = std::max(max_features, tracked);
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    // The per-feature table (an iteration's projections, a level's reference positions and Jacobians) lives in LDS while it fits
    // beside the product ring (64 B per tracked feature, up to kDirectLdsFeatures); larger problems keep that table in a context-owned device buffer instead —
    // same kernel, same arithmetic, same order of the sums.
    const bool feat_in_global = max_features > ftk::kDirectLdsFeatures;
    if (feat_in_global) {
        const size_t per = align_up(sizeof(float) * 16 * (size_t)max_features, 256);
        const int rc = ensure_device_buffer(ctx, &ctx->direct_feat, &ctx->direct_feat_bytes, per * (size_t)n_problems);
        if (rc != FTK_OK) {
            return rc;
        }
        for (int32_t k = 0; k < n_problems; ++k) {
            host[(size_t)k].feat = reinterpret_cast<float4 *>(static_cast<uint8_t *>(ctx->direct_feat) + per * (size_t)k);
        }
    }
    // the problem table travels through a context-owned device buffer (separate from the scratch the host-buffer wrapper uses)
    const size_t table_bytes = sizeof(ftk::DirectProblem) * (size_t)n_problems;
    if (table_bytes > ctx->direct_table_bytes) {
        if (ctx->direct_table) {
            FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            FTK_HIP(ctx, hipFree(ctx->direct_table));
            ctx->direct_table = nullptr;
            ctx->direct_table_bytes = 0;
        }
        FTK_HIP(ctx, hipMalloc(&ctx->direct_table, align_up(table_bytes, 4096)));
        ctx->direct_table_bytes = align_up(table_bytes, 4096);
    }
    // pageable host -> device copy: synchronous with respect to the host buffer, so `host` may go out of scope
    FTK_HIP(ctx, hipMemcpyAsync(ctx->direct_table, host.data(), table_bytes, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ftk::DirectParams p;
    p.problems = static_cast<const ftk::DirectProblem *>(ctx->direct_table);
    p.tree = ctx->reduction == FTK_REDUCTION_TREE ? 1 : 0;
    p.n_levels = n_levels;
    p.max_track_points = opt->max_track_points;
    p.max_iteration = opt->max_iteration;
    p.half_rows = opt->half_rows;
    p.half_cols = opt->half_cols;
    p.patch_rows = 2 * opt->half_rows + 1;
    p.patch_cols = 2 * opt->half_cols + 1;
    p.converge = opt->max_converge_step;
    p.method = opt->method;
    // ONE problem (or a handful: a stereo pair, a small rig) with enough terms: spread over the chip (direct_track_spread_kernel) — the
    // one-workgroup kernel is bound by what a single compute unit can issue per iteration.  Exact sums only; FTK_DIRECT_SPREAD=0 keeps
    // the one-workgroup kernel, =n sets the number of producer workgroups per problem (default 32).  Larger batches fill the chip with
    // one workgroup per problem.
    p.spread = 0;
    p.spread_ws = nullptr;
    p.spread_ws_words = 0;
    p.spread_poison = ftk_env::on(FTK_ENV(ctx, direct_spread_poison)) ? 1 : 0;
    ctx->direct_spread_launched = 0;
    {
        const char *env = FTK_ENV(ctx, direct_spread);
        int producers = env ? atoi(env) : 32;
        producers = producers < 0 ? 0 : (producers > 200 ? 200 : producers);
        const long long terms = (long long)max_features * p.patch_rows * p.patch_cols;
        long long min_terms = 64ll * 256;  // below about 256 chunks the producers of one compute unit keep up with the chain
        if (const char *min_env = FTK_ENV(ctx, direct_spread_min_terms)) {
            min_terms = atoll(min_env);  // tests: spread even tiny problems (producers whose waves own no chunk)
        }
        int max_problems = kDirectSpreadMaxProblems;
        if (const char *max_env = FTK_ENV(ctx, direct_spread_max_problems)) {
            max_problems = atoi(max_env);  // experiments: scripts/direct_batch_time.py
        }
        bool spread = producers > 0 && !ctx->direct_spread_off && n_problems <= max_problems && !p.tree && opt->method == FTK_METHOD_DIRECT && !feat_in_global &&
                      max_features > 0 && terms >= min_terms && terms < (1ll << 31);
        if (spread) {
            // Every workgroup of the launch must be resident at once (consumer and producers wait for each other): size the producers from
            // what THIS device holds — occupancy of the kernel as launched x its compute units (256 on a whole MI355X, 32 on a CPX partition),
            // an eighth left free for whatever else runs — and keep the one-workgroup kernel when fewer than 1 + 2 fit per problem.
            if (ctx->direct_spread_resident < 0 || ctx->direct_spread_resident_features != max_features) {
                ctx->direct_spread_resident = ftk::direct_spread_resident_groups(max_features, ctx->device);
                ctx->direct_spread_resident_features = max_features;
            }
            int resident = ctx->direct_spread_resident;
            if (const char *cap_env = FTK_ENV(ctx, direct_spread_resident)) {
                resident = std::min(resident, atoi(cap_env));  // tests: pretend to be a small partition
            }
            const int usable = resident - resident / 8;
            const int fit = usable / n_problems - 1;
            producers = std::min(producers, fit);
            spread = producers >= kDirectSpreadMinProducers || (env && producers >= 1 && producers == std::min(atoi(env), fit));  // (an explicit FTK_DIRECT_SPREAD=n that fits is honoured: tests)
        }
        size_t ws = 0;
        if (spread) {
            // Workspace: [chunk][7][64] values per problem (the Jacobian row and the residual of every term).  Not beyond 512 MB in total
            // (127 x 127 patches x 768 features would be 347 MB per problem: two such problems keep the one-workgroup kernel), word
            // offsets must fit 32 bits, and a buffer that would have
            // to GROW while the stream is being captured is an error the caller can act on, not a hipMalloc inside the capture.
            ws = align_up(ftk::direct_spread_ws_bytes(max_features, p.patch_rows, p.patch_cols), 256);
            if (ws / sizeof(uint32_t) > 0xFFFFFFFFull || ws * (size_t)n_problems > (512ull << 20)) {
                spread = false;
            }
        }
        if (spread && ws * (size_t)n_problems > ctx->direct_spread_bytes) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(ctx->stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
                spread = false;  // the one-workgroup kernel needs no workspace: same result, capturable
            }
        }
        if (spread) {
            const int rc = ensure_device_buffer(ctx, &ctx->direct_spread, &ctx->direct_spread_bytes, ws * (size_t)n_problems);
            if (rc != FTK_OK) {
                return rc;
            }
            for (int32_t k = 0; k < n_problems; ++k) {  // header + chunk flags of every problem: zero before the launch
                FTK_HIP(ctx, hipMemsetAsync(static_cast<uint8_t *>(ctx->direct_spread) + ws * (size_t)k, 0,
                                            ftk::direct_spread_clear_bytes(max_features, p.patch_rows, p.patch_cols), ctx->stream));
            }
            p.spread = producers;
            p.spread_ws = static_cast<uint32_t *>(ctx->direct_spread);
            p.spread_ws_words = (uint32_t)(ws / sizeof(uint32_t));
            ctx->direct_spread_launched = n_problems;
            ctx->direct_spread_stride = ws;
        }
    }
    FTK_HIP(ctx, ftk::direct_track_launch(p, n_problems, feat_in_global ? 0u : max_features, ctx->stream));
    return FTK_OK;
}

int ftk_direct_track(ftk_context *ctx, const ftk_direct_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur, const float *K,
                     const float *p_c_in_ref, const float *ref_uv, float *cur_uv, int32_t n, float *q_rc_wxyz, float *p_rc, uint8_t *status,
                     int status_valid, uint32_t *iterations) {
    FTK_TRACE_SCOPE("ftk_direct_track");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "direct_track: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "direct_track: negative feature count");
    }
    if (n == 0) {
        return FTK_OK;  // the class returns false for an empty ref_pixel_uv (:38); nothing to compute here
    }
    if (!opt || !ref || !cur || !K || !p_c_in_ref || !ref_uv || !cur_uv || !q_rc_wxyz || !p_rc || !status) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "direct_track: null argument");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pts_bytes = align_up(sizeof(float) * 3 * (size_t)n, 256);
    const size_t uv_bytes = align_up(sizeof(float) * 2 * (size_t)n, 256);
    const size_t st_bytes = align_up((size_t)n, 256);
    const size_t total = pts_bytes + 2 * uv_bytes + st_bytes + 256 + 256;
    int rc = ensure_scratch(ctx, total);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch);
    float *d_pts = reinterpret_cast<float *>(base);
    float *d_ref = reinterpret_cast<float *>(base + pts_bytes);
    float *d_cur = reinterpret_cast<float *>(base + pts_bytes + uv_bytes);
    uint8_t *d_st = base + pts_bytes + 2 * uv_bytes;
    float *d_pose = reinterpret_cast<float *>(base + pts_bytes + 2 * uv_bytes + st_bytes);
    uint32_t *d_it = reinterpret_cast<uint32_t *>(base + pts_bytes + 2 * uv_bytes + st_bytes + 256);
    float pose[7] = {q_rc_wxyz[0], q_rc_wxyz[1], q_rc_wxyz[2], q_rc_wxyz[3], p_rc[0], p_rc[1], p_rc[2]};
    FTK_HIP(ctx, hipMemcpyAsync(d_pts, p_c_in_ref, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_ref, ref_uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_cur, cur_uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_st, status, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_pose, pose, sizeof(pose), hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `pose` is a stack buffer
    ftk_direct_problem prob;
    prob.ref = ref;
    prob.cur = cur;
    for (int i = 0; i < 4; ++i) {
        prob.K[i] = K[i];
    }
    prob.d_p_c_in_ref = d_pts;
    prob.d_ref_uv = d_ref;
    prob.d_cur_uv = d_cur;
    prob.n = n;
    prob.d_pose = d_pose;
    prob.d_status = d_st;
    prob.status_valid = status_valid;
    prob.d_iterations = d_it;
    rc = ftk_direct_track_batch_device(ctx, opt, &prob, 1);
    if (rc != FTK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    if (ctx->direct_spread_launched > 0) {
        // The spread kernel's bounded waits ran out (its 1 + NP workgroups were not co-resident: a CU mask, a partition smaller than the
        // runtime reported, long kernels of other streams): header word 1 is set and the pose is NaN.  A synchronous caller must never
        // get that with FTK_OK — run the problem again on the one-workgroup kernel (same sums, same result as a good spread launch).
        uint32_t poisoned = 0;
        FTK_HIP(ctx, hipMemcpyAsync(&poisoned, static_cast<uint32_t *>(ctx->direct_spread) + 1, sizeof(poisoned), hipMemcpyDeviceToHost, ctx->stream));
        FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (poisoned != 0) {
            FTK_HIP(ctx, hipMemcpyAsync(d_cur, cur_uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            FTK_HIP(ctx, hipMemcpyAsync(d_st, status, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            for (int i = 0; i < 4; ++i) {
                pose[i] = q_rc_wxyz[i];
            }
            for (int i = 0; i < 3; ++i) {
                pose[4 + i] = p_rc[i];
            }
            FTK_HIP(ctx, hipMemcpyAsync(d_pose, pose, sizeof(pose), hipMemcpyHostToDevice, ctx->stream));
            FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->direct_spread_off = true;
            rc = ftk_direct_track_batch_device(ctx, opt, &prob, 1);
            ctx->direct_spread_off = false;
            if (rc != FTK_OK) {
                (void)hipStreamSynchronize(ctx->stream);
                return rc;
            }
            ++ctx->direct_spread_reruns;
            // not a failure — the result below is the one-workgroup kernel's — but worth telling: ftk_last_error() carries the note
            ctx->error = "note: ftk_direct_track: the spread launch was not co-resident (its bounded waits ran out); the problem was re-run on one workgroup";
        }
    }
    uint32_t it = 0;
    FTK_HIP(ctx, hipMemcpyAsync(cur_uv, d_cur, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(status, d_st, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(pose, d_pose, sizeof(pose), hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(&it, d_it, sizeof(it), hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) {
        q_rc_wxyz[i] = pose[i];
    }
    for (int i = 0; i < 3; ++i) {
        p_rc[i] = pose[4 + i];
    }
    if (iterations) {
        *iterations = it;
    }
    return FTK_OK;
}

/* ---- float-descriptor matcher -------------------------------------------------------------- */

int ftk_cosine_match_device(ftk_context *ctx, const float *d_ref_desc, int32_t n_ref, const float *d_cur_desc, int32_t n_cur, int32_t dim,
                            float max_distance, const float *d_pred_uv, const float *d_cur_uv, int32_t max_col_distance,
                            int32_t max_row_distance, int32_t *d_index_pairs) {
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "cosine_match_device: null context");
    }
    FTK_LOCK(ctx);
    if (n_ref < 0 || n_cur < 0 || dim < 1) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "cosine_match_device: bad sizes (n_ref %d, n_cur %d, dim %d)", n_ref, n_cur, dim);
    }
    if (dim > 4096) {
        return fail(ctx, FTK_E_UNSUPPORTED, "cosine_match_device: dim %d > 4096", dim);
    }
    if (n_ref == 0 || n_cur == 0) {
        return FTK_OK;
    }
    if (!d_ref_desc || !d_cur_desc || !d_index_pairs || (d_pred_uv && !d_cur_uv)) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "cosine_match_device: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    ftk::CosineParams p;
    p.small_off = ftk_env::off(FTK_ENV(ctx, cosine_small)) ? 1 : 0;
    p.small_any = ftk_env::on(FTK_ENV(ctx, cosine_small_any)) ? 1 : 0;
    p.ref = d_ref_desc;
    p.cur = d_cur_desc;
    p.pred_uv = d_pred_uv;
    p.cur_uv = d_cur_uv;
    p.index_pairs = d_index_pairs;
    p.n_ref = n_ref;
    p.n_cur = n_cur;
    p.dim = dim;
    p.dim_pad = (int32_t)align_up((size_t)dim, 64);
    // dim <= 256 (SuperPoint, DISK): the ref-stationary contraction — 128 ref rows for the whole K resident in LDS,
    // cur streamed in 256-row tiles, one 8-wave workgroup per CU.  Longer descriptors use the chunked kernel.
    // dim <= 256 (SuperPoint, DISK): the ref fragments stay on chip for the whole walk over cur — in registers
    // (cosine_gemm_rr_kernel, 512 ref rows per workgroup; the default) or in LDS (cosine_gemm_rs_kernel, 128 rows;
    // FTK_COSINE_KERNEL=rs).  Longer descriptors, or FTK_COSINE_KERNEL=chunked, use the chunked kernel.
    const char *kernel_env = FTK_ENV(ctx, cosine_kernel);
    const bool want_chunked = (kernel_env && !strcmp(kernel_env, "chunked")) || (FTK_ENV(ctx, cosine_chunked) && atoi(FTK_ENV(ctx, cosine_chunked)) == 1);
    p.ref_stationary = (p.dim_pad <= 256 && !want_chunked) ? ((kernel_env && !strcmp(kernel_env, "rs")) ? 1 : 2) : 0;
    const int cur_tile = p.ref_stationary == 2 ? 64 : (p.ref_stationary == 1 ? 256 : 128);
    const int row_group = p.ref_stationary == 2 ? 512 : 128;
    p.n_ref_pad = (int32_t)align_up((size_t)n_ref, (size_t)row_group);
    p.n_cur_pad = (int32_t)align_up((size_t)n_cur, (size_t)cur_tile);
    p.max_distance = max_distance;
    p.max_col = (float)max_col_distance;
    p.max_row = (float)max_row_distance;
    // Keep the whole grid co-resident in ONE round (on-chip ref: one workgroup per CU -> <= 256; chunked: two per
    // CU -> <= 512), each workgroup walking a contiguous run of cur tiles: a second, partly filled round costs more
    // than slightly longer runs.
    const int row_tiles = p.n_ref_pad / row_group, tiles_total = p.n_cur_pad / cur_tile;
    int splits = (p.ref_stationary ? 256 : 512) / row_tiles;
    if (const char *env = FTK_ENV(ctx, cosine_splits)) {
        splits = atoi(env);  // experiment override
    }
    splits = std::max(1, std::min(splits, tiles_total));
    if (p.ref_stationary == 2 && !FTK_ENV(ctx, cosine_splits)) {
        // At least TWO tiles per split: a walk's first step has no running maximum to cut against yet, so it lists its whole share
        // of every row; with one-tile splits that is all there is, the rows' lists overflow (kCosineCandCap) and the recheck falls
        // back to the exact scan of every pair — 2 000 x 2 000 x 256: 32 splits 17.6 + 3 591 us (contraction + recheck), 16 splits
        // 20.2 + 11.8 us; 1 000 x 1 000 x 128: 16 splits 11.4 + 1 099 us, 8 splits 14.1 + 7.6 us (scripts/trace_cosine_shape.sh).
        splits = std::max(1, std::min(splits, tiles_total / 2));
    }
    p.splits = splits;
    p.tiles_per_split = (tiles_total + splits - 1) / splits;
    // workspace carve-up (every region 256-byte aligned)
    size_t off = 0;
    auto carve = [&off](size_t bytes) {
        const size_t at = off;
        off += align_up(bytes, 256);
        return at;
    };
    const size_t o_ref_h = carve(sizeof(uint16_t) * (size_t)p.n_ref_pad * p.dim_pad);
    const size_t o_cur_h = carve(sizeof(uint16_t) * (size_t)p.n_cur_pad * p.dim_pad);
    const size_t o_ref_norm = carve(sizeof(float) * (size_t)p.n_ref_pad);
    const size_t o_cur_norm = carve(sizeof(float) * (size_t)p.n_cur_pad);
    const size_t o_cur_bias = carve(sizeof(float) * (size_t)p.n_cur_pad);
    const size_t o_cur_info = carve(sizeof(float) * 4 * (size_t)p.n_cur_pad);
    const size_t o_tile_box = carve(sizeof(float) * 4 * ((size_t)p.n_cur_pad / 64 + 1));
    const size_t o_ref_irr = carve((size_t)p.n_ref_pad);
    // row_max | cand_count | irregular_count are adjacent: ONE memset clears them (key 0 = "no candidate yet")
    const size_t o_row_max = carve(sizeof(uint32_t) * (size_t)p.n_ref_pad);
    const size_t o_cnt = carve(sizeof(uint32_t) * (size_t)p.n_ref_pad);
    const size_t o_irr_cnt = carve(sizeof(uint32_t));
    const size_t o_clear_end = off;
    const size_t o_cand = carve(sizeof(int32_t) * (size_t)p.n_ref_pad * ftk::kCosineCandCap);
    // the on-chip-ref kernels walk cur ONCE (running row maximum + scored candidate lists); FTK_COSINE_TWO_PASS=1 with
    // FTK_COSINE_KERNEL=rs keeps the maximum-then-collect pair of launches for comparison
    const bool single_walk =
        p.ref_stationary == 2 || (p.ref_stationary == 1 && !(FTK_ENV(ctx, cosine_two_pass) && atoi(FTK_ENV(ctx, cosine_two_pass)) == 1));
    const size_t o_cand_score = single_walk ? carve(sizeof(float) * (size_t)p.n_ref_pad * ftk::kCosineCandCap) : 0;
    const size_t o_irr_list = carve(sizeof(int32_t) * ftk::kCosineIrregularCap);
    const int rc = ensure_cosine_ws(ctx, off);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *ws = static_cast<uint8_t *>(ctx->cosine_ws);
    p.ref_h = reinterpret_cast<_Float16 *>(ws + o_ref_h);
    p.cur_h = reinterpret_cast<_Float16 *>(ws + o_cur_h);
    p.ref_norm = reinterpret_cast<float *>(ws + o_ref_norm);
    p.cur_norm = reinterpret_cast<float *>(ws + o_cur_norm);
    p.cur_bias = reinterpret_cast<float *>(ws + o_cur_bias);
    p.cur_info = reinterpret_cast<float4 *>(ws + o_cur_info);
    // NearbyMatch tile lists (float_matcher_kernels.hip): worth their extra launch from a few thousand candidates on
    p.tile_box = (d_pred_uv && p.n_cur_pad / 64 >= 32) ? reinterpret_cast<float4 *>(ws + o_tile_box) : nullptr;
    p.ref_irregular = ws + o_ref_irr;
    p.row_max = reinterpret_cast<uint32_t *>(ws + o_row_max);
    p.cand_count = reinterpret_cast<uint32_t *>(ws + o_cnt);
    p.cand = reinterpret_cast<int32_t *>(ws + o_cand);
    p.cand_score = single_walk ? reinterpret_cast<float *>(ws + o_cand_score) : nullptr;
    p.irregular_count = reinterpret_cast<uint32_t *>(ws + o_irr_cnt);
    p.irregular_list = reinterpret_cast<int32_t *>(ws + o_irr_list);
    p.clear_begin = ws + o_row_max;
    p.clear_bytes = o_clear_end - o_row_max;
    FTK_HIP(ctx, ftk::cosine_match_launch(p, ctx->stream));
    return FTK_OK;
}

int ftk_cosine_match(ftk_context *ctx, const float *ref_desc, int32_t n_ref, const float *cur_desc, int32_t n_cur, int32_t dim, float max_distance,
                     const float *pred_uv, const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *index_pairs,
                     int *matched_ok) {
    FTK_TRACE_SCOPE("ftk_cosine_match");
    if (!ctx) {
        return fail(nullptr, FTK_E_INVALID_ARGUMENT, "cosine_match: null context");
    }
    FTK_LOCK(ctx);
    if (matched_ok) {
        *matched_ok = 0;
    }
    if (n_ref < 0 || n_cur < 0 || dim < 1) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "cosine_match: bad sizes (n_ref %d, n_cur %d, dim %d)", n_ref, n_cur, dim);
    }
    if (n_cur == 0) {
        return FTK_OK;  // descriptor_matcher.h:58,94 — `return false`
    }
    if (matched_ok) {
        *matched_ok = 1;
    }
    if (n_ref == 0) {
        return FTK_OK;
    }
    if (!ref_desc || !cur_desc || !index_pairs || (pred_uv && !cur_uv)) {
        return fail(ctx, FTK_E_INVALID_ARGUMENT, "cosine_match: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ref_bytes = align_up(sizeof(float) * (size_t)n_ref * dim, 256);
    const size_t cur_bytes = align_up(sizeof(float) * (size_t)n_cur * dim, 256);
    const size_t pred_bytes = pred_uv ? align_up(sizeof(float) * 2 * (size_t)n_ref, 256) : 0;
    const size_t cuv_bytes = pred_uv ? align_up(sizeof(float) * 2 * (size_t)n_cur, 256) : 0;
    const size_t idx_bytes = align_up(sizeof(int32_t) * (size_t)n_ref, 256);
    int rc = ensure_scratch(ctx, ref_bytes + cur_bytes + pred_bytes + cuv_bytes + idx_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(ctx->scratch);
    float *d_ref = reinterpret_cast<float *>(base);
    float *d_cur = reinterpret_cast<float *>(base + ref_bytes);
    float *d_pred = pred_uv ? reinterpret_cast<float *>(base + ref_bytes + cur_bytes) : nullptr;
    float *d_cuv = pred_uv ? reinterpret_cast<float *>(base + ref_bytes + cur_bytes + pred_bytes) : nullptr;
    int32_t *d_idx = reinterpret_cast<int32_t *>(base + ref_bytes + cur_bytes + pred_bytes + cuv_bytes);
    // one H2D per call through the pinned block (see ftk_hamming_match)
    const size_t in_bytes = ref_bytes + cur_bytes + pred_bytes + cuv_bytes + idx_bytes;
    rc = ensure_pinned(ctx, in_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *hbase = static_cast<uint8_t *>(ctx->pinned);
    memcpy(hbase, ref_desc, sizeof(float) * (size_t)n_ref * dim);
    memcpy(hbase + ref_bytes, cur_desc, sizeof(float) * (size_t)n_cur * dim);
    if (pred_uv) {
        memcpy(hbase + ref_bytes + cur_bytes, pred_uv, sizeof(float) * 2 * (size_t)n_ref);
        memcpy(hbase + ref_bytes + cur_bytes + pred_bytes, cur_uv, sizeof(float) * 2 * (size_t)n_cur);
    }
    uint8_t *h_idx = hbase + ref_bytes + cur_bytes + pred_bytes + cuv_bytes;
    memcpy(h_idx, index_pairs, sizeof(int32_t) * (size_t)n_ref);
    FTK_HIP(ctx, hipMemcpyAsync(base, hbase, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = ftk_cosine_match_device(ctx, d_ref, n_ref, d_cur, n_cur, dim, max_distance, d_pred, d_cuv, max_col_distance, max_row_distance, d_idx);
    if (rc != FTK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    FTK_HIP(ctx, hipMemcpyAsync(h_idx, d_idx, sizeof(int32_t) * (size_t)n_ref, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(index_pairs, h_idx, sizeof(int32_t) * (size_t)n_ref);
    return FTK_OK;
}

int ftk_fill_matched_pixels(const int32_t *index_pairs, int32_t n_ref, const float *cur_uv, int32_t n_cur, float *matched_uv, uint8_t *status) {
    if (n_ref < 0 || n_cur < 0 || (n_ref > 0 && (!index_pairs || !matched_uv || !status)) || (n_cur > 0 && !cur_uv)) {
        return FTK_E_INVALID_ARGUMENT;
    }
    for (int32_t i = 0; i < n_ref; ++i) {
        if (status[i] > FTK_TRACKED) {
            continue;
        }
        const int32_t j = index_pairs[i];
        if (j >= 0 && j < n_cur) {
            matched_uv[2 * i] = cur_uv[2 * j];
            matched_uv[2 * i + 1] = cur_uv[2 * j + 1];
            status[i] = FTK_TRACKED;
        } else {
            status[i] = FTK_LARGE_RESIDUAL;
        }
    }
    return FTK_OK;
}

}  // extern "C"
