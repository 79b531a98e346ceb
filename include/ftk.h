/*
 * ftk.h — C ABI of the MI355X-native sparse feature tracker (libftk_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of Horizon1026/Feature_Tracker: the pyramidal
 * Lucas-Kanade trackers and the BRIEF descriptor matcher.  Every entry point names the
 * reference interface it replaces (file:line relative to the reference repo).  Signatures use
 * plain pointers and sizes only; the host-side C++ classes with the reference's names
 * (feature_tracker_amd/host/) and the Python binding (feature_tracker_amd/) sit on top.
 *
 * Conventions
 *   - (u, v) pairs are contiguous {float u(=x, col), float v(=y, row)} — the memory layout of
 *     std::vector<Vec2> (optical_flow.h:38-42).
 *   - status is one uint8_t per feature with the TrackStatus codes (src/feature_tracker.h:8-14).
 *   - images are 8-bit gray, row-major, pitch == cols (GrayImage).
 *   - 2x2 matrices (prior) are row-major [m00, m01, m10, m11].
 *   - every function returns FTK_OK (0) or a negative FTK_E_* code; ftk_last_error() gives text.
 *     There is NO CPU fallback: without a usable HIP device every compute call fails.
 *   - a context is bound to one device and one HIP stream.  Calls on ONE context may come from several
 *     threads: every entry point holds the context's lock for its duration (the context-owned scratch,
 *     pinned staging and workspaces are reused by every call), so they are serialised, not concurrent —
 *     separate tracker / matcher objects sharing a context stay independent, as in the reference.
 *     For concurrency use one context per thread.  ftk_last_error() text is valid until the next call
 *     on that context.
 */
#ifndef FTK_H_
#define FTK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTK_ABI_VERSION 1
#define FTK_MAX_LEVELS 12

enum {
    FTK_OK = 0,
    FTK_E_INVALID_ARGUMENT = -1,
    FTK_E_NO_DEVICE = -2,
    FTK_E_HIP = -3,
    FTK_E_UNSUPPORTED = -4,
    FTK_E_OUT_OF_MEMORY = -5,
};

/* TrackStatus, src/feature_tracker.h:8-14 */
enum {
    FTK_NOT_TRACKED = 0,
    FTK_TRACKED = 1,
    FTK_LARGE_RESIDUAL = 2,
    FTK_OUTSIDE = 3,
    FTK_NUMERIC_ERROR = 4,
};

/* OpticalFlowMethod, src/optical_flow_tracker/optical_flow.h:12-18.  kSse / kNeon (3, 4) take
 * the reference's `default:` branch, i.e. behave as kFast (basic_klt.cpp:31-34). */
enum { FTK_METHOD_INVERSE = 0, FTK_METHOD_DIRECT = 1, FTK_METHOD_FAST = 2, FTK_METHOD_SSE = 3, FTK_METHOD_NEON = 4 };

/* which tracker class: OpticalFlowBasicKlt / OpticalFlowAffineKlt / OpticalFlowLssdKlt */
enum { FTK_MODEL_BASIC = 0, FTK_MODEL_AFFINE = 1, FTK_MODEL_LSSD = 2 };

/* GrayImage view (Slam_Utility datatype_image.h; call sites basic_klt.cpp:22-23) */
typedef struct ftk_image {
    const uint8_t *data;
    int32_t rows;
    int32_t cols;
} ftk_image;

/* OpticalFlowOptions, src/optical_flow_tracker/optical_flow.h:20-28 (same defaults) */
typedef struct ftk_klt_options {
    uint32_t max_track_points;         /* kMaxTrackPointsNumber   = 500  */
    uint32_t max_iteration;            /* kMaxIteration           = 15   */
    uint32_t max_tolerance_large_step; /* kMaxToleranceLargeStep  = 3    */
    int32_t half_rows;                 /* kPatchRowHalfSize       = 6    */
    int32_t half_cols;                 /* kPatchColHalfSize       = 6    */
    float max_converge_step;           /* kMaxConvergeStep        = 4e-2 */
    int32_t method;                    /* kMethod                 = kFast */
} ftk_klt_options;

typedef struct ftk_context ftk_context;
typedef struct ftk_pyramid ftk_pyramid;

/* ---- runtime ----------------------------------------------------------------------------- */

int ftk_abi_version(void);
/* "key=value; ..." text describing this build of the library: source_hash (over every source file of libftk_hip.so; bench.py
 * compares it with the hash recorded next to committed counter figures), compiler (hipcc's version line), arch, mllvm (the
 * internal LLVM options the toolchain accepted when the library was built — each is probed, csrc/Makefile) and mllvm_rejected. */
const char *ftk_build_info(void);
/* Number of visible HIP devices (0 when none / no driver). Does not create a HIP context. */
int ftk_device_count(void);
/* stream: a hipStream_t to launch on (borrowed, e.g. a torch stream), or NULL to let the context
 * create and own one.  device < 0 keeps the calling thread's current device. */
int ftk_context_create(int device, void *stream, ftk_context **out);
void ftk_context_destroy(ftk_context *ctx);
/* Text of the last failure on this context (or of the last failed ftk_context_create when ctx is NULL).  One successful call leaves
 * text too: ftk_direct_track, when its spread launch could not become co-resident and the problem was re-run on one workgroup
 * ("note: ..."; the result is correct, the call took two launches). */
const char *ftk_last_error(const ftk_context *ctx);
int ftk_synchronize(ftk_context *ctx);
/* First-use cost out of the caller's timed region.  The reference constructs its tracker / matcher objects BEFORE it starts its
 * timer (test/test_optical_flow.cpp:64 vs :69, test_descriptor_matcher_brief.cpp:79 vs :84); the first real call of a process
 * otherwise pays for loading the kernels' code objects onto the device and for the first pinned / device staging allocations
 * (about 1.3 ms of a 1.5 ms first TrackFeatures).  `what` is a mask of the families to prepare; the C++ classes call this from
 * their constructors.  Synchronous; results of later calls do not depend on it. */
enum { FTK_WARM_KLT = 1, FTK_WARM_HAMMING = 2, FTK_WARM_COSINE = 4, FTK_WARM_DIRECT = 8, FTK_WARM_FEATURES = 16, FTK_WARM_ALL = 31 };
int ftk_warmup(ftk_context *ctx, unsigned what);
void ftk_default_klt_options(ftk_klt_options *opt);

/*
 * How the trackers sum their normal equations.
 *   FTK_REDUCTION_EXACT (default, the contract): every sum in the reference's row-major pixel order, one dependent add per
 *     term (basic_klt.cpp:139-144, affine_klt.cpp:229-256, lssd_klt.cpp:214-215) — results bit-identical to the CPU path.
 *   FTK_REDUCTION_TREE (throughput mode; measured and reported, never asserted, never the default): the SAME per-pixel products,
 *     summed as per-lane partials combined by a cross-lane butterfly.  The sums then differ from the reference's in the last
 *     bits, and because the convergence test |v|^2 < kMaxConvergeStep turns such differences into an extra or a missing
 *     iteration, a small fraction of the features moves by more than the 1e-3 px the contract allows (bench.py reports max /
 *     p99 / fraction > 1e-3 px and status mismatches next to the speed).  It exists to show what bit-exactness costs.
 *     Implemented for every tracker variant except the two non-fast affine ones (24 sums of a 13 x 13 patch already run as 24
 *     parallel chains; they ignore the setting and stay exact) and for the direct method (ftk_direct_track*), whose 50 700-term
 *     chains per iteration are where the summation order costs most.
 */
enum { FTK_REDUCTION_EXACT = 0, FTK_REDUCTION_TREE = 1 };
int ftk_set_reduction_mode(ftk_context *ctx, int mode);

/* Diagnostics.  The library's FTK_* environment switches (launch shapes, kernel choices; none changes a result) are read ONCE, when
 * the context is created — no entry point calls getenv.  A test or a sweep that flips one for an existing context calls this
 * to have them read again.  Nothing in the reference corresponds to it. */
int ftk_context_refresh_env(ftk_context *ctx);

/* ---- image pyramids resident in HBM ------------------------------------------------------ */

/* A level may have up to 2^24 - 1 rows / columns and fewer than 2^32 pixels (the trackers address pixels with 32-bit
 * offsets); larger levels fail with FTK_E_UNSUPPORTED in the three calls below. */

/* Copies the levels of a host ImagePyramid (ImagePyramid::GetImageConst(i), basic_klt.cpp:22-23)
 * into ONE device allocation, level after level, each level 256-byte aligned. */
int ftk_pyramid_upload(ftk_context *ctx, const ftk_image *host_levels, int32_t n_levels, ftk_pyramid **out);
/* Borrows levels that already live in device memory (e.g. torch tensors); nothing is copied. */
int ftk_pyramid_wrap_device(ftk_context *ctx, const ftk_image *device_levels, int32_t n_levels, ftk_pyramid **out);
/* Replaces ImagePyramid::CreateImagePyramid (call sites test/test_optical_flow.cpp:70-71):
 * uploads (or borrows, image_on_device != 0) the raw image as level 0 and builds levels
 * 1..n_levels-1 on the device with the truncating 2x2 box mean. */
int ftk_pyramid_build(ftk_context *ctx, const uint8_t *image, int32_t rows, int32_t cols, int32_t n_levels, int image_on_device,
                      ftk_pyramid **out);
/* The next frame into an EXISTING pyramid (same geometry): level 0 is overwritten with `image` (rows x cols of the pyramid's
 * level 0) and levels >= 1 are rebuilt on the device — ImagePyramid::CreateImagePyramid for a tracker that is called frame after
 * frame, without an allocation per frame.  image_location: FTK_IMAGE_HOST (pageable or pinned host memory; synchronous: the
 * buffer is free again on return), FTK_IMAGE_DEVICE (device memory, stream-ordered), FTK_IMAGE_HOST_ASYNC (PINNED host memory
 * that stays valid and unchanged until the stream has passed this call; no synchronisation — when the device can address the
 * buffer (hipHostMalloc / hipHostRegister memory) the pyramid launch reads the frame itself over PCIe, without a copy-engine
 * transfer first; otherwise it is copied as with FTK_IMAGE_HOST, stream-ordered).  Only pyramids that own their level 0 (ftk_pyramid_upload,
 * ftk_pyramid_build of a host image) can be refilled. */
enum { FTK_IMAGE_HOST = 0, FTK_IMAGE_DEVICE = 1, FTK_IMAGE_HOST_ASYNC = 2 };
int ftk_pyramid_update(ftk_context *ctx, ftk_pyramid *pyr, const uint8_t *image, int image_location);
int ftk_pyramid_levels(const ftk_pyramid *pyr);
/* Device-side descriptor of one level (data is a device pointer). */
int ftk_pyramid_level(const ftk_pyramid *pyr, int32_t level, ftk_image *out);
/* Copies one level back to host memory (rows*cols bytes); for tests and for callers that still
 * need the pyramid on the host. */
int ftk_pyramid_download_level(ftk_context *ctx, const ftk_pyramid *pyr, int32_t level, uint8_t *host_out);
void ftk_pyramid_destroy(ftk_pyramid *pyr);

/* ---- KLT trackers ------------------------------------------------------------------------ */

/*
 * Replaces OpticalFlow{Basic,Affine,Lssd}Klt::TrackMultipleLevel
 * (basic_klt.cpp:7-57, affine_klt.cpp:6-59, lssd_klt.cpp:7-61) when single_level == 0 and
 * ::TrackSingleLevel (basic_klt.cpp:59-86, affine_klt.cpp:61-91, lssd_klt.cpp:63-94) when
 * single_level != 0 (only level 0 of the pyramids is used).
 *
 * Host buffers, synchronous.  cur_uv and status are in/out exactly as in the reference
 * (prediction in, result out; features whose incoming status > kTracked are skipped;
 * features beyond max_track_points are left untouched).  prior is predict_affine_
 * (affine_klt.h:50; used by the single-level path only) or predict_R_cr_ (lssd_klt.h:53);
 * NULL means identity.  consider_luminance is consider_patch_luminance_ (lssd_klt.h:54).
 * iters (optional, n entries) receives per feature the number of Gauss-Newton iterations that
 * sampled the images, summed over levels — the it(f,l) of the bytes-moved model.
 * The input normalisation of OpticalFlow::TrackFeatures (optical_flow.cpp:6-26: empty input,
 * level mismatch, vector resizing) lives in the host-side classes above this ABI.
 */
int ftk_klt_track(ftk_context *ctx, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur, const float *ref_uv,
                  float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance, int single_level, uint32_t *iters);

/*
 * Same computation on device-resident buffers, asynchronous on the context's stream.
 * d_cur_uv_out / d_status_out may alias the *_in buffers (in-place, as the reference) or be
 * separate (repeatable launches for benchmarking).  d_iters may be NULL.
 * The (u, v) arrays are read and written a PAIR at a time (one 64-bit access per feature): pass them 8-byte aligned — any
 * hipMalloc'ed / pinned buffer and any offset into one by whole features is.
 * Launch order (performance only, results are independent of it): calls of >= 4 096 features keep every feature's
 * iteration count in the context, and from the third consecutive call with the same n on the features are launched
 * longest-first by the counts of two calls before (sorted by one extra workgroup of the launch in between; frame-to-frame
 * coherence of a tracker's feature list; environment FTK_KLT_SCHED=0 keeps list order).
 */
int ftk_klt_track_device(ftk_context *ctx, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur,
                         const float *d_ref_uv, const float *d_cur_uv_in, float *d_cur_uv_out, const uint8_t *d_status_in,
                         uint8_t *d_status_out, int32_t n, const float *prior, int consider_luminance, int single_level, uint32_t *d_iters);

/* Replaces OpticalFlow::ExtractExtendPatchInReferenceImage (optical_flow.cpp:49-102, a public
 * helper of the reference API).  Host buffers: ex_patch has ex_rows*ex_cols floats, valid one
 * byte per pixel; *valid_count receives the return value of the reference function. */
int ftk_extract_extend_patch(ftk_context *ctx, const ftk_pyramid *ref, int32_t level, float u, float v, int32_t ex_rows, int32_t ex_cols,
                             float *ex_patch, uint8_t *valid, uint32_t *valid_count);

/* ---- BRIEF descriptors (the step in front of the matcher) ------ */

/*
 * Replaces feature_detector::BriefDescriptor::Compute (un-vendored Feature_Detector repo; call
 * sites test/test_descriptor_matcher_brief.cpp:70-76, kLength = 256, kHalfPatchSize = 8) for the
 * sampling pattern this repo defines (oracle/oracle_brief.c): bit i = S(p + a_i) < S(p + b_i) on
 * the 3x3 box sums S, p = the feature rounded to the pixel grid, offsets from a fixed-seed LCG.
 * Output is bit-packed — ceil(n_bits / 32) uint32 words per descriptor, bit i in bit (i % 32) of
 * word i / 32 — i.e. exactly what ftk_hamming_match* reads, so on the device path the descriptors
 * never visit the host.  Features closer than half_patch + 1 px to the border get all-zero words.
 */
int ftk_brief_compute(ftk_context *ctx, const ftk_pyramid *image, int32_t level, const float *uv, int32_t n, int32_t n_bits,
                      int32_t half_patch, uint32_t *words);
int ftk_brief_compute_device(ftk_context *ctx, const ftk_pyramid *image, int32_t level, const float *d_uv, int32_t n, int32_t n_bits,
                             int32_t half_patch, uint32_t *d_words);

/* ---- Harris corners (the step in front of the trackers) -------- */

/*
 * Replaces feature_detector::FeaturePointHarrisDetector::DetectGoodFeatures (un-vendored
 * Feature_Detector repo; call sites test/test_optical_flow.cpp:34-39: kMinFeatureDistance = 25,
 * kMinValidResponse = 40, at most 300 features) for the definition in oracle/oracle_harris.c:
 * 3x3 Sobel, 5x5 structure tensor in exact integers, response = (det - 0.04 tr^2) * 1e-6 in fp32,
 * candidates above min_response, window-maximum suppression over (2*min_distance - 1)^2, strongest
 * max_count survivors as (u, v) = (col, row).  uv holds 2*max_count floats; *n_out <= max_count.
 */
int ftk_harris_detect(ftk_context *ctx, const ftk_pyramid *image, int32_t level, int32_t max_count, int32_t min_distance, float min_response,
                      float *uv, int32_t *n_out);
/* The response map alone (rows*cols floats, host memory; 0 outside the 11-pixel border). */
int ftk_harris_response(ftk_context *ctx, const ftk_pyramid *image, int32_t level, float *response);

/* ---- diagnostics ---------------------------------------------------------------------------- */

/*
 * Solves n independent 6x6 systems A x = b with the device's Eigen-compatible pivoted LDLT — the
 * primitive inside the affine trackers (affine_klt.cpp:103, affine_klt_fast.cpp:39 `H.ldlt().solve(b)`)
 * and the direct method (direct_method_tracker.cpp:170) — so that tests can compare it with the oracle
 * on matrices a tracker would rarely produce (ties, zero pivots, NaN).  a: n x 36 row-major symmetric,
 * b: n x 6, x: n x 6; host buffers.
 */
int ftk_ldlt6_solve(ftk_context *ctx, const float *a, const float *b, float *x, int32_t n);

/* ---- direct method ---------------------------------------------- */

/* DirectMethodOptions, src/direct_method_tracker/direct_method_tracker.h:20-28 */
typedef struct ftk_direct_options {
    uint32_t max_track_points;   /* kMaxTrackPointsNumber (500)  */
    uint32_t max_iteration;      /* kMaxIteration         (15)   */
    int32_t half_rows;           /* kPatchRowHalfSize     (6)    */
    int32_t half_cols;           /* kPatchColHalfSize     (6)    */
    float max_converge_step;     /* kMaxConvergeStep      (1e-6) */
    float max_converge_residual; /* kMaxConvergeResidual  (2.0; the reference never reads it) */
    int32_t method;              /* kMethod (FTK_METHOD_DIRECT); kInverse / kFast are empty stubs in the reference and no-ops here */
} ftk_direct_options;
void ftk_default_direct_options(ftk_direct_options *opt);

/*
 * Replaces DirectMethod::TrackFeatures, camera-frame overload (direct_method_tracker.cpp:35-86) with
 * TrackSingleLevel -> TrackAllFeaturesDirect (:88-106, :115-192): photometric Gauss-Newton on ONE pose
 * (q_rc, p_rc) over all features jointly, coarse to fine.  K = {fx, fy, cx, cy}; p_c_in_ref = n x 3 points in
 * the reference camera frame; cur_uv in/out (the caller applies ":42-44 sizes differ -> cur = ref");
 * q_rc = (w, x, y, z) and p_rc in/out; status in/out with status_valid = 0 meaning "was not sized n"
 * (reset to kTracked, :73-75).  iterations (optional) = Gauss-Newton iterations over all levels.
 * The sums of the normal equations keep the scalar loop's order (feature by feature, pixel by pixel),
 * so pose, pixels and iteration counts are those of the scalar code.  The world-frame overload
 * (:8-33) is host-side quaternion algebra around this call and lives in the C++ class.
 * Any number of tracked features: up to 3072 their per-iteration projections live in LDS, above that in a
 * context-owned device buffer (same arithmetic, same order of the sums).
 */
int ftk_direct_track(ftk_context *ctx, const ftk_direct_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur, const float *K,
                     const float *p_c_in_ref, const float *ref_uv, float *cur_uv, int32_t n, float *q_rc_wxyz, float *p_rc, uint8_t *status,
                     int status_valid, uint32_t *iterations);

/* A batch of independent pose problems in ONE launch (one workgroup each), buffers device-resident.
 * d_pose = 7 floats: q_rc (w, x, y, z) then p_rc.  All problems share the options and the pyramid depth. */
typedef struct ftk_direct_problem {
    const ftk_pyramid *ref;
    const ftk_pyramid *cur;
    float K[4];
    const float *d_p_c_in_ref;
    const float *d_ref_uv;
    float *d_cur_uv;
    int32_t n;
    float *d_pose;
    uint8_t *d_status;
    int32_t status_valid;
    uint32_t *d_iterations; /* may be NULL */
} ftk_direct_problem;
int ftk_direct_track_batch_device(ftk_context *ctx, const ftk_direct_options *opt, const ftk_direct_problem *problems, int32_t n_problems);

/* ---- features sharded over the GPUs of one node ------------------------ */

/*
 * The reference has no parallelism of any kind; what shards is the independence of its units: every feature's
 * computation reads the two pyramids and its own (ref_uv, cur_uv, status) only (basic_klt.cpp:13-54, affine_klt.cpp:12-56,
 * lssd_klt.cpp:13-58), every reference descriptor's scan is its own (descriptor_matcher.h:67-76).  One process per GPU;
 * rank r of `world` works on the contiguous block ftk_shard_bounds(n, world, r) with both pyramids (or all candidates)
 * replicated, and ONE all-gather of the packed result shards — RCCL ncclAllGather over xGMI, issued from this library on the
 * context's stream right behind the kernel — gives every rank the complete result in the original order, identical to the
 * single-GPU result.  kMaxTrackPointsNumber stays a cap on the GLOBAL feature index (basic_klt.cpp:9).
 *
 * Bootstrap: rank 0 calls ftk_comm_unique_id and hands the 128 bytes to the other ranks by any means (a file, MPI,
 * torch.distributed ...); every rank then calls ftk_comm_create.  RCCL is bound at run time (dlopen): without it these
 * calls fail with FTK_E_UNSUPPORTED and everything else keeps working.  world == 1 with a NULL id needs no RCCL at all.
 */
typedef struct ftk_comm ftk_comm;
#define FTK_UNIQUE_ID_BYTES 128
/* [begin, end) of rank's block: the first n % world ranks hold one unit more. */
void ftk_shard_bounds(int32_t n, int32_t world, int32_t rank, int32_t *begin, int32_t *end);
/* Bytes of ONE rank's packed tracker shard: ceil(n / world) * (8 B (u, v) + 1 B status), rounded up to 16 B. */
size_t ftk_klt_shard_bytes(int32_t n, int32_t world);
int ftk_comm_unique_id(void *id_out /* FTK_UNIQUE_ID_BYTES */);
int ftk_comm_create(ftk_context *ctx, int32_t rank, int32_t world, const void *unique_id, ftk_comm **out);
void ftk_comm_destroy(ftk_comm *comm);
int ftk_comm_rank(const ftk_comm *comm);
int ftk_comm_world(const ftk_comm *comm);

/*
 * ftk_klt_track_device over the ranks of `comm`.  Every rank passes the SAME full-length device buffers (n features);
 * on return (asynchronously, stream-ordered) every rank's d_cur_uv_out / d_status_out hold all n results.
 * Launches: the tracker kernel on this rank's block, ncclAllGather of the packed shards, one scatter kernel.
 * d_iters (optional, n entries) receives only this rank's block.
 * Failure symmetry: a rank whose tracker launch fails still takes part in the collective with a POISONED shard (every byte
 * 0xFF: status 255, NaN coordinates) and returns its error, so its peers finish instead of blocking; the host-buffer form below
 * turns a poisoned block into FTK_E_HIP on every rank, the device form leaves it visible in the data.  (A rank that cannot even
 * allocate its exchange buffers cannot contribute: destroy the communicator's process group in that case.)
 * HIP-graph capture: the exchange buffers grow on demand (hipFree / hipMalloc, which a capture does not allow), so make ONE eager
 * call with the largest n before capturing any.
 */
int ftk_klt_track_sharded_device(ftk_context *ctx, ftk_comm *comm, int model, const ftk_klt_options *opt, const ftk_pyramid *ref,
                                 const ftk_pyramid *cur, const float *d_ref_uv, const float *d_cur_uv_in, float *d_cur_uv_out,
                                 const uint8_t *d_status_in, uint8_t *d_status_out, int32_t n, const float *prior, int consider_luminance,
                                 int single_level, uint32_t *d_iters);
/* Host-buffer form (synchronous; cur_uv / status in/out as in ftk_klt_track): what the C++ OpticalFlow classes call when the
 * process is one rank of several.  iters (optional) is filled for this rank's block only, 0 elsewhere. */
int ftk_klt_track_sharded(ftk_context *ctx, ftk_comm *comm, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur,
                          const float *ref_uv, float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance, int single_level,
                          uint32_t *iters);
/* The two halves of the call above for callers that bring their own collective (e.g. torch.distributed): rank's block
 * tracked into its packed shard (ftk_klt_shard_bytes(n, world) bytes), and the scatter of `world` gathered shards. */
int ftk_klt_track_shard_device(ftk_context *ctx, int32_t rank, int32_t world, int model, const ftk_klt_options *opt, const ftk_pyramid *ref,
                               const ftk_pyramid *cur, const float *d_ref_uv, const float *d_cur_uv_in, const uint8_t *d_status_in, int32_t n,
                               const float *prior, int consider_luminance, int single_level, void *d_packed_shard, uint32_t *d_iters);
int ftk_klt_unpack_shards_device(ftk_context *ctx, const void *d_gathered, int32_t n, int32_t world, float *d_cur_uv_out, uint8_t *d_status_out);
/* ftk_hamming_match_device with the reference rows sharded over the ranks (candidates replicated); d_index_pairs (n_ref
 * entries, in/out as in the single-GPU call) is complete on every rank afterwards. */
int ftk_hamming_match_sharded_device(ftk_context *ctx, ftk_comm *comm, const uint32_t *d_ref_words, int32_t n_ref, const uint32_t *d_cur_words,
                                     int32_t n_cur, int32_t n_words, int32_t n_bits, float max_distance, const float *d_pred_uv,
                                     const float *d_cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *d_index_pairs);

/* ---- descriptor matcher ------------------------------------------------------------------ */

/*
 * Replaces DescriptorMatcher<BriefType>::ForceMatch (descriptor_matcher.h:55-79) and
 * ::NearbyMatch (:90-124) for the per-bit Hamming distance of
 * test/test_descriptor_matcher_brief.cpp:33-45.  Descriptors are bit-packed: n_words uint32
 * per descriptor (any n_words >= 1: widths other than 1, 2, 4, 8, 16 words are zero-padded on the device
 * or take a generic scan — same indices), bit i of the descriptor in bit (i % 32) of word i / 32, unused
 * high bits 0.
 * n_bits == 0 reproduces ComputeDistance's "empty descriptor" answer (kMaxInt32).
 * 256- and 512-bit descriptors (n_words 8 / 16) are compared on the matrix cores (the distances are exact integers out of
 * int8 MFMAs: same indices as the popcount scans that serve the other widths).  Device descriptor pointers are read with 16-byte
 * loads: pass 16-byte-aligned arrays (any hipMalloc'ed buffer, and any row offset into one when n_words is a multiple of 4).
 * pred_uv == NULL selects ForceMatch; otherwise NearbyMatch with the window test
 * |pred.u - cur.u| > max_col_distance || |pred.v - cur.v| > max_row_distance -> skip.
 * index_pairs is in/out (n_ref entries): written only where a candidate beats the threshold,
 * exactly like the reference, whose caller-visible reset to -1 happens only on a size mismatch
 * (descriptor_matcher.h:60-62) and therefore lives in the host-side class.
 * Returns FTK_OK with *matched_ok = 0 when n_cur == 0 (the reference's `return false`).
 */
int ftk_hamming_match(ftk_context *ctx, const uint32_t *ref_words, int32_t n_ref, const uint32_t *cur_words, int32_t n_cur, int32_t n_words,
                      int32_t n_bits, float max_distance, const float *pred_uv, const float *cur_uv, int32_t max_col_distance,
                      int32_t max_row_distance, int32_t *index_pairs, int *matched_ok);

/* Device-resident, asynchronous variant.  d_workspace must hold n_ref uint64 (packed
 * (distance, index) keys); pass NULL to let the context allocate and cache one. */
int ftk_hamming_match_device(ftk_context *ctx, const uint32_t *d_ref_words, int32_t n_ref, const uint32_t *d_cur_words, int32_t n_cur,
                             int32_t n_words, int32_t n_bits, float max_distance, const float *d_pred_uv, const float *d_cur_uv,
                             int32_t max_col_distance, int32_t max_row_distance, int32_t *d_index_pairs, uint64_t *d_workspace);

/*
 * Float descriptors.  Replaces DescriptorMatcher<T>::ForceMatch
 * (descriptor_matcher.h:55-79) / ::NearbyMatch (:90-124) for the cosine distance the reference's
 * SuperPoint / DISK callers define (test/test_descriptor_matcher_superpoint.cpp:32-34,
 * test_descriptor_matcher_disk.cpp:32-34):
 *     0.5f - ref.dot(cur) / ref.norm() / cur.norm() * 0.5f
 * Descriptors are row-major float[n][dim] (256 for SuperPoint, 128 for DISK; any dim <= 4096).
 * The all-pairs contraction runs on the matrix cores in fp16 only to shortlist the pairs that can
 * decide a row; every deciding comparison is made on the distance evaluated in fp32 in Eigen's
 * reduction order, so index_pairs is what the scalar loop returns (ties -> lowest index, strict
 * threshold).  pred_uv == NULL selects ForceMatch.  index_pairs in/out and *matched_ok as in
 * ftk_hamming_match.  NearbyMatch (both matchers): blocks of (reference rows x candidates) whose
 * bounding boxes lie farther apart than the window are skipped — same indices for any order of the
 * features, less time when they are in spatial order (a detector scanning the image).
 */
int ftk_cosine_match(ftk_context *ctx, const float *ref_desc, int32_t n_ref, const float *cur_desc, int32_t n_cur, int32_t dim, float max_distance,
                     const float *pred_uv, const float *cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *index_pairs,
                     int *matched_ok);
/* Device-resident, asynchronous variant (workspace cached in the context). */
int ftk_cosine_match_device(ftk_context *ctx, const float *d_ref_desc, int32_t n_ref, const float *d_cur_desc, int32_t n_cur, int32_t dim,
                            float max_distance, const float *d_pred_uv, const float *d_cur_uv, int32_t max_col_distance,
                            int32_t max_row_distance, int32_t *d_index_pairs);

/* Replaces DescriptorMatcher::FillMatchedPixelByPairIndices (descriptor_matcher.h:135-157).
 * Pure index -> pixel gather on host buffers (O(n_ref), not worth a launch); status is in/out. */
int ftk_fill_matched_pixels(const int32_t *index_pairs, int32_t n_ref, const float *cur_uv, int32_t n_cur, float *matched_uv, uint8_t *status);

#ifdef __cplusplus
}
#endif
#endif /* FTK_H_ */
