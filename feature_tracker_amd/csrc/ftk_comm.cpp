// ftk_comm.cpp — the multi-GPU side of the C ABI (include/ftk.h, "features sharded over the GPUs of one node").
//
// features (tracker) and reference rows (matcher) are independent units, so rank r of `world`
// processes works on the contiguous block ftk_shard_bounds(n, world, r) with the pyramids / candidates replicated, and
// ONE all-gather of the packed result shards gives every rank the complete result in the original order.  No other
// exchange exists on this path.  The collective is RCCL's ncclAllGather over xGMI, issued from here on the context's
// stream right behind the kernel (asynchronous, HIP-graph capturable).  RCCL is bound at run time (dlopen of
// librccl.so.1 — the copy a host process has already loaded, e.g. PyTorch's, is reused), so single-GPU users need no
// RCCL and the communicator entry points fail loudly where it is missing.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "ftk_internal.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) {  // a copy the process already holds (same SONAME) first
            api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (api.handle) {
                break;
            }
        }
        for (int i = 0; i < 2 && !api.handle; ++i) {
            api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.handle) {
            api.error = std::string("RCCL is not loadable (") + dlerror() + ")";
            return;
        }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.handle, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
            api.error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
            api.handle = nullptr;
        }
    });
    return api;
}

}  // namespace

struct ftk_comm {
    ftk_context *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int32_t rank = 0, world = 1;
    // this rank's packed result shard and the gathered buffer (world shards), grown on demand
    void *packed = nullptr;
    size_t packed_bytes = 0;
    void *gathered = nullptr;
    size_t gathered_bytes = 0;
    // device staging of the host-buffer entry point (ref_uv | cur_uv | status | iters for all n features)
    void *stage = nullptr;
    size_t stage_bytes = 0;
};

namespace {

// One rank's packed tracker shard: [u, v] float pairs of `cap` features, then `cap` status bytes, padded to 16 B (every
// rank's slice of the gathered buffer keeps the pairs aligned).  Same layout as feature_tracker_amd/dist.py.
size_t klt_shard_bytes(int32_t cap) { return ((size_t)cap * 9 + 15) / 16 * 16; }

int32_t shard_cap(int32_t n, int32_t world) { return (n + world - 1) / world; }

void shard_range(int32_t n, int32_t world, int32_t rank, int32_t *begin, int32_t *end) {
    const int32_t base = n / world, extra = n % world;
    *begin = rank * base + (rank < extra ? rank : extra);
    *end = *begin + base + (rank < extra ? 1 : 0);
}

int ensure_comm_buffers(ftk_comm *c, size_t shard_bytes) {
    int rc = ftk_ensure_device_buffer(c->ctx, &c->packed, &c->packed_bytes, shard_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    return ftk_ensure_device_buffer(c->ctx, &c->gathered, &c->gathered_bytes, shard_bytes * (size_t)c->world);
}

int all_gather(ftk_comm *c, size_t shard_bytes) {
    if (c->comm == nullptr) {  // world == 1 made without an RCCL id: nothing to exchange, the "gathered" buffer is the shard
        FTK_HIP(c->ctx, hipMemcpyAsync(c->gathered, c->packed, shard_bytes, hipMemcpyDeviceToDevice, c->ctx->stream));
        return FTK_OK;
    }
    const ncclResult_t r = rccl().AllGather(c->packed, c->gathered, shard_bytes, ncclUint8, c->comm, c->ctx->stream);
    if (r != ncclSuccess) {
        return ftk_fail(c->ctx, FTK_E_HIP, "ncclAllGather failed: %s", rccl().GetErrorString(r));
    }
    return FTK_OK;
}

}  // namespace

extern "C" {

void ftk_shard_bounds(int32_t n, int32_t world, int32_t rank, int32_t *begin, int32_t *end) {
    int32_t b = 0, e = 0;
    if (n > 0 && world > 0 && rank >= 0 && rank < world) {
        shard_range(n, world, rank, &b, &e);
    }
    if (begin) {
        *begin = b;
    }
    if (end) {
        *end = e;
    }
}

size_t ftk_klt_shard_bytes(int32_t n, int32_t world) { return (n > 0 && world > 0) ? klt_shard_bytes(shard_cap(n, world)) : 0; }

int ftk_comm_unique_id(void *id_out) {
    if (!id_out) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "comm_unique_id: null output");
    }
    RcclApi &api = rccl();
    if (!api.handle) {
        return ftk_fail(nullptr, FTK_E_UNSUPPORTED, "comm_unique_id: %s", api.error.c_str());
    }
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) {
        return ftk_fail(nullptr, FTK_E_HIP, "ncclGetUniqueId failed: %s", api.GetErrorString(r));
    }
    static_assert(sizeof(id) == FTK_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return FTK_OK;
}

int ftk_comm_create(ftk_context *ctx, int32_t rank, int32_t world, const void *unique_id, ftk_comm **out) {
    if (!ctx) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "comm_create: null context");
    }
    FTK_LOCK(ctx);
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id)) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "comm_create: bad rank %d / world %d / id", rank, world);
    }
    *out = nullptr;
    ftk_comm *c = new (std::nothrow) ftk_comm();
    if (!c) {
        return ftk_fail(ctx, FTK_E_OUT_OF_MEMORY, "comm_create: host allocation failed");
    }
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (world > 1 || unique_id) {
        RcclApi &api = rccl();
        if (!api.handle) {
            delete c;
            return ftk_fail(ctx, FTK_E_UNSUPPORTED, "comm_create: %s", api.error.c_str());
        }
        const hipError_t de = hipSetDevice(ctx->device);
        if (de != hipSuccess) {
            delete c;
            return ftk_fail(ctx, FTK_E_HIP, "comm_create: hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(de));
        }
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        const ncclResult_t r = api.CommInitRank(&c->comm, world, id, rank);
        if (r != ncclSuccess) {
            delete c;
            return ftk_fail(ctx, FTK_E_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, api.GetErrorString(r));
        }
    }
    *out = c;
    return FTK_OK;
}

void ftk_comm_destroy(ftk_comm *c) {
    if (!c) {
        return;
    }
    {
        FTK_LOCK(c->ctx);
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        if (c->comm) {
            (void)rccl().CommDestroy(c->comm);
        }
        if (c->packed) {
            (void)hipFree(c->packed);
        }
        if (c->gathered) {
            (void)hipFree(c->gathered);
        }
        if (c->stage) {
            (void)hipFree(c->stage);
        }
    }
    delete c;
}

int ftk_comm_rank(const ftk_comm *c) { return c ? c->rank : -1; }
int ftk_comm_world(const ftk_comm *c) { return c ? c->world : 0; }

int ftk_klt_track_shard_device(ftk_context *ctx, int32_t rank, int32_t world, int model, const ftk_klt_options *opt, const ftk_pyramid *ref,
                               const ftk_pyramid *cur, const float *d_ref_uv, const float *d_cur_uv_in, const uint8_t *d_status_in, int32_t n,
                               const float *prior, int consider_luminance, int single_level, void *d_packed_shard, uint32_t *d_iters) {
    if (!ctx) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "klt_track_shard_device: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0 || world < 1 || rank < 0 || rank >= world || !opt) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_shard_device: bad n %d / rank %d / world %d", n, rank, world);
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!d_ref_uv || !d_cur_uv_in || !d_status_in || !d_packed_shard) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_shard_device: null buffer");
    }
    int32_t begin = 0, end = 0;
    shard_range(n, world, rank, &begin, &end);
    const int32_t m = end - begin, cap = shard_cap(n, world);
    if (m <= 0) {
        return FTK_OK;
    }
    // kMaxTrackPointsNumber caps the GLOBAL feature index (basic_klt.cpp:9): this block tracks what is left of it
    ftk_klt_options local = *opt;
    const int64_t left = (int64_t)opt->max_track_points - begin;
    local.max_track_points = left <= 0 ? 0u : (left > m ? (uint32_t)m : (uint32_t)left);
    float *uv_out = static_cast<float *>(d_packed_shard);
    uint8_t *st_out = static_cast<uint8_t *>(d_packed_shard) + (size_t)cap * 8;
    return ftk_klt_track_device(ctx, model, &local, ref, cur, d_ref_uv + 2 * (size_t)begin, d_cur_uv_in + 2 * (size_t)begin, uv_out,
                                d_status_in + begin, st_out, m, prior, consider_luminance, single_level, d_iters ? d_iters + begin : nullptr);
}

int ftk_klt_unpack_shards_device(ftk_context *ctx, const void *d_gathered, int32_t n, int32_t world, float *d_cur_uv_out, uint8_t *d_status_out) {
    if (!ctx) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "klt_unpack_shards_device: null context");
    }
    FTK_LOCK(ctx);
    if (n < 0 || world < 1) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_unpack_shards_device: bad n %d / world %d", n, world);
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!d_gathered || !d_cur_uv_out || !d_status_out) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_unpack_shards_device: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const int32_t cap = shard_cap(n, world);
    FTK_HIP(ctx, ftk::unpack_klt_shards_launch(static_cast<const uint8_t *>(d_gathered), n, world, cap, (int64_t)klt_shard_bytes(cap), d_cur_uv_out,
                                               d_status_out, ctx->stream));
    return FTK_OK;
}

int ftk_klt_track_sharded_device(ftk_context *ctx, ftk_comm *comm, int model, const ftk_klt_options *opt, const ftk_pyramid *ref,
                                 const ftk_pyramid *cur, const float *d_ref_uv, const float *d_cur_uv_in, float *d_cur_uv_out,
                                 const uint8_t *d_status_in, uint8_t *d_status_out, int32_t n, const float *prior, int consider_luminance,
                                 int single_level, uint32_t *d_iters) {
    if (!ctx) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "klt_track_sharded_device: null context");
    }
    FTK_LOCK(ctx);
    if (!comm || comm->ctx != ctx) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_sharded_device: the communicator belongs to another context");
    }
    if (n < 0) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_sharded_device: negative feature count");
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!d_cur_uv_out || !d_status_out) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_sharded_device: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t shard = klt_shard_bytes(shard_cap(n, comm->world));
    int rc = ensure_comm_buffers(comm, shard);
    if (rc != FTK_OK) {
        return rc;
    }
    rc = ftk_klt_track_shard_device(ctx, comm->rank, comm->world, model, opt, ref, cur, d_ref_uv, d_cur_uv_in, d_status_in, n, prior,
                                    consider_luminance, single_level, comm->packed, d_iters);
    if (rc != FTK_OK) {
        // The peers are (or will be) inside the collective: leaving without it would block them for ever.  Contribute a POISONED
        // shard instead — every byte 0xFF: status 255 is no TrackStatus and (u, v) are NaNs — so that every rank, this one
        // included, sees the failure in the data (the host-buffer entry turns it into an error code), and report the local error.
        const std::string local_error = ctx->error;
        if (hipMemsetAsync(comm->packed, 0xFF, shard, ctx->stream) == hipSuccess && all_gather(comm, shard) == FTK_OK) {
            (void)ftk_klt_unpack_shards_device(ctx, comm->gathered, n, comm->world, d_cur_uv_out, d_status_out);
        }
        ctx->error = local_error;
        return rc;
    }
    rc = all_gather(comm, shard);
    if (rc != FTK_OK) {
        return rc;
    }
    return ftk_klt_unpack_shards_device(ctx, comm->gathered, n, comm->world, d_cur_uv_out, d_status_out);
}

int ftk_klt_track_sharded(ftk_context *ctx, ftk_comm *comm, int model, const ftk_klt_options *opt, const ftk_pyramid *ref, const ftk_pyramid *cur,
                          const float *ref_uv, float *cur_uv, uint8_t *status, int32_t n, const float *prior, int consider_luminance, int single_level,
                          uint32_t *iters) {
    if (!ctx) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "klt_track_sharded: null context");
    }
    FTK_LOCK(ctx);
    if (!comm || comm->ctx != ctx) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_sharded: the communicator belongs to another context");
    }
    if (n < 0) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_sharded: negative feature count");
    }
    if (n == 0) {
        return FTK_OK;
    }
    if (!ref_uv || !cur_uv || !status) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "klt_track_sharded: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    const size_t uv_bytes = ftk_align_up(sizeof(float) * 2 * (size_t)n, 256), st_bytes = ftk_align_up((size_t)n, 256);
    const size_t it_bytes = ftk_align_up(sizeof(uint32_t) * (size_t)n, 256);
    int rc = ftk_ensure_device_buffer(ctx, &comm->stage, &comm->stage_bytes, 2 * uv_bytes + st_bytes + it_bytes);
    if (rc != FTK_OK) {
        return rc;
    }
    uint8_t *base = static_cast<uint8_t *>(comm->stage);
    float *d_ref = reinterpret_cast<float *>(base), *d_cur = reinterpret_cast<float *>(base + uv_bytes);
    uint8_t *d_st = base + 2 * uv_bytes;
    uint32_t *d_it = reinterpret_cast<uint32_t *>(base + 2 * uv_bytes + st_bytes);
    FTK_HIP(ctx, hipMemcpyAsync(d_ref, ref_uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_cur, cur_uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(d_st, status, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (iters) {
        FTK_HIP(ctx, hipMemsetAsync(d_it, 0, sizeof(uint32_t) * (size_t)n, ctx->stream));
    }
    rc = ftk_klt_track_sharded_device(ctx, comm, model, opt, ref, cur, d_ref, d_cur, d_cur, d_st, d_st, n, prior, consider_luminance, single_level,
                                      iters ? d_it : nullptr);
    if (rc != FTK_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    FTK_HIP(ctx, hipMemcpyAsync(cur_uv, d_cur, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FTK_HIP(ctx, hipMemcpyAsync(status, d_st, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (iters) {
        FTK_HIP(ctx, hipMemcpyAsync(iters, d_it, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    FTK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // A rank whose tracker launch failed contributes a poisoned shard (every byte 0xFF) instead of leaving its peers blocked in
    // the collective: report it here, on every rank, instead of handing NaNs to the caller.
    for (int32_t r = 0; r < comm->world; ++r) {
        int32_t b = 0, e = 0;
        shard_range(n, comm->world, r, &b, &e);
        bool poisoned = e > b;
        for (int32_t i = b; i < e && poisoned; ++i) {
            uint32_t bits[2];
            memcpy(bits, cur_uv + 2 * (size_t)i, sizeof(bits));
            poisoned = status[i] == 0xFF && bits[0] == 0xFFFFFFFFu && bits[1] == 0xFFFFFFFFu;
        }
        if (poisoned) {
            return ftk_fail(ctx, FTK_E_HIP, "klt_track_sharded: rank %d of %d reported a failed tracker launch (poisoned result shard)", r, comm->world);
        }
    }
    return FTK_OK;
}

int ftk_hamming_match_sharded_device(ftk_context *ctx, ftk_comm *comm, const uint32_t *d_ref_words, int32_t n_ref, const uint32_t *d_cur_words,
                                     int32_t n_cur, int32_t n_words, int32_t n_bits, float max_distance, const float *d_pred_uv,
                                     const float *d_cur_uv, int32_t max_col_distance, int32_t max_row_distance, int32_t *d_index_pairs) {
    if (!ctx) {
        return ftk_fail(nullptr, FTK_E_INVALID_ARGUMENT, "hamming_match_sharded_device: null context");
    }
    FTK_LOCK(ctx);
    if (!comm || comm->ctx != ctx) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_sharded_device: the communicator belongs to another context");
    }
    if (n_ref < 0 || n_cur < 0 || n_words < 1) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_sharded_device: bad sizes");
    }
    if (n_ref == 0 || n_cur == 0) {
        return FTK_OK;
    }
    if (!d_ref_words || !d_cur_words || !d_index_pairs) {
        return ftk_fail(ctx, FTK_E_INVALID_ARGUMENT, "hamming_match_sharded_device: null buffer");
    }
    FTK_HIP(ctx, hipSetDevice(ctx->device));
    // reference rows block-partitioned, candidates replicated; index_pairs is in/out (stale entries
    // survive, descriptor_matcher.h:60-62), so the shard is seeded from the caller's entries of this block
    int32_t begin = 0, end = 0;
    shard_range(n_ref, comm->world, comm->rank, &begin, &end);
    const int32_t m = end - begin, cap = shard_cap(n_ref, comm->world);
    const size_t shard = ((size_t)cap * sizeof(int32_t) + 15) / 16 * 16;
    int rc = ensure_comm_buffers(comm, shard);
    if (rc != FTK_OK) {
        return rc;
    }
    int32_t *local = static_cast<int32_t *>(comm->packed);
    if (m > 0) {
        FTK_HIP(ctx, hipMemcpyAsync(local, d_index_pairs + begin, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToDevice, ctx->stream));
        rc = ftk_hamming_match_device(ctx, d_ref_words + (size_t)begin * n_words, m, d_cur_words, n_cur, n_words, n_bits, max_distance,
                                      d_pred_uv ? d_pred_uv + 2 * (size_t)begin : nullptr, d_cur_uv, max_col_distance, max_row_distance, local, nullptr);
        if (rc != FTK_OK) {
            return rc;
        }
    }
    rc = all_gather(comm, shard);
    if (rc != FTK_OK) {
        return rc;
    }
    for (int32_t r = 0; r < comm->world; ++r) {
        int32_t b = 0, e = 0;
        shard_range(n_ref, comm->world, r, &b, &e);
        if (e > b) {
            FTK_HIP(ctx, hipMemcpyAsync(d_index_pairs + b, static_cast<const uint8_t *>(comm->gathered) + shard * (size_t)r, sizeof(int32_t) * (size_t)(e - b),
                                        hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    return FTK_OK;
}

}  // extern "C"
