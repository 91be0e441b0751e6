// peer.cpp — section C of include/dsc_mi355x.h: raw device buffers, IPC export / open and stream-ordered direct
// pushes, the building blocks of the multi-GPU reassembly.  The reference has no counterpart
// (dsc/include/dsc_backend.h:11-13: CPU only, no communication layer).
//
// Design (one process per GPU): every rank owns a [P x shard] destination; rank r's transform writes slot r in
// place, and after each chunk of rows it pushes that chunk into slot r of every peer's destination with one
// asynchronous copy per peer, each on its own stream — xGMI is point-to-point, one link per peer, so the P-1
// copies run on P-1 different links at once (the "direct / mesh" all-gather of SURVEY 8e, ~S/153 GB/s instead of
// the ring's 7 S/153 GB/s).  A push waits (event) for the work enqueued so far on the context's stream and
// nothing waits for the push until dsc_peer_wait: the transform of the next chunk runs underneath it.
#include "dsc_internal.h"

#include <cstring>

static_assert(sizeof(dsc_ipc_handle) >= sizeof(hipIpcMemHandle_t), "dsc_ipc_handle must hold a hipIpcMemHandle_t");

namespace {
constexpr int kLanes = 8;
}

// Soft failure for the calls whose errors the caller reports (a missing peer, an unsupported mapping): print, return.
#define PEER_TRY(call, ret)                                                                                      \
    do {                                                                                                         \
        hipError_t perr_ = (call);                                                                               \
        if (perr_ != hipSuccess) {                                                                               \
            fprintf(stderr, "%s: %s -> %s\n", __func__, #call, hipGetErrorString(perr_));                        \
            (void) hipGetLastError();                                                                            \
            return ret;                                                                                          \
        }                                                                                                        \
    } while (0)

extern "C" void *dsc_device_alloc(dsc_ctx *ctx, size_t nbytes) {
    DSC_ASSERT(ctx != nullptr && nbytes > 0);
    void *p = nullptr;
    PEER_TRY(hipSetDevice(ctx->device), nullptr);
    PEER_TRY(hipMalloc(&p, nbytes), nullptr);
    return p;
}

extern "C" void dsc_device_free(dsc_ctx *ctx, void *ptr) {
    if (ptr == nullptr) return;
    HIP_CHECK(hipSetDevice(ctx->device));
    HIP_CHECK(hipFree(ptr));
}

extern "C" dsc_tensor *dsc_tensor_from_device_ptr(dsc_ctx *ctx, void *ptr, size_t nbytes, int n_dim, const int *shape, dsc_dtype dtype) {
    DSC_ASSERT(ctx != nullptr && ptr != nullptr && shape != nullptr);
    DSC_ASSERT(((size_t) ptr & 7) == 0);
    return dsc_new_tensor_over(ctx, ptr, nbytes, n_dim, shape, dtype);
}

extern "C" int dsc_ipc_export(dsc_ctx *ctx, void *ptr, dsc_ipc_handle *out) {
    DSC_ASSERT(ctx != nullptr && ptr != nullptr && out != nullptr);
    hipIpcMemHandle_t h;
    PEER_TRY(hipSetDevice(ctx->device), -1);
    PEER_TRY(hipIpcGetMemHandle(&h, ptr), -1);
    memset(out, 0, sizeof(*out));
    memcpy(out->bytes, &h, sizeof(h));
    return 0;
}

extern "C" void *dsc_ipc_open(dsc_ctx *ctx, const dsc_ipc_handle *handle) {
    DSC_ASSERT(ctx != nullptr && handle != nullptr);
    hipIpcMemHandle_t h;
    memcpy(&h, handle->bytes, sizeof(h));
    void *p = nullptr;
    PEER_TRY(hipSetDevice(ctx->device), nullptr);
    PEER_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), nullptr);
    return p;
}

extern "C" int dsc_ipc_close(dsc_ctx *ctx, void *mapped) {
    if (mapped == nullptr) return 0;
    PEER_TRY(hipSetDevice(ctx->device), -1);
    PEER_TRY(hipIpcCloseMemHandle(mapped), -1);
    return 0;
}

extern "C" int dsc_peer_lanes(void) { return kLanes; }

static int ensure_lanes(dsc_ctx *ctx) {
    if (!ctx->peer_streams.empty()) return 0;
    PEER_TRY(hipSetDevice(ctx->device), -1);
    for (int i = 0; i < kLanes; ++i) {
        hipStream_t s;
        PEER_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), -1);
        ctx->peer_streams.push_back(s);
    }
    PEER_TRY(hipEventCreateWithFlags(&ctx->peer_ready, hipEventDisableTiming), -1);
    return 0;
}

extern "C" int dsc_peer_push(dsc_ctx *ctx, void *dst, const void *src, size_t nbytes, int lane) {
    DSC_ASSERT(ctx != nullptr && dst != nullptr && src != nullptr);
    DSC_ASSERT(lane >= 0 && lane < kLanes);
    if (nbytes == 0) return 0;
    if (ensure_lanes(ctx) != 0) return -1;
    hipStream_t s = ctx->peer_streams[lane];
    PEER_TRY(hipEventRecord(ctx->peer_ready, ctx->stream), -1);       // everything enqueued so far produced `src`
    PEER_TRY(hipStreamWaitEvent(s, ctx->peer_ready, 0), -1);
    PEER_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, s), -1);
    return 0;
}

extern "C" int dsc_peer_wait(dsc_ctx *ctx) {
    DSC_ASSERT(ctx != nullptr);
    for (hipStream_t s : ctx->peer_streams) PEER_TRY(hipStreamSynchronize(s), -1);
    return 0;
}

void dsc_peer_release(dsc_ctx *ctx) {
    for (hipStream_t s : ctx->peer_streams) HIP_CHECK(hipStreamDestroy(s));
    if (!ctx->peer_streams.empty()) HIP_CHECK(hipEventDestroy(ctx->peer_ready));
    ctx->peer_streams.clear();
}
