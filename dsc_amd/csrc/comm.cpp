// comm.cpp — section C of include/dsc_mi355x.h, the collective half: an RCCL communicator (one rank per process and GPU) and the
// all-gather that puts the P output shards of a batch-sharded transform next to each other on every GPU (north_star:
// "an RCCL all-gather over xGMI to reassemble the output").  A C or C++ host — the reference's users write C++ against
// dsc/api/dsc_api.h:24-34 — gets the collective without Python; dsc_amd/shard.py's 'allgather' drives the same entry points.
// The reference has no counterpart (dsc/include/dsc_backend.h:11-13: one backend, no communication layer).
//
// RCCL is bound at first use with dlopen (the library is 570 MB; a single-GPU user of the FFT path never loads it) and RTLD_LOCAL,
// so that a host process that carries its own RCCL — PyTorch does — keeps its symbols; when that copy answers to the same soname
// the loader hands back the very same library.
//
//   bootstrap     rank 0: dsc_comm_unique_id(&id); the host ships the 128 bytes to the other ranks by whatever it has (MPI, a file,
//                 torch.distributed's store); every rank: dsc_comm_init_rank(ctx, &id, P, rank)
//   collective    dsc_shard_allgather(ctx, comm, dest, rows, row_bytes): ONE in-place ncclAllGather on the persistent destination
//                 dest[P][rows][row_bytes] — this rank's shard already lies in dest[rank], where its transform wrote it — enqueued on
//                 the context's stream, i.e. ordered after the transforms that produced the shard; dsc_synchronize() waits for it
//   mesh          dsc_shard_exchange_rows(...): rows [row0, row0 + n) of every slot in one group of P-1 ncclSend + P-1 ncclRecv — every GPU
//                 talks to all peers at once, one xGMI link per peer, and a chunk can travel while the next one is transformed
#include "dsc_internal.h"

#include <dlfcn.h>

#include <cstring>

namespace {

// the few RCCL types and entry points this file needs, declared here so that no RCCL header is required to build the library
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                            // ncclSuccess = 0
enum { kNcclChar = 0, kNcclFloat = 7 };              // ncclDataType_t: ncclInt8 = 0, ncclFloat32 = 7 (rccl.h)

struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api *rccl() {
    static rccl_api api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    const char *names[] = {getenv("DSC_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (n == nullptr || *n == 0) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) {
        fprintf(stderr, "dsc_comm: cannot load RCCL (%s)\n", dlerror());
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char *name) { void *p = dlsym(api.lib, name); if (!p) { fprintf(stderr, "dsc_comm: RCCL lacks %s\n", name); ok = false; } return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId)) sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank)) sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy)) sym("ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather)) sym("ncclAllGather");
    api.Send = (decltype(api.Send)) sym("ncclSend");
    api.Recv = (decltype(api.Recv)) sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart)) sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd)) sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString)) sym("ncclGetErrorString");
    if (!ok) { dlclose(api.lib); api.lib = nullptr; return nullptr; }
    return &api;
}

}  // namespace

struct dsc_comm {
    ncclComm_t comm;
    int n_ranks, rank, device;
};

#define RCCL_TRY(api, call, ret)                                                                                 \
    do {                                                                                                         \
        ncclResult_t r_ = (call);                                                                                \
        if (r_ != 0) {                                                                                           \
            fprintf(stderr, "%s: %s -> %s\n", __func__, #call, (api)->GetErrorString(r_));                       \
            return ret;                                                                                          \
        }                                                                                                        \
    } while (0)

static_assert(sizeof(dsc_comm_id) == sizeof(ncclUniqueId), "dsc_comm_id carries an ncclUniqueId");

extern "C" int dsc_comm_unique_id(dsc_comm_id *out) {
    DSC_ASSERT(out != nullptr);
    rccl_api *a = rccl();
    if (a == nullptr) return -1;
    ncclUniqueId id;
    RCCL_TRY(a, a->GetUniqueId(&id), -1);
    memcpy(out->bytes, &id, sizeof(id));
    return 0;
}

extern "C" dsc_comm *dsc_comm_init_rank(dsc_ctx *ctx, const dsc_comm_id *id, int n_ranks, int rank) {
    DSC_ASSERT(ctx != nullptr && id != nullptr);
    DSC_ASSERT(n_ranks >= 1 && rank >= 0 && rank < n_ranks);
    rccl_api *a = rccl();
    if (a == nullptr) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    ncclUniqueId uid;
    memcpy(&uid, id->bytes, sizeof(uid));
    ncclComm_t c = nullptr;
    RCCL_TRY(a, a->CommInitRank(&c, n_ranks, uid, rank), nullptr);
    return new dsc_comm{c, n_ranks, rank, ctx->device};
}

extern "C" int dsc_comm_n_ranks(const dsc_comm *comm) { return comm ? comm->n_ranks : 0; }
extern "C" int dsc_comm_rank(const dsc_comm *comm) { return comm ? comm->rank : -1; }

extern "C" void dsc_comm_free(dsc_comm *comm) {
    if (comm == nullptr) return;
    rccl_api *a = rccl();
    if (a != nullptr && comm->comm != nullptr) (void) a->CommDestroy(comm->comm);
    delete comm;
}

// bytes -> (count, datatype): whole floats when the size allows (the shards are c32 / f32 rows), bytes otherwise
static inline void wire_type(size_t nbytes, size_t *count, int *dtype) {
    if (nbytes % 4 == 0) { *count = nbytes / 4; *dtype = kNcclFloat; }
    else { *count = nbytes; *dtype = kNcclChar; }
}

extern "C" int dsc_shard_allgather(dsc_ctx *ctx, dsc_comm *comm, void *dest, size_t rows, size_t row_bytes) {
    DSC_ASSERT(ctx != nullptr && comm != nullptr && dest != nullptr);
    DSC_ASSERT(comm->device == ctx->device);
    rccl_api *a = rccl();
    if (a == nullptr) return -1;
    const size_t shard = rows * row_bytes;
    if (shard == 0) return 0;
    size_t count; int dt;
    wire_type(shard, &count, &dt);
    // in place: the send buffer is this rank's slot of the receive buffer (RCCL's documented in-place form)
    RCCL_TRY(a, a->AllGather((const char *) dest + (size_t) comm->rank * shard, dest, count, dt, comm->comm, ctx->stream), -1);
    return 0;
}

extern "C" int dsc_shard_exchange_rows(dsc_ctx *ctx, dsc_comm *comm, void *dest, size_t rows, size_t row_bytes, size_t row0, size_t n_rows) {
    DSC_ASSERT(ctx != nullptr && comm != nullptr && dest != nullptr);
    DSC_ASSERT(comm->device == ctx->device);
    DSC_ASSERT(row0 + n_rows <= rows);
    rccl_api *a = rccl();
    if (a == nullptr) return -1;
    if (n_rows == 0 || row_bytes == 0 || comm->n_ranks == 1) return 0;
    const size_t shard = rows * row_bytes, off = row0 * row_bytes;
    size_t count; int dt;
    wire_type(n_rows * row_bytes, &count, &dt);
    const int P = comm->n_ranks, me = comm->rank;
    RCCL_TRY(a, a->GroupStart(), -1);
    for (int k = 1; k < P; ++k) {                    // staggered: at step k every rank addresses a different peer
        const int to = (me + k) % P, from = (me - k + P) % P;
        RCCL_TRY(a, a->Send((const char *) dest + (size_t) me * shard + off, count, dt, to, comm->comm, ctx->stream), -1);
        RCCL_TRY(a, a->Recv((char *) dest + (size_t) from * shard + off, count, dt, from, comm->comm, ctx->stream), -1);
    }
    RCCL_TRY(a, a->GroupEnd(), -1);
    return 0;
}
