// comm.hip -- the one collective of the sweep behind the C ABI: an all-gather of the per-image score rows over RCCL / xGMI.
//
// Replaces detection/utils.py:75-115 (`all_gather`: pickle every rank's object, pad the byte tensors to the longest rank, gather, unpickle)
// and the process-group set-up of :302-324 (`init_distributed_mode`, backend "nccl").  The pool shards by position (rank r scores pool
// positions p with p % world == r), so every rank contributes the same number of fixed-size rows
// (consistency, cls_corr[C - 1]) as float64 and no padding protocol or index column is needed: row j of rank r IS position r + j * world.
//
// RCCL is bound at first use with dlopen("librccl.so.1") -- inside a PyTorch process that resolves to the copy torch has already
// loaded (same SONAME), a C / C++ host gets /opt/rocm/lib's -- so the library carries no link-time dependency on a collective library
// that the single-GPU path never touches.  One communicator per (context, process); the collective runs on the context's stream.
#include "common.h"
#include "../../include/cald_hip.h"
#include <dlfcn.h>
#include <mutex>
#include <cstring>

int cald_internal_fail(int code, const char* fmt, ...);
hipStream_t cald_internal_stream(cald_ctx* c);
int cald_internal_device(cald_ctx* c);

namespace {
typedef void* rcclComm_t;
struct RcclId { char internal[128]; };                    // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(rcclComm_t*, int, RcclId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*CommCount)(rcclComm_t, int*) = nullptr;
    int (*CommUserRank)(rcclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* err = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
const int kNcclFloat64 = 8;                                // ncclDataType_t: ncclFloat64 = ncclDouble = 8 (rccl.h)

const Rccl& rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& r = g_rccl;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.so) break; }
        if (!r.so) { r.err = "librccl.so.1 not found (dlopen)"; return; }
        r.GetUniqueId = (int (*)(RcclId*))dlsym(r.so, "ncclGetUniqueId");
        r.CommInitRank = (int (*)(rcclComm_t*, int, RcclId, int))dlsym(r.so, "ncclCommInitRank");
        r.CommDestroy = (int (*)(rcclComm_t))dlsym(r.so, "ncclCommDestroy");
        r.AllGather = (int (*)(const void*, void*, size_t, int, rcclComm_t, hipStream_t))dlsym(r.so, "ncclAllGather");
        r.CommCount = (int (*)(rcclComm_t, int*))dlsym(r.so, "ncclCommCount");
        r.CommUserRank = (int (*)(rcclComm_t, int*))dlsym(r.so, "ncclCommUserRank");
        r.GetErrorString = (const char* (*)(int))dlsym(r.so, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.CommCount || !r.CommUserRank) r.err = "librccl.so.1 lacks an expected symbol";
    });
    return g_rccl;
}
const char* rccl_str(int code) { const Rccl& r = rccl(); return r.GetErrorString ? r.GetErrorString(code) : "?"; }
}   // namespace

// owned: the communicator was created by cald_comm_init_rank and dies with the wrapper; an adopted one belongs to the host framework
struct cald_comm { rcclComm_t comm; cald_ctx* ctx; int world, rank; bool owned; };

extern "C" int cald_comm_unique_id(void* id128_out) {
    if (!id128_out) return cald_internal_fail(CALD_ERR_INVALID, "null argument");
    const Rccl& r = rccl();
    if (r.err) return cald_internal_fail(CALD_ERR_UNSUPPORTED, "RCCL unavailable: %s", r.err);
    RcclId id;
    const int rc = r.GetUniqueId(&id);
    if (rc) return cald_internal_fail(CALD_ERR_HIP, "ncclGetUniqueId failed: %s", rccl_str(rc));
    memcpy(id128_out, &id, sizeof(id));
    return 0;
}

extern "C" int cald_comm_init_rank(cald_ctx* ctx, const void* id128, int world_size, int rank, cald_comm** out) {
    if (!ctx || !id128 || !out) return cald_internal_fail(CALD_ERR_INVALID, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return cald_internal_fail(CALD_ERR_INVALID, "rank %d outside a world of %d", rank, world_size);
    const Rccl& r = rccl();
    if (r.err) return cald_internal_fail(CALD_ERR_UNSUPPORTED, "RCCL unavailable: %s", r.err);
    if (hipSetDevice(cald_internal_device(ctx)) != hipSuccess) return cald_internal_fail(CALD_ERR_HIP, "hipSetDevice failed");
    RcclId id; memcpy(&id, id128, sizeof(id));
    rcclComm_t comm = nullptr;
    const int rc = r.CommInitRank(&comm, world_size, id, rank);      // one process per GPU: the communicator lives on the context's device
    if (rc) return cald_internal_fail(CALD_ERR_HIP, "ncclCommInitRank(world %d, rank %d) failed: %s", world_size, rank, rccl_str(rc));
    cald_comm* c = new cald_comm{comm, ctx, world_size, rank, true};
    *out = c;
    return 0;
}

extern "C" int cald_comm_destroy(cald_comm* comm) {
    if (!comm) return 0;
    const Rccl& r = rccl();
    if (!r.err && comm->comm) {
        hipSetDevice(cald_internal_device(comm->ctx));
        hipStreamSynchronize(cald_internal_stream(comm->ctx));      // a collective of ours may still be in flight on the context's stream
        if (comm->owned) r.CommDestroy(comm->comm);                 // an adopted communicator stays alive: the host framework owns it
    }
    delete comm;
    return 0;
}

// The communicator may also be one the host framework already owns (an ncclComm_t created elsewhere): cald_comm_adopt wraps it without
// taking ownership of anything but the wrapper.
extern "C" int cald_comm_adopt(cald_ctx* ctx, void* rccl_comm, cald_comm** out) {
    if (!ctx || !rccl_comm || !out) return cald_internal_fail(CALD_ERR_INVALID, "null argument");
    const Rccl& r = rccl();
    if (r.err) return cald_internal_fail(CALD_ERR_UNSUPPORTED, "RCCL unavailable: %s", r.err);
    int world = 0, rank = 0;
    int rc = r.CommCount(rccl_comm, &world); if (!rc) rc = r.CommUserRank(rccl_comm, &rank);
    if (rc) return cald_internal_fail(CALD_ERR_HIP, "not a usable RCCL communicator: %s", rccl_str(rc));
    *out = new cald_comm{rccl_comm, ctx, world, rank, false};
    return 0;
}

extern "C" int cald_allgather_scores(cald_comm* comm, const double* send_dev, double* recv_dev, int64_t rows_per_rank, int row_len) {
    if (!comm || !send_dev || !recv_dev) return cald_internal_fail(CALD_ERR_INVALID, "null argument");
    if (rows_per_rank < 0 || row_len < 1) return cald_internal_fail(CALD_ERR_INVALID, "bad geometry (%lld rows x %d)", (long long)rows_per_rank, row_len);
    const Rccl& r = rccl();
    if (r.err) return cald_internal_fail(CALD_ERR_UNSUPPORTED, "RCCL unavailable: %s", r.err);
    if (hipSetDevice(cald_internal_device(comm->ctx)) != hipSuccess) return cald_internal_fail(CALD_ERR_HIP, "hipSetDevice failed");
    if (rows_per_rank == 0) return 0;
    const int rc = r.AllGather(send_dev, recv_dev, (size_t)rows_per_rank * (size_t)row_len, kNcclFloat64, comm->comm, cald_internal_stream(comm->ctx));
    if (rc) return cald_internal_fail(CALD_ERR_HIP, "ncclAllGather failed: %s", rccl_str(rc));
    return 0;          // asynchronous on the context's stream, like every other operator; cald_ctx_sync() or a stream-ordered copy follows
}

extern "C" int cald_comm_info(const cald_comm* comm, int* world_size, int* rank) {
    if (!comm) return cald_internal_fail(CALD_ERR_INVALID, "null argument");
    if (world_size) *world_size = comm->world;
    if (rank) *rank = comm->rank;
    return 0;
}
