// C1: result gather across GPUs -- one process per GPU, RCCL (ncclAllGather) over xGMI.
// Windows are independent, so this is the only collective on the path; it carries the per-window result
// table (a few doubles per window) and replaces the re-ordering role of the reference's sorter / writer
// threads (popgenWindows.py:108-157).  librccl is opened lazily so single-GPU use never loads it.
#include "pg_ctx.h"

#include <dlfcn.h>
#include <cstring>

namespace {

typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
enum { RCCL_INT8 = 0, RCCL_INT32 = 2, RCCL_FLOAT64 = 8 };   // ncclDataType_t values (rccl.h)
enum { RCCL_SUM = 0 };

struct Api {
    void *lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rcclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} api;

int load_api() {
    if (api.lib) return PG_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return pg_fail(PG_ERR_RCCL, "cannot open librccl: %s", dlerror());
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(lib, "ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce || !api.GetErrorString) {
        dlclose(lib);
        return pg_fail(PG_ERR_RCCL, "librccl lacks an expected nccl* symbol");
    }
    api.lib = lib;
    return PG_OK;
}

#define RCCLCHK(expr)                                                                             \
    do {                                                                                          \
        int _r = (expr);                                                                          \
        if (_r != 0) return pg_fail(PG_ERR_RCCL, "%s: %s", #expr, api.GetErrorString(_r));        \
    } while (0)

}  // namespace

extern "C" int pg_comm_unique_id(void *uid128_out) {
    if (!uid128_out) return pg_fail(PG_ERR_ARG, "null output");
    int rc = load_api();
    if (rc != PG_OK) return rc;
    rcclUniqueId id;
    RCCLCHK(api.GetUniqueId(&id));
    memcpy(uid128_out, &id, 128);
    return PG_OK;
}

extern "C" int pg_comm_init(pg_ctx *c, int n_ranks, int rank, const void *uid128) {
    if (!c || !uid128) return pg_fail(PG_ERR_ARG, "pg_comm_init: null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return pg_fail(PG_ERR_ARG, "bad rank %d of %d", rank, n_ranks);
    if (c->comm) return pg_fail(PG_ERR_STATE, "communicator already initialised");
    int rc = load_api();
    if (rc != PG_OK) return rc;
    HIPCHK(hipSetDevice(c->device));
    rcclUniqueId id;
    memcpy(&id, uid128, 128);
    rcclComm_t comm = nullptr;
    RCCLCHK(api.CommInitRank(&comm, n_ranks, id, rank));
    c->comm = comm;
    c->comm_ranks = n_ranks;
    c->comm_rank = rank;
    return PG_OK;
}

extern "C" int pg_comm_allgather_f64(pg_ctx *c, const double *send, double *recv, int64_t count) {
    if (!c || !c->comm) return pg_fail(PG_ERR_STATE, "communicator not initialised");
    if (count < 0 || (count > 0 && (!send || !recv))) return pg_fail(PG_ERR_ARG, "bad buffers");
    if (count == 0) return PG_OK;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    // page-locked staging on both sides: [send | recv], asynchronous copies, one synchronisation
    const size_t n_recv = (size_t)count * c->comm_ranks;
    if ((rc = c->comm_pin.ensure((size_t)count + n_recv)) != PG_OK) return rc;
    memcpy(c->comm_pin.p, send, (size_t)count * 8);
    if ((rc = c->comm_send.upload(c->comm_pin.p, (size_t)count, c->stream)) != PG_OK) return rc;
    if ((rc = c->comm_recv.ensure(n_recv)) != PG_OK) return rc;
    RCCLCHK(api.AllGather(c->comm_send.p, c->comm_recv.p, (size_t)count, RCCL_FLOAT64, c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(c->comm_pin.p + count, c->comm_recv.p, n_recv * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(recv, c->comm_pin.p + count, n_recv * 8);
    return PG_OK;
}

extern "C" int pg_comm_barrier(pg_ctx *c) {
    if (!c || !c->comm) return pg_fail(PG_ERR_STATE, "communicator not initialised");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = c->comm_send.ensure(1)) != PG_OK) return rc;
    if ((rc = c->comm_recv.ensure((size_t)c->comm_ranks > 1 ? c->comm_ranks : 1)) != PG_OK) return rc;
    HIPCHK(hipMemsetAsync(c->comm_send.p, 0, 8, c->stream));
    RCCLCHK(api.AllReduce(c->comm_send.p, c->comm_recv.p, 1, RCCL_FLOAT64, RCCL_SUM, c->comm, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PG_OK;
}

extern "C" int pg_comm_destroy(pg_ctx *c) {
    if (!c) return PG_OK;
    if (c->comm && api.CommDestroy) (void)api.CommDestroy(c->comm);
    c->comm = nullptr;
    c->comm_send.release();
    c->comm_recv.release();
    c->comm_pin.release();
    return PG_OK;
}
